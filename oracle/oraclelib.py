"""TEST INFRASTRUCTURE ONLY: ctypes view of oracle/liboracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PAD = 32


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


class LaCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "mb_w", "mb_h", "stride", "lambda_", "me_method", "subpel_refine", "me_range", "mv_range", "subme",
        "mbcmp_satd", "fpelcmp_satd", "weighted_bipred", "aq_mode", "bframe_bias", "n_slices", "do_edges")] + \
        [("cost_mv", C.c_void_p)]


class Weight(C.Structure):
    _fields_ = [("on", C.c_int), ("scale", C.c_int), ("denom", C.c_int), ("offset", C.c_int)]


class CellOut(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("cost_est", "cost_est_aq", "intra_mbs", "intra_cost_est", "intra_cost_est_aq")]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def plane_stride(width_lowres):
    return (width_lowres + 2 * PAD + 63) // 64 * 64


class Oracle:
    """Bit-depth bound view; planes are numpy arrays [lines+2*PAD, stride] with origin at [PAD, PAD]."""

    def __init__(self, bit_depth=8):
        self.lib = _lib()
        self.d = bit_depth
        self.pfx = "or%d_" % bit_depth
        self.dtype = np.uint8 if bit_depth == 8 else np.uint16
        self.coef_dtype = np.int16 if bit_depth == 8 else np.int32
        self.ucoef_dtype = np.uint16 if bit_depth == 8 else np.uint32
        self.isz = np.dtype(self.dtype).itemsize

    def f(self, name, restype=None):
        fn = getattr(self.lib, self.pfx + name)
        fn.restype = restype
        return fn

    # ---- configuration -------------------------------------------------------------------------
    def make_cfg(self, mb_w, mb_h, *, me_method, subpel_refine, me_range, mv_range, subme, mbcmp_satd,
                 fpelcmp_satd=0, weighted_bipred=1, aq_mode=1, bframe_bias=0, lam=None, cost_mv=None, n_slices=1,
                 do_edges=1):
        lam = lam if lam is not None else (1 if self.d == 8 else 4)
        n = 2 * 4 * mv_range
        if cost_mv is None:
            cost_mv = np.zeros(2 * n + 1, np.uint16)
            self.lib.or_cost_mv_table(C.c_void_p(cost_mv.ctypes.data + 2 * n), n, lam)
        self._cost_mv = cost_mv
        cfg = LaCfg(mb_w, mb_h, plane_stride(8 * mb_w), lam, me_method, subpel_refine, me_range, mv_range, subme,
                    mbcmp_satd, fpelcmp_satd, weighted_bipred, aq_mode, bframe_bias, n_slices,
                    int(bool(do_edges) or mb_w <= 2 or mb_h <= 2),
                    cost_mv.ctypes.data + 2 * n)
        cfg._keep = cost_mv
        return cfg

    # ---- planes --------------------------------------------------------------------------------
    def alloc_planes(self, cfg, n=4):
        return np.zeros((n, 8 * cfg.mb_h + 2 * PAD, cfg.stride), self.dtype)

    def origin(self, plane):
        """address of pixel (0,0) of a padded plane"""
        return plane.ctypes.data + (PAD * plane.shape[-1] + PAD) * self.isz

    def lowres_init(self, cfg, luma):
        luma = np.ascontiguousarray(luma, self.dtype)
        h, w = luma.shape
        pl = self.alloc_planes(cfg)
        self.f("lowres_init")(_p(luma), w, w, h, cfg.mb_w, cfg.mb_h,
                              *[C.c_void_p(self.origin(pl[i])) for i in range(4)], cfg.stride)
        return pl

    def intra_costs(self, cfg, planes):
        out = np.zeros(cfg.mb_w * cfg.mb_h, np.uint16)
        self.f("intra_costs")(C.byref(cfg), C.c_void_p(self.origin(planes[0])), _p(out))
        return out

    def _plane_ptrs(self, planes):
        arr = (C.c_void_p * 4)(*[self.origin(planes[i]) for i in range(4)])
        return arr

    def weight_plane(self, cfg, plane0, wt):
        dst = np.zeros_like(plane0)
        self.f("weight_plane")(C.c_void_p(self.origin(dst)), C.c_void_p(self.origin(plane0)), cfg.stride,
                               8 * cfg.mb_w, 8 * cfg.mb_h, C.byref(wt))
        return dst

    def weight_cost(self, cfg, fenc_planes, ref_planes, wt, intra_cost):
        fn = self.f("weight_cost", C.c_uint)
        return fn(C.byref(cfg), C.c_void_p(self.origin(fenc_planes[0])), C.c_void_p(self.origin(ref_planes[0])),
                  C.byref(wt) if wt is not None else None, _p(intra_cost))

    def search_field(self, cfg, fenc_planes, ref_planes, wt=None, wplane=None):
        n = cfg.mb_w * cfg.mb_h
        mvs = np.zeros((n, 2), np.int16)
        costs = np.zeros(n, np.int32)
        self.f("search_field")(C.byref(cfg), C.c_void_p(self.origin(fenc_planes[0])), self._plane_ptrs(ref_planes),
                               C.c_void_p(self.origin(wplane)) if wplane is not None else None,
                               C.byref(wt) if wt is not None else None, _p(mvs), _p(costs))
        return mvs, costs

    def cell(self, cfg, fenc_planes, ref0, ref1, dist_scale_factor, mvs0, costs0, mvs1, costs1, ref1_l0_mvs,
             intra_cost, inv_qscale, with_intra, alias_intra=False):
        """ref0/ref1 None for the intra-only cell; ref1 None for P."""
        n = cfg.mb_w * cfg.mb_h
        lc = intra_cost if alias_intra else np.zeros(n, np.uint16)
        rows = np.zeros(cfg.mb_h, np.int32)
        rows_i = np.zeros(cfg.mb_h, np.int32)
        out = CellOut()
        b_bidir = 1 if ref1 is not None else 0
        self.f("cell")(C.byref(cfg), C.c_void_p(self.origin(fenc_planes[0])),
                       self._plane_ptrs(ref0) if ref0 is not None else None,
                       self._plane_ptrs(ref1) if ref1 is not None else None,
                       b_bidir, dist_scale_factor, None, _p(mvs0), _p(costs0), _p(mvs1), _p(costs1),
                       _p(ref1_l0_mvs), _p(intra_cost), _p(inv_qscale), int(with_intra), _p(lc), _p(rows), _p(rows_i),
                       C.byref(out))
        return lc, rows, rows_i, out

    def aq_frame(self, luma, mb_w, mb_h, aq_mode=1, aq_strength=1.0, cb=None, cr=None, chroma_format=1, quant_offsets=None):
        luma = np.ascontiguousarray(luma, self.dtype)
        h, w = luma.shape
        inv = np.zeros(mb_w * mb_h, np.uint16)
        qp = np.zeros(mb_w * mb_h, np.float32)
        ssd = C.c_uint64(0)
        fn = self.f("aq_frame_fmt", C.c_uint64)
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                       C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.c_void_p]
        cstride = cb.shape[1] if cb is not None else (w + 1) // 2
        s = fn(_p(luma), w, w, h, mb_w, mb_h, _p(cb), _p(cr), cstride, aq_mode, aq_strength, _p(inv), _p(qp),
               C.byref(ssd), chroma_format, _p(np.ascontiguousarray(quant_offsets, np.float32)) if quant_offsets is not None else None)
        return inv, qp, int(s), int(ssd.value)
