"""TEST INFRASTRUCTURE ONLY: ctypes view of oracle/_ref/libx264ref{8,10}.so (the real reference, C path).

Used by tests/ (to pin the oracle restatement and the host logic) and by bench.py's cpu_baseline leg.
The product package x264_amd never imports this module.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAT = 18 * 18  # (X264_BFRAME_MAX+2)^2


def lib_path(bit_depth=8, seam=False):
    """seam=True: the 8-bit reference built with its accelerator hook bound to libx264hip.so (x264_amd/csrc/slicetype_hip.c in place of
    encoder/slicetype-cl.c / common/opencl.c; oracle/build_ref.sh step 4)"""
    return os.path.join(_HERE, "_ref", "libx264ref%d%s.so" % (bit_depth, "hip" if seam else ""))


def available(bit_depth=8, seam=False):
    return os.path.exists(lib_path(bit_depth, seam))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Ref:
    def __init__(self, width, height, preset="medium", tune="", opts="", bit_depth=8, seam=False):
        self.lib = C.CDLL(lib_path(bit_depth, seam))
        L = self.lib
        L.rh_open.restype = C.c_void_p
        L.rh_open.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
        self.bit_depth = bit_depth
        self.dtype = np.uint8 if bit_depth == 8 else np.uint16
        self.width, self.height = width, height
        self.ctx = L.rh_open(width, height, preset.encode(), tune.encode(), opts.encode())
        if not self.ctx:
            raise RuntimeError("rh_open failed")
        self.ctx = C.c_void_p(self.ctx)
        cfg = np.zeros(40, np.int32)
        L.rh_get_config(self.ctx, _ptr(cfg))
        names = ["mb_w", "mb_h", "bframes", "b_adapt", "rc_lookahead", "mv_range", "me_range", "me_method",
                 "subpel_refine", "lambda", "weightp", "weighted_bipred", "aq_mode", "mb_tree", "scenecut",
                 "keyint_max", "keyint_min", "b_pyramid", "b_bias", "delay", "slicetype_length", "vbv",
                 "open_gop", "intra_refresh", "subme", "mbcmp_satd", "fpelcmp_satd", "psy", "rc_method",
                 "bframe_delay", "refs", "aq_strength_q16", "lookahead_threads", "threads"]
        self.cfg = {k: int(v) for k, v in zip(names, cfg)}
        self.mb_w, self.mb_h = self.cfg["mb_w"], self.cfg["mb_h"]
        self.n_mb = self.mb_w * self.mb_h
        self.n_frames = 0

    def close(self):
        if self.ctx:
            self.lib.rh_close(self.ctx)
            self.ctx = None

    def cost_mv(self):
        self.lib.rh_cost_mv.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        n = self.lib.rh_cost_mv(self.ctx, None, 0)
        out = np.zeros(2 * n + 1, np.uint16)
        self.lib.rh_cost_mv(self.ctx, _ptr(out), out.size)
        return out

    def add_frame(self, luma, cb=None, cr=None):
        """cb / cr: the I420 chroma planes [(H+1)//2, (W+1)//2] (default: mid-grey, i.e. no chroma energy in AQ)"""
        luma = np.ascontiguousarray(luma, dtype=self.dtype)
        assert luma.shape == (self.height, self.width)
        self.lib.rh_add_frame.argtypes = [C.c_void_p] * 4
        if cb is not None:
            cb = np.ascontiguousarray(cb, dtype=self.dtype); cr = np.ascontiguousarray(cr, dtype=self.dtype)
            assert cb.shape == cr.shape  # (H+1)//2 x (W+1)//2 for I420, H x (W+1)//2 for I422, H x W for I444 (csp= option)
        r = self.lib.rh_add_frame(self.ctx, _ptr(luma), _ptr(cb), _ptr(cr))
        assert r >= 0
        self.n_frames = r + 1
        return r

    def frame_cost(self, p0, p1, b):
        return self.lib.rh_frame_cost(self.ctx, p0, p1, b)

    def weights_analyse(self, b, ref):
        out = np.zeros(4, np.int32)
        self.lib.rh_weights_analyse.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.lib.rh_weights_analyse(self.ctx, b, ref, _ptr(out))
        return tuple(int(x) for x in out)

    def weight(self, idx):
        out = np.zeros(4, np.int32)
        self.lib.rh_get_weight.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self.lib.rh_get_weight(self.ctx, idx, _ptr(out))
        return tuple(int(x) for x in out)

    def lowres_geometry(self):
        g = np.zeros(5, np.int32)
        self.lib.rh_lowres_geometry.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.rh_lowres_geometry(self.ctx, _ptr(g))
        return dict(width=int(g[0]), lines=int(g[1]), stride=int(g[2]), padh=int(g[3]), padv=int(g[4]))

    def lowres(self, idx, plane):
        g = self.lowres_geometry()
        out = np.zeros((g["lines"] + 2 * g["padv"], g["width"] + 2 * g["padh"]), self.dtype)
        self.lib.rh_get_lowres.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.lib.rh_get_lowres(self.ctx, idx, plane, _ptr(out))
        return out

    def frame_stats(self, idx):
        inv = np.zeros(self.n_mb, np.uint16)
        intra = np.zeros(self.n_mb, np.uint16)
        ss = np.zeros(2, np.uint64)
        self.lib.rh_get_frame_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.rh_get_frame_stats(self.ctx, idx, _ptr(inv), _ptr(intra), _ptr(ss))
        return inv, intra, (int(ss[0]), int(ss[1]))

    def mvs(self, idx, lst, dist):
        mv = np.zeros((self.n_mb, 2), np.int16)
        cost = np.zeros(self.n_mb, np.int32)
        self.lib.rh_get_mvs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        r = self.lib.rh_get_mvs(self.ctx, idx, lst, dist, _ptr(mv), _ptr(cost))
        assert r == 0
        return mv, cost

    def cell(self, idx, d0, d1):
        lc = np.zeros(self.n_mb, np.uint16)
        rows = np.zeros(self.mb_h, np.int32)
        summ = np.zeros(3, np.int32)
        self.lib.rh_get_cell.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        r = self.lib.rh_get_cell(self.ctx, idx, d0, d1, _ptr(lc), _ptr(rows), _ptr(summ))
        assert r == 0
        return lc, rows, tuple(int(x) for x in summ)

    def accel_state(self):
        """1 = slicetype_frame_cost goes through the accelerator hook, 0 = C path (never asked for, or fell back at open), -1 = the hook failed"""
        return int(self.lib.rh_accel_state(self.ctx))

    def encode_run(self, luma_frames, chroma=None):
        """A whole x264_encoder_encode run.  Returns dict(frame, type, cost, cost_aq, map_crc [n, 8], bytes, stream_crc, seconds), frames in
        coded order (see rh_encode_run in ref_harness.c for what the CRCs cover)."""
        fr = np.ascontiguousarray(luma_frames, dtype=self.dtype)
        n = fr.shape[0]
        luma_only = 1
        if chroma is not None:
            cb, cr = (np.ascontiguousarray(c, dtype=self.dtype).reshape(n, -1) for c in chroma)
            fr = np.ascontiguousarray(np.concatenate([fr.reshape(n, -1), cb, cr], axis=1))
            luma_only = 0
        frame, typ, nbytes = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        cost, cost_aq = np.zeros((n, 18, 18), np.int32), np.zeros((n, 18, 18), np.int32)
        crc = np.zeros((n, 8), np.uint32)
        stream, sec = C.c_uint32(0), C.c_double(0)
        f = self.lib.rh_encode_run
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        r = f(self.ctx, _ptr(fr), n, luma_only, _ptr(frame), _ptr(typ), _ptr(cost), _ptr(cost_aq), _ptr(crc), _ptr(nbytes), C.byref(stream), C.byref(sec))
        assert r == n, (r, n)
        return dict(frame=frame, type=typ, cost=cost, cost_aq=cost_aq, map_crc=crc, bytes=nbytes, stream_crc=int(stream.value), seconds=sec.value)

    def lookahead_run(self, luma_frames, with_qp_offsets=False, forced_types=None, with_vbv=False, rc_cells=None, pts=None, chroma=None, quant_offsets=None):
        """rc_cells: [n, 2] (b-p0, p1-b) per OUTPUT index -> also runs the real x264_rc_analyse_slice on every leaving frame
        (out["rc"][k] = [cost, i_row_satd..., i_row_satds[0][0]...])."""
        """luma_frames: [n, H, W]; returns dict(idx, type, cost, cost_aq, intra_mbs, seconds, seconds_prep[, qp_offset])."""
        fr = np.ascontiguousarray(luma_frames, dtype=self.dtype)
        n = fr.shape[0]
        luma_only = 1
        if chroma is not None:  # (cb, cr): [n, (H+1)//2, (W+1)//2] each -> one Y, Cb, Cr record per frame
            cb, cr = (np.ascontiguousarray(c, dtype=self.dtype).reshape(n, -1) for c in chroma)
            fr = np.ascontiguousarray(np.concatenate([fr.reshape(n, -1), cb, cr], axis=1))
            luma_only = 0
        qp = np.zeros((n, self.n_mb), np.float32) if with_qp_offsets else None
        prop = np.zeros((n, self.n_mb), np.uint16) if with_qp_offsets else None
        self.lib.rh_set_qp_dump.argtypes = [C.c_void_p]
        self.lib.rh_set_prop_dump.argtypes = [C.c_void_p]
        self.lib.rh_set_qp_dump(_ptr(qp))
        self.lib.rh_set_prop_dump(_ptr(prop))
        self.lib.rh_set_quant_offsets.argtypes = [C.c_void_p]
        qo = np.ascontiguousarray(quant_offsets, np.float32) if quant_offsets is not None else None
        assert qo is None or qo.shape == (n, self.n_mb)
        self.lib.rh_set_quant_offsets(_ptr(qo))
        self.lib.rh_set_pts.argtypes = [C.c_void_p]
        ptsa = np.ascontiguousarray(pts, np.int64) if pts is not None else None
        self.lib.rh_set_pts(_ptr(ptsa))
        self.lib.rh_set_forced_types.argtypes = [C.c_void_p]
        ft = np.ascontiguousarray(forced_types, np.int32) if forced_types is not None else None
        assert ft is None or ft.size == n
        self.lib.rh_set_forced_types(_ptr(ft))
        self.lib.rh_set_vbv_dump.argtypes = [C.c_void_p] * 3
        pt = np.zeros((n, 251), np.int32) if with_vbv else None
        ps = np.zeros((n, 251), np.int32) if with_vbv else None
        rows = np.full((n, 18, 18, self.mb_h), -2, np.int32) if with_vbv else None
        self.lib.rh_set_vbv_dump(_ptr(pt), _ptr(ps), _ptr(rows))
        self.lib.rh_set_rc_dump.argtypes = [C.c_void_p] * 2
        rcc = np.ascontiguousarray(rc_cells, np.int32) if rc_cells is not None else None
        rco = np.zeros((n, 1 + 2 * self.mb_h), np.int32) if rc_cells is not None else None
        self.lib.rh_set_rc_dump(_ptr(rcc), _ptr(rco))
        idx = np.zeros(n, np.int32)
        typ = np.zeros(n, np.int32)
        cost = np.zeros((n, 18, 18), np.int32)
        cost_aq = np.zeros((n, 18, 18), np.int32)
        imbs = np.zeros((n, 18), np.int32)
        sec = C.c_double(0)
        sec_prep = C.c_double(0)
        f = self.lib.rh_lookahead_run
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.POINTER(C.c_double)] * 2
        r = f(self.ctx, _ptr(fr), n, luma_only, _ptr(idx), _ptr(typ), _ptr(cost), _ptr(cost_aq), _ptr(imbs),
              C.byref(sec), C.byref(sec_prep))
        self.lib.rh_set_forced_types(None)
        self.lib.rh_set_quant_offsets(None)
        self.lib.rh_set_pts(None)
        self.lib.rh_set_vbv_dump(None, None, None)
        self.lib.rh_set_rc_dump(None, None)
        assert r == n, (r, n)
        self.lib.rh_set_qp_dump(None)
        self.lib.rh_set_prop_dump(None)
        out = dict(idx=idx, type=typ, cost=cost, cost_aq=cost_aq, intra_mbs=imbs,
                   seconds=sec.value, seconds_prep=sec_prep.value)
        if qp is not None:
            out["qp_offset"] = qp
            out["propagate"] = prop
        if with_vbv:
            out["planned_type"], out["planned_satd"], out["row_satds"] = pt, ps, rows
        if rco is not None:
            out["rc"] = rco
        return out
