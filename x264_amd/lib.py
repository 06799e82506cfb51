"""ctypes binding of the C ABI in include/x264hip.h (libx264hip.so).  No CPU fallback: every call
goes to the HIP library and errors surface as X264HipError."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libx264hip.so")
BFRAME_MAX = 16


class X264HipError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__("%s failed: %s (%d)" % (what, _strerror(code), code))


class Params(C.Structure):
    _fields_ = [("bit_depth", C.c_int), ("width", C.c_int), ("height", C.c_int), ("bframes", C.c_int), ("lambda_", C.c_int),
                ("me_method", C.c_int), ("subpel_refine", C.c_int), ("me_range", C.c_int), ("mv_range", C.c_int),
                ("subme", C.c_int), ("mbcmp_satd", C.c_int), ("fpelcmp_satd", C.c_int), ("weighted_bipred", C.c_int),
                ("aq_mode", C.c_int), ("aq_strength", C.c_float), ("bframe_bias", C.c_int), ("max_frames", C.c_int),
                ("no_edges", C.c_int), ("lookahead_slices", C.c_int), ("chroma_format", C.c_int), ("cost_mv", C.c_void_p)]


class Weight(C.Structure):
    _fields_ = [("on", C.c_int), ("scale", C.c_int), ("denom", C.c_int), ("offset", C.c_int)]


class Cost(C.Structure):
    _fields_ = [("cost_est", C.c_int), ("cost_est_aq", C.c_int), ("intra_mbs", C.c_int), ("intra_cost_est", C.c_int),
                ("intra_cost_est_aq", C.c_int)]


ME_MVC_MAX = 10


class MeRequest(C.Structure):
    """x264hip_me_request"""
    _fields_ = [("i_pixel", C.c_int), ("me_method", C.c_int), ("subpel_refine", C.c_int), ("me_range", C.c_int), ("mbcmp_satd", C.c_int),
                ("fpelcmp_satd", C.c_int), ("x", C.c_int), ("y", C.c_int), ("mvp", C.c_int * 2), ("lim_min", C.c_int * 2),
                ("lim_max", C.c_int * 2), ("spel_min", C.c_int * 2), ("spel_max", C.c_int * 2), ("n_mvc", C.c_int),
                ("mvc", (C.c_int16 * 2) * ME_MVC_MAX)]


class MbtreeOp(C.Structure):
    _fields_ = [("type", C.c_int), ("slot_b", C.c_int), ("slot_p0", C.c_int), ("slot_p1", C.c_int), ("dist_p0", C.c_int),
                ("dist_p1", C.c_int), ("referenced", C.c_int), ("bipred_weight", C.c_int), ("fps_factor", C.c_float),
                ("fps_factor_i", C.c_int), ("weightdelta", C.c_float), ("strength", C.c_float)]


class CellRef(C.Structure):
    """x264hip_cell_ref"""
    _fields_ = [("slot_b", C.c_int), ("slot_p0", C.c_int), ("slot_p1", C.c_int), ("dist_p0", C.c_int), ("dist_p1", C.c_int), ("with_ref1_l0", C.c_int)]


_lib = None


def load():
    """Load libx264hip.so; raises if it has not been built (the product never falls back to CPU code)."""
    global _lib
    if _lib is None:
        path = os.environ.get("X264HIP_LIB", LIB_PATH)  # X264HIP_LIB: an alternative build of the same library (profiling builds)
        if not os.path.exists(path):
            raise ImportError("%s is missing: run `python -m x264_amd.build` (needs hipcc)" % path)
        _lib = C.CDLL(path)
        _lib.x264hip_strerror.restype = C.c_char_p
        _lib.x264hip_open.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Params)]
    return _lib


def _strerror(code):
    try:
        return load().x264hip_strerror(code).decode()
    except Exception:
        return "?"


def _ck(code, what):
    if code != 0:
        raise X264HipError(code, what)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def cost_mv_table(mv_range, lam):
    """Centred cost_mv table exactly as x264_analyse_init_costs builds it (encoder/analyse.c:143-202):
    float32 log2f, lambda*log + .5f truncated, saturated to u16.  Returns (array, centre index)."""
    n = 2 * 4 * mv_range
    i = np.arange(0, n + 1, dtype=np.float32)
    logs = np.log2(i + np.float32(1.0)).astype(np.float32) * np.float32(2.0) + np.float32(1.718)
    logs[0] = np.float32(0.718)
    v = (np.float32(lam) * logs + np.float32(0.5)).astype(np.int64)
    v = np.minimum(v, 65535).astype(np.uint16)
    return np.concatenate([v[:0:-1], v]), n


def search_profile(L, ctx_handle, enable=-1):
    """(total_ms, launches, searches) of the search kernel measured with HIP events on its own stream."""
    ms, nl, ns = C.c_double(), C.c_uint64(), C.c_uint64()
    _ck(L.x264hip_search_profile(ctx_handle, int(enable), C.byref(ms), C.byref(nl), C.byref(ns)), "search_profile")
    return ms.value, int(nl.value), int(ns.value)


def search_profile_latency(L, ctx_handle):
    """(total_ms, launches, searches) of the profiled search launches that ran on the latency form (read before search_profile resets)"""
    ms, nl, ns = C.c_double(), C.c_uint64(), C.c_uint64()
    _ck(L.x264hip_search_profile_latency(ctx_handle, C.byref(ms), C.byref(nl), C.byref(ns)), "search_profile_latency")
    return ms.value, int(nl.value), int(ns.value)


def kernel_profile(L, ctx_handle):
    """[(total_ms, launches, units)] per X264HIP_KPROF_* class (x264hip_kernel_profile; filled while x264hip_search_profile was on with bit 1 set)"""
    ms, nl, nu = (C.c_double * 6)(), (C.c_uint64 * 6)(), (C.c_uint64 * 6)()
    L.x264hip_kernel_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _ck(L.x264hip_kernel_profile(ctx_handle, ms, nl, nu), "kernel_profile")
    return [(ms[k], nl[k], nu[k]) for k in range(6)]


def cell_profile(L, ctx_handle):
    """(total_ms, launches, cells) of the cost cell launches in the window x264hip_search_profile opened."""
    ms, nl, ns = C.c_double(), C.c_uint64(), C.c_uint64()
    _ck(L.x264hip_cell_profile(ctx_handle, C.byref(ms), C.byref(nl), C.byref(ns)), "cell_profile")
    return ms.value, int(nl.value), int(ns.value)


class Context:
    """Thin object view of x264hip_ctx."""

    def __init__(self, width, height, *, bit_depth=8, bframes=3, lam=None, me_method=1, subpel_refine=4, me_range=16,
                 mv_range=512, subme=7, mbcmp_satd=1, fpelcmp_satd=0, weighted_bipred=1, aq_mode=1, aq_strength=1.0,
                 bframe_bias=0, max_frames=64, cost_mv=None, device=0, no_edges=0, lookahead_slices=1, chroma_format=1):
        L = load()
        lam = lam if lam is not None else (1 if bit_depth == 8 else 4)
        if cost_mv is None:
            cost_mv, centre = cost_mv_table(mv_range, lam)
        else:
            cost_mv = np.ascontiguousarray(cost_mv, np.uint16)
            centre = (cost_mv.size - 1) // 2
        self._cost_mv = cost_mv
        self.params = Params(bit_depth, width, height, bframes, lam, me_method, subpel_refine, me_range, mv_range, subme,
                             mbcmp_satd, fpelcmp_satd, weighted_bipred, aq_mode, aq_strength, bframe_bias, max_frames,
                             no_edges, lookahead_slices, chroma_format, cost_mv.ctypes.data + 2 * centre)
        self.h = C.c_void_p()
        _ck(L.x264hip_open(C.byref(self.h), device, C.byref(self.params)), "x264hip_open")
        self.L = L
        self.dtype = np.uint8 if bit_depth == 8 else np.uint16
        mw, mh, st = C.c_int(), C.c_int(), C.c_int()
        _ck(L.x264hip_geometry(self.h, C.byref(mw), C.byref(mh), C.byref(st)), "geometry")
        self.mb_w, self.mb_h, self.stride = mw.value, mh.value, st.value
        self.n_mb = self.mb_w * self.mb_h
        self.bframes = bframes
        self.width, self.height = width, height

    def close(self):
        if self.h:
            self.L.x264hip_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_name(self):
        buf = C.create_string_buffer(256)
        _ck(self.L.x264hip_device_name(self.h, buf, 256), "device_name")
        return buf.value.decode()

    def frame_put(self, slot, luma, inv_qscale=None, device_ptr=None, stride=None):
        if device_ptr is not None:
            _ck(self.L.x264hip_frame_put(self.h, slot, C.c_void_p(device_ptr), stride or self.width, 1, None, None, 0,
                                         _p(inv_qscale)), "frame_put")
            return
        luma = np.ascontiguousarray(luma, self.dtype)
        assert luma.shape == (self.height, self.width)
        _ck(self.L.x264hip_frame_put(self.h, slot, _p(luma), self.width, 0, None, None, 0, _p(inv_qscale)), "frame_put")

    def frame_stats(self, slot):
        s, q = C.c_uint64(), C.c_uint64()
        _ck(self.L.x264hip_frame_stats(self.h, slot, C.byref(s), C.byref(q)), "frame_stats")
        return int(s.value), int(q.value)

    def frame_cost(self, slot_p0, slot_p1, slot_b, d0, d1, do_search=(0, 0), weight=None, with_intra=False,
                   ref1_l0_valid=False):
        ds = (C.c_int * 2)(*[int(x) for x in do_search])
        out = Cost()
        w = Weight(*weight) if weight is not None else None
        _ck(self.L.x264hip_frame_cost(self.h, slot_p0, slot_p1, slot_b, d0, d1, ds, C.byref(w) if w else None,
                                      int(with_intra), int(ref1_l0_valid), C.byref(out)), "frame_cost")
        return out

    def weight_cost(self, slot_fenc, slot_ref, weight=None):
        c = C.c_uint()
        w = Weight(*weight) if weight is not None else None
        _ck(self.L.x264hip_weight_cost(self.h, slot_fenc, slot_ref, C.byref(w) if w else None, C.byref(c)), "weight_cost")
        return int(c.value)

    def prefetch(self, slots, frame_numbers):
        s = np.asarray(slots, np.int32)
        f = np.asarray(frame_numbers, np.int32)
        _ck(self.L.x264hip_prefetch(self.h, _p(s), _p(f), int(s.size)), "prefetch")

    def synchronize(self):
        _ck(self.L.x264hip_synchronize(self.h), "synchronize")

    def lowres(self, slot, plane):
        out = np.zeros((8 * self.mb_h + 64, 8 * self.mb_w + 64), self.dtype)
        _ck(self.L.x264hip_get_lowres(self.h, slot, plane, _p(out), out.shape[1]), "get_lowres")
        return out

    def mc_luma_probe(self, slot, reqs, weight=None):
        """x264hip_mc_luma_probe: reqs = (n, 4) int32 rows (x, y, mvx, mvy) -> (n, 8, 8) predicted blocks"""
        reqs = np.ascontiguousarray(reqs, np.int32).reshape(-1, 4)
        out = np.zeros((len(reqs), 8, 8), self.dtype)
        w = Weight(*weight) if weight is not None else None
        _ck(self.L.x264hip_mc_luma_probe(self.h, slot, len(reqs), _p(reqs), C.byref(w) if w else None, _p(out)), "mc_luma_probe")
        return out

    def mvs(self, slot, lst, dist_m1):
        mv = np.zeros((self.n_mb, 2), np.int16)
        cost = np.zeros(self.n_mb, np.int32)
        _ck(self.L.x264hip_get_mvs(self.h, slot, lst, dist_m1, _p(mv), _p(cost)), "get_mvs")
        return mv, cost

    def lowres_costs(self, slot, d0, d1):
        lc = np.zeros(self.n_mb, np.uint16)
        rows = np.zeros(self.mb_h, np.int32)
        _ck(self.L.x264hip_get_lowres_costs(self.h, slot, d0, d1, _p(lc), _p(rows)), "get_lowres_costs")
        return lc, rows

    def intra_costs(self, slot):
        return self.lowres_costs(slot, 0, 0)[0]

    def inv_qscale(self, slot):
        out = np.zeros(self.n_mb, np.uint16)
        _ck(self.L.x264hip_get_inv_qscale(self.h, slot, _p(out)), "get_inv_qscale")
        return out

    def qp_offsets(self, slot):
        out = np.zeros(self.n_mb, np.float32)
        _ck(self.L.x264hip_get_qp_offsets(self.h, slot, _p(out)), "get_qp_offsets")
        return out

    def propagate_cost(self, slot):
        out = np.zeros(self.n_mb, np.uint16)
        _ck(self.L.x264hip_get_propagate_cost(self.h, slot, _p(out)), "get_propagate_cost")
        return out

    def frame_cost_recalculate(self, slot_b, d0, d1, use_aq_offsets=False):
        score = C.c_int()
        _ck(self.L.x264hip_frame_cost_recalculate(self.h, slot_b, d0, d1, int(use_aq_offsets), C.byref(score)), "frame_cost_recalculate")
        return score.value

    def mbtree(self, ops):
        arr = (MbtreeOp * len(ops))(*ops)
        _ck(self.L.x264hip_mbtree(self.h, arr, len(ops)), "mbtree")

    # ---- batched vtable primitives (device pointers are plain integers) ----
    def pixel_cmp_batch(self, satd, size_idx, fenc_ptr, ref_ptr, stride, blocks_w, blocks_h, mv_ptr, out_ptr):
        _ck(self.L.x264hip_pixel_cmp_batch(self.h, int(satd), int(size_idx), C.c_void_p(fenc_ptr), C.c_void_p(ref_ptr), int(stride),
                                           int(blocks_w), int(blocks_h), C.c_void_p(mv_ptr), C.c_void_p(out_ptr)), "pixel_cmp_batch")

    @staticmethod
    def me_requests(reqs):
        """the request table of x264hip_me_search_batch as the C array the call takes"""
        return (MeRequest * len(reqs))(*reqs)

    def me_search_batch(self, reqs, fenc_ptr, fenc_stride, ref_ptrs, ref_stride, integral_ptr, integral_lower, cost_mv_ptr):
        """reqs: list of MeRequest, or the array me_requests() makes of one (a caller that repeats a batch builds it once);
        pointers are device addresses of pixel / element (0,0).  Returns int32 [n, 4]."""
        n = len(reqs)
        arr = reqs if isinstance(reqs, C.Array) else (MeRequest * n)(*reqs)
        refs = (C.c_void_p * 4)(*ref_ptrs)
        out = np.zeros((n, 4), np.int32)
        self.L.x264hip_me_search_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t,
                                                   C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_void_p]
        _ck(self.L.x264hip_me_search_batch(self.h, n, arr, fenc_ptr, fenc_stride, refs, ref_stride, integral_ptr, integral_lower,
                                           cost_mv_ptr, _p(out)), "me_search_batch")
        return out

    def me_search_batch_dev(self, n, reqs_ptr, fenc_ptr, fenc_stride, ref_ptrs, ref_stride, integral_ptr, integral_lower, cost_mv_ptr, me_method, me_range_max, out_ptr):
        """x264hip_me_search_batch_dev: request table ([n] MeRequest) and results ([n, 4] int32) on the device; enqueued, not waited for"""
        refs = (C.c_void_p * 4)(*ref_ptrs)
        self.L.x264hip_me_search_batch_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t,
                                                       C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _ck(self.L.x264hip_me_search_batch_dev(self.h, int(n), reqs_ptr, fenc_ptr, fenc_stride, refs, ref_stride, integral_ptr, integral_lower, cost_mv_ptr,
                                               int(me_method), int(me_range_max), out_ptr), "me_search_batch_dev")

    def frame_filter(self, luma_ptr, luma_stride, width, height, plane_ptrs, stride, padh, padv, sum8_ptr=None, sum4_ptr=None):
        self.L.x264hip_frame_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int,
                                                C.c_void_p, C.c_void_p]
        pl = (C.c_void_p * 4)(*plane_ptrs)
        _ck(self.L.x264hip_frame_filter(self.h, luma_ptr, luma_stride, width, height, pl, stride, padh, padv, sum8_ptr, sum4_ptr), "frame_filter")

    def integral_init(self, plane_ptr, stride, width, height, sum8_ptr, sum4_ptr):
        self.L.x264hip_integral_init.argtypes = [C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _ck(self.L.x264hip_integral_init(self.h, plane_ptr, stride, width, height, sum8_ptr, sum4_ptr), "integral_init")

    def pixel_metric_batch(self, metric, size_idx, a_ptr, b_ptr, stride, blocks_w, blocks_h, out_ptr):
        self.L.x264hip_pixel_metric_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int,
                                                      C.c_void_p]
        _ck(self.L.x264hip_pixel_metric_batch(self.h, metric, size_idx, a_ptr, b_ptr, stride, blocks_w, blocks_h, out_ptr), "pixel_metric_batch")

    def frame_dct_quant4x4(self, fenc_ptr, fenc_stride, fdec_ptr, fdec_stride, width, height, mf, bias, coefs_ptr, nz_ptr):
        mf = np.ascontiguousarray(mf); bias = np.ascontiguousarray(bias)
        _ck(self.L.x264hip_frame_dct_quant4x4(self.h, C.c_void_p(fenc_ptr), C.c_ssize_t(fenc_stride), C.c_void_p(fdec_ptr), C.c_ssize_t(fdec_stride),
                                              int(width), int(height), _p(mf), _p(bias), C.c_void_p(coefs_ptr), C.c_void_p(nz_ptr)), "frame_dct_quant4x4")

    def frame_dct_quant8x8(self, fenc_ptr, fenc_stride, fdec_ptr, fdec_stride, width, height, mf, bias, coefs_ptr, nz_ptr):
        mf = np.ascontiguousarray(mf); bias = np.ascontiguousarray(bias)
        _ck(self.L.x264hip_frame_dct_quant8x8(self.h, C.c_void_p(fenc_ptr), C.c_ssize_t(fenc_stride), C.c_void_p(fdec_ptr), C.c_ssize_t(fdec_stride),
                                              int(width), int(height), _p(mf), _p(bias), C.c_void_p(coefs_ptr), C.c_void_p(nz_ptr)), "frame_dct_quant8x8")

    def hpel_filter(self, dsth_ptr, dstv_ptr, dstc_ptr, src_ptr, stride, width, height):
        _ck(self.L.x264hip_hpel_filter(self.h, C.c_void_p(dsth_ptr), C.c_void_p(dstv_ptr), C.c_void_p(dstc_ptr), C.c_void_p(src_ptr),
                                       C.c_ssize_t(stride), int(width), int(height)), "hpel_filter")

    # ---- multi-plane forms: lists of device addresses, one per plane set (x264hip_*_multi) ----
    @staticmethod
    def _ptrs(lst):
        return (C.c_void_p * len(lst))(*[int(v) for v in lst])

    def pixel_cmp_batch_multi(self, satd, size_idx, fenc_ptrs, ref_ptrs, stride, blocks_w, blocks_h, mv_ptrs, out_ptrs):
        _ck(self.L.x264hip_pixel_cmp_batch_multi(self.h, int(satd), int(size_idx), len(fenc_ptrs), self._ptrs(fenc_ptrs), self._ptrs(ref_ptrs), int(stride),
                                                 int(blocks_w), int(blocks_h), self._ptrs(mv_ptrs), self._ptrs(out_ptrs)), "pixel_cmp_batch_multi")

    def hpel_filter_multi(self, dsth_ptrs, dstv_ptrs, dstc_ptrs, src_ptrs, stride, width, height):
        _ck(self.L.x264hip_hpel_filter_multi(self.h, len(src_ptrs), self._ptrs(dsth_ptrs), self._ptrs(dstv_ptrs), self._ptrs(dstc_ptrs), self._ptrs(src_ptrs),
                                             C.c_ssize_t(stride), int(width), int(height)), "hpel_filter_multi")

    def frame_dct_quant4x4_multi(self, fenc_ptrs, fenc_stride, fdec_ptrs, fdec_stride, width, height, mf, bias, coefs_ptrs, nz_ptrs):
        mf = np.ascontiguousarray(mf); bias = np.ascontiguousarray(bias)
        _ck(self.L.x264hip_frame_dct_quant4x4_multi(self.h, len(fenc_ptrs), self._ptrs(fenc_ptrs), C.c_ssize_t(fenc_stride), self._ptrs(fdec_ptrs), C.c_ssize_t(fdec_stride),
                                                    int(width), int(height), _p(mf), _p(bias), self._ptrs(coefs_ptrs), self._ptrs(nz_ptrs)), "frame_dct_quant4x4_multi")

    def device_copy(self, dst_ptr, src_ptr, nbytes):
        _ck(self.L.x264hip_device_copy(self.h, C.c_void_p(dst_ptr), C.c_void_p(src_ptr), C.c_size_t(nbytes)), "device_copy")

    def frame_init_lowres_core(self, src_ptr, dst_ptrs, src_stride, dst_stride, width, height):
        _ck(self.L.x264hip_frame_init_lowres_core(self.h, C.c_void_p(src_ptr), *[C.c_void_p(p) for p in dst_ptrs], C.c_ssize_t(src_stride),
                                                  C.c_ssize_t(dst_stride), int(width), int(height)), "frame_init_lowres_core")

    def dct_quant_batch(self, is8x8, fenc, fdec, mf, bias):
        """fenc: (n, N, 16) pixels, fdec: (n, N, 32) pixels (host arrays); returns (coefs (n, N*N), nz (n,))."""
        n, N = fenc.shape[0], 8 if is8x8 else 4
        cdt = np.int16 if self.params.bit_depth == 8 else np.int32
        fenc = np.ascontiguousarray(fenc, self.dtype); fdec = np.ascontiguousarray(fdec, self.dtype)
        mf = np.ascontiguousarray(mf); bias = np.ascontiguousarray(bias)
        coefs = np.zeros((n, N * N), cdt); nz = np.zeros(n, np.int32)
        _ck(self.L.x264hip_dct_quant_batch(self.h, int(is8x8), n, _p(fenc), _p(fdec), _p(mf), _p(bias), _p(coefs), _p(nz)), "dct_quant_batch")
        return coefs, nz

    # ---- the remaining vtable entries in batch form (x264hip_dct_batch / quant_batch / var2_batch / ads_batch) ----
    DCT_COEFS = {0: 16, 1: 64, 2: 256, 3: 64, 4: 256, 5: 4, 6: 8, 7: 16, 8: 8}
    QUANT_COEFS = {0: 16, 1: 64, 2: 64, 3: 16, 4: 4}

    def dct_batch(self, kind, fenc=None, fdec=None, coefs=None):
        """fenc (n,16,16) / fdec (n,16,32) pixel buffers; kinds 7, 8 transform `coefs` (n, 16 / 8) in place.  Returns coefs (n, count)."""
        cdt = np.int16 if self.params.bit_depth == 8 else np.int32
        if kind >= 7:
            out = np.ascontiguousarray(coefs, cdt).copy()
            n = out.shape[0]
            _ck(self.L.x264hip_dct_batch(self.h, kind, n, None, None, _p(out)), "dct_batch")
            return out
        fenc = np.ascontiguousarray(fenc, self.dtype); fdec = np.ascontiguousarray(fdec, self.dtype)
        n = fenc.shape[0]
        assert fenc.shape[1:] == (16, 16) and fdec.shape[1:] == (16, 32)
        out = np.zeros((n, self.DCT_COEFS[kind]), cdt)
        _ck(self.L.x264hip_dct_batch(self.h, kind, n, _p(fenc), _p(fdec), _p(out)), "dct_batch")
        return out

    def quant_batch(self, kind, coefs, mf=None, bias=None, mf_dc=0, bias_dc=0):
        cdt = np.int16 if self.params.bit_depth == 8 else np.int32
        udt = np.uint16 if self.params.bit_depth == 8 else np.uint32
        out = np.ascontiguousarray(coefs, cdt).copy()
        n = out.shape[0]
        assert out.shape[1] == self.QUANT_COEFS[kind]
        nz = np.zeros(n, np.int32)
        mf = np.ascontiguousarray(mf, udt) if mf is not None else None
        bias = np.ascontiguousarray(bias, udt) if bias is not None else None
        _ck(self.L.x264hip_quant_batch(self.h, kind, n, _p(out), _p(mf), _p(bias), int(mf_dc), int(bias_dc), _p(nz)), "quant_batch")
        return out, nz

    def var2_batch(self, height, fenc, fdec):
        fenc = np.ascontiguousarray(fenc, self.dtype); fdec = np.ascontiguousarray(fdec, self.dtype)
        n = fenc.shape[0]
        var = np.zeros(n, np.int32); ssd = np.zeros((n, 2), np.int32)
        _ck(self.L.x264hip_var2_batch(self.h, int(height), n, _p(fenc), _p(fdec), _p(var), _p(ssd)), "var2_batch")
        return var, ssd

    def ads_batch(self, calls, sums, cost_mvx, n_mvs):
        """calls: list of dicts(n_dc, delta, width, thresh, enc_dc[<=4], sums_off, cost_off, mvs_off); returns (mvs, counts)."""
        arr = (AdsCall * len(calls))()
        for a, c in zip(arr, calls):
            a.n_dc, a.delta, a.width, a.thresh = c["n_dc"], c["delta"], c["width"], c["thresh"]
            for k, v in enumerate(c["enc_dc"]):
                a.enc_dc[k] = int(v)
            a.sums_off, a.cost_off, a.mvs_off = c["sums_off"], c["cost_off"], c["mvs_off"]
        sums = np.ascontiguousarray(sums, np.uint16); cost_mvx = np.ascontiguousarray(cost_mvx, np.uint16)
        mvs = np.zeros(n_mvs, np.int16); counts = np.zeros(len(calls), np.int32)
        self.L.x264hip_ads_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        _ck(self.L.x264hip_ads_batch(self.h, len(calls), C.cast(arr, C.c_void_p), _p(sums), sums.size, _p(cost_mvx), cost_mvx.size, _p(mvs), mvs.size, _p(counts)),
            "ads_batch")
        return mvs, counts

    def last_search_ms(self):
        ms, ns, nb = C.c_float(), C.c_int(), C.c_int()
        _ck(self.L.x264hip_last_search_ms(self.h, C.byref(ms), C.byref(ns), C.byref(nb)), "last_search_ms")
        return ms.value, ns.value, nb.value

    def counters(self):
        out = np.zeros(8, np.uint64)
        _ck(self.L.x264hip_counters(self.h, _p(out), 8), "counters")
        return out

    def search_profile(self, enable=-1):
        return search_profile(self.L, self.h, enable)


class AdsCall(C.Structure):
    """x264hip_ads_call"""
    _fields_ = [("n_dc", C.c_int), ("delta", C.c_int), ("width", C.c_int), ("thresh", C.c_int), ("enc_dc", C.c_int * 4),
                ("sums_off", C.c_longlong), ("cost_off", C.c_longlong), ("mvs_off", C.c_longlong)]


# ---- host-side lookahead (x264hip_lookahead_*) -----------------------------------------------------------
class LaParams(C.Structure):
    _fields_ = [("dev", Params), ("keyint_max", C.c_int), ("keyint_min", C.c_int), ("scenecut_threshold", C.c_int),
                ("b_adapt", C.c_int), ("b_pyramid", C.c_int), ("rc_lookahead", C.c_int), ("mb_tree", C.c_int),
                ("weightp", C.c_int), ("open_gop", C.c_int), ("frame_refs", C.c_int), ("psy", C.c_int),
                ("rc_is_cqp", C.c_int), ("fps_num", C.c_int), ("fps_den", C.c_int), ("qcompress", C.c_float), ("vbv", C.c_int), ("vfr_input", C.c_int), ("timebase_num", C.c_int),
                ("timebase_den", C.c_int), ("intra_refresh", C.c_int)]


FRAME_PUT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int)
FRAME_STATS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
WEIGHT_COST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(Weight), C.POINTER(C.c_uint))
FRAME_COST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                            C.POINTER(Weight), C.c_int, C.c_int, C.POINTER(Cost))
PREFETCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int)
MBTREE_HOOK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(MbtreeOp), C.c_int)
MBTREE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(MbtreeOp), C.c_int)
QP_OFFSETS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float))
PUT_BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int)


PREFETCH_WEIGHTS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(Weight))


RECALC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int))
ROW_SATDS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int))
ADD_QOFFS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float))
PUT_BATCH_YUV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p),
                                C.POINTER(C.c_void_p), C.c_int)
FRAME_PUT_YUV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int)


GOP_HINT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)
FLUSH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class Backend(C.Structure):
    _fields_ = [("user", C.c_void_p), ("frame_put", FRAME_PUT_FN), ("frame_stats", FRAME_STATS_FN),
                ("weight_cost", WEIGHT_COST_FN), ("frame_cost", FRAME_COST_FN), ("prefetch", PREFETCH_FN),
                ("mbtree", MBTREE_FN), ("get_qp_offsets", QP_OFFSETS_FN), ("frame_put_batch", PUT_BATCH_FN),
                ("prefetch_weight_costs", PREFETCH_WEIGHTS_FN), ("frame_cost_recalculate", RECALC_FN),
                ("get_row_satds", ROW_SATDS_FN), ("frame_put_yuv", FRAME_PUT_YUV_FN),
                ("add_quant_offsets", ADD_QOFFS_FN), ("frame_put_batch_yuv", PUT_BATCH_YUV_FN), ("gop_hint", GOP_HINT_FN), ("flush", FLUSH_FN),
                ("prefetch_weighted_fields", PREFETCH_WEIGHTS_FN)]


class Picture(C.Structure):
    """x264hip_picture"""
    _fields_ = [("planes", C.c_void_p * 3), ("strides", C.c_int * 3), ("is_device", C.c_int), ("i_type", C.c_int), ("i_pts", C.c_int64),
                ("quant_offsets", C.c_void_p)]


LOOKAHEAD_MAX = 250


class LaVbv(C.Structure):
    _fields_ = [("n_planned", C.c_int), ("planned_type", C.c_int * (LOOKAHEAD_MAX + 1)), ("planned_satd", C.c_int * (LOOKAHEAD_MAX + 1)),
                ("dist_p0", C.c_int), ("dist_p1", C.c_int), ("satd", C.c_int)]


class LaFrameOut(C.Structure):
    _fields_ = [("frame", C.c_int), ("type", C.c_int), ("bframes", C.c_int), ("keyframe", C.c_int),
                ("cost_est", (C.c_int * (BFRAME_MAX + 2)) * (BFRAME_MAX + 2)),
                ("cost_est_aq", (C.c_int * (BFRAME_MAX + 2)) * (BFRAME_MAX + 2)),
                ("intra_mbs", C.c_int * (BFRAME_MAX + 2))]


class FrameOutList(list):
    """the outputs of one x264hip_lookahead_run_frames call: a list of LaFrameOut views, and the block they live in as ONE numpy record
    array (.records) -- a caller with eight threads summarising 160 frames each must not spend its time in per-frame Python"""

    def __init__(self, array, count):
        super().__init__(array[i] for i in range(count))
        self.array = array
        n = BFRAME_MAX + 2
        dt = np.dtype([("frame", np.int32), ("type", np.int32), ("bframes", np.int32), ("keyframe", np.int32), ("cost_est", np.int32, (n, n)),
                       ("cost_est_aq", np.int32, (n, n)), ("intra_mbs", np.int32, (n,))])
        assert dt.itemsize == C.sizeof(LaFrameOut)
        self.records = np.frombuffer(array, dtype=dt, count=count) if count else np.zeros(0, dt)


# x264 presets relevant to the lookahead (common/base.c:489-609) and defaults (base.c:344-485)
PRESETS = {
    "medium": dict(),
    "slow": dict(subme=8, rc_lookahead=50, frame_refs=5),
    "slower": dict(subme=9, rc_lookahead=60, b_adapt=2, me="umh", frame_refs=8),
    "veryslow": dict(subme=10, rc_lookahead=60, b_adapt=2, me="umh", me_range=24, bframes=8, frame_refs=16),
    "fast": dict(subme=6, rc_lookahead=30, weightp=1, frame_refs=2),
    "faster": dict(subme=4, rc_lookahead=20, weightp=1, frame_refs=2),
    "veryfast": dict(subme=2, rc_lookahead=10, weightp=1, frame_refs=1),
    "superfast": dict(me="dia", subme=1, rc_lookahead=0, mb_tree=0, weightp=1, frame_refs=1),
    "ultrafast": dict(me="dia", subme=0, rc_lookahead=0, mb_tree=0, weightp=0, frame_refs=1, scenecut=0, bframes=0, b_adapt=0,
                      aq_mode=0, weighted_bipred=0, transform_8x8=0),
    "placebo": dict(subme=11, rc_lookahead=60, b_adapt=2, me="tesa", me_range=24, bframes=16, frame_refs=16),
}
# x264_param_apply_tune (common/base.c:606-700), the fields the lookahead reads; applied after the preset, before overrides
TUNES = {
    "": {}, "film": {}, "touhou": dict(_refs_x2=1, aq_strength=1.3),
    "animation": dict(_refs_x2=1, aq_strength=0.6, _bframes_add=2),
    "grain": dict(aq_strength=0.5, qcompress=0.8),
    "stillimage": dict(aq_strength=1.2),
    "psnr": dict(aq_mode=0, psy=0),
    "ssim": dict(aq_mode=2, psy=0),
    "fastdecode": dict(weighted_bipred=0, weightp=0),
    "zerolatency": dict(rc_lookahead=0, bframes=0, mb_tree=0),
}
_ME = {"dia": 0, "hex": 1, "umh": 2, "esa": 3, "tesa": 4}


# common/tables.c x264_levels (H.264 Table A-1): (level_idc, MB/s, frame MBs, DPB MBs, kbit/s, CPB kbit, mv_range)
LEVELS = [(10, 1485, 99, 396, 64, 175, 64), (9, 1485, 99, 396, 128, 350, 64), (11, 3000, 396, 900, 192, 500, 128),
          (12, 6000, 396, 2376, 384, 1000, 128), (13, 11880, 396, 2376, 768, 2000, 128), (20, 11880, 396, 2376, 2000, 2000, 128),
          (21, 19800, 792, 4752, 4000, 4000, 256), (22, 20250, 1620, 8100, 4000, 4000, 256), (30, 40500, 1620, 8100, 10000, 10000, 256),
          (31, 108000, 3600, 18000, 14000, 14000, 512), (32, 216000, 5120, 20480, 20000, 20000, 512),
          (40, 245760, 8192, 32768, 20000, 25000, 512), (41, 245760, 8192, 32768, 50000, 62500, 512),
          (42, 522240, 8704, 34816, 50000, 62500, 512), (50, 589824, 22080, 110400, 135000, 135000, 512),
          (51, 983040, 36864, 184320, 240000, 240000, 512), (52, 2073600, 36864, 184320, 240000, 240000, 512),
          (60, 4177920, 139264, 696320, 240000, 240000, 8192), (61, 8355840, 139264, 696320, 480000, 480000, 8192),
          (62, 16711680, 139264, 696320, 800000, 800000, 8192)]


def mv_range_for(width, height, fps_num=25, fps_den=1, bit_depth=8, frame_refs=3, bframes=3, b_pyramid=2, keyint_max=250,
                 transform_8x8=1, bitrate=0, vbv_maxrate=0, vbv_bufsize=0, chroma_format=1):
    """param.analyse.i_mv_range of the automatically chosen level (encoder.c:1243-1268): the first level of the table that
    x264_validate_levels (encoder/set.c:876-913) accepts -- frame size, decoded picture buffer, VBV rate and buffer against the
    profile's limits, macroblock rate -- or the last one."""
    mb_w, mb_h = (width + 15) // 16, (height + 15) // 16
    mbs = mb_w * mb_h
    # x264_sps_init (encoder/set.c:114-157)
    cbp_factor = 16 if chroma_format >= 2 else 12 if bit_depth > 8 else 5 if transform_8x8 else 4
    reorder = 2 if b_pyramid else 1 if bframes else 0
    dec_buffering = 0 if keyint_max == 1 else min(16, max(frame_refs, 1 + reorder, 4 if b_pyramid else 1, 1))
    if bitrate and not vbv_bufsize:  # encoder.c:1248-1249: ABR without VBV is checked as if maxrate were twice the bitrate
        vbv_maxrate = bitrate * 2
    for idc, mbps, frame_size, dpb, kbps, cpb, mvr in LEVELS:
        if (frame_size >= mbs and frame_size * 8 >= mb_w * mb_w and frame_size * 8 >= mb_h * mb_h and mbs * dec_buffering <= dpb and
                vbv_maxrate <= kbps * cbp_factor // 4 and vbv_bufsize <= cpb * cbp_factor // 4 and
                (fps_den <= 0 or mbs * fps_num // fps_den <= mbps)):
            return mvr
    return LEVELS[-1][6]


def la_config(width, height, preset="medium", bit_depth=8, tune="", **over):
    """Effective lookahead configuration for an x264 preset (+ overrides), as validate_parameters and
    lowres_context_init derive it (encoder/encoder.c:423-1407, encoder/slicetype.c:45-61)."""
    c = dict(bframes=3, b_adapt=1, b_pyramid=2, rc_lookahead=40, me="hex", me_range=16, subme=7, weightp=2,
             weighted_bipred=1, mb_tree=1, aq_mode=1, aq_strength=1.0, scenecut=40, keyint_max=250, keyint_min=0,
             open_gop=0, frame_refs=3, psy=1, rc_is_cqp=0, bframe_bias=0, fps=25.0, mv_range=0, fps_num=25, fps_den=1,
             qcompress=0.6, threads=1, lookahead_threads=0, bitrate=0, vbv_maxrate=0, vbv_bufsize=0, transform_8x8=1, intra_refresh=0, vfr_input=0, timebase_num=0, timebase_den=0, chroma_format=1)
    c.update(PRESETS[preset])
    for t in filter(None, tune.replace(",", " ").split()):
        tv = dict(TUNES[t])
        if tv.pop("_refs_x2", 0) and c["frame_refs"] > 1:
            c["frame_refs"] *= 2
        c["bframes"] += tv.pop("_bframes_add", 0)
        c.update(tv)
    c.update(over)
    clip = lambda v, lo, hi: lo if v < lo else hi if v > hi else v  # noqa: E731
    # the order below is validate_parameters' own (encoder/encoder.c, line numbers in the comments)
    c["keyint_max"] = clip(c["keyint_max"], 1, 1 << 30)                      # :611 (1<<30 = X264_KEYINT_MAX_INFINITE)
    if c["keyint_max"] == 1:                                                 # :612-618
        c["weightp"] = 0
        c["frame_refs"] = 1
        c["intra_refresh"] = 0
    c["subme"] = clip(c["subme"], 0, 11)                                     # :922
    if c["rc_is_cqp"]:                                                       # :951-966
        c["aq_mode"] = 0
        c["mb_tree"] = 0
    c["bitrate"] = clip(c["bitrate"], 0, 2000000)                             # :973-1009, rc method ABR when a bitrate is given
    c["vbv_bufsize"] = clip(c["vbv_bufsize"], 0, 2000000)
    c["vbv_maxrate"] = clip(c["vbv_maxrate"], 0, 2000000)
    if c["vbv_bufsize"]:
        if c["rc_is_cqp"]:
            c["vbv_maxrate"] = c["vbv_bufsize"] = 0
        elif not c["vbv_maxrate"]:
            if c["bitrate"]:
                c["vbv_maxrate"] = c["bitrate"]
            else:
                c["vbv_bufsize"] = 0
        elif c["bitrate"] and c["vbv_maxrate"] < c["bitrate"]:
            c["bitrate"] = c["vbv_maxrate"]
    elif c["vbv_maxrate"]:
        c["vbv_maxrate"] = 0
    c["vbv"] = int(c["vbv_bufsize"] > 0)
    c["frame_refs"] = clip(c["frame_refs"], 1, 16)                           # :1064
    c["scenecut"] = max(c["scenecut"], 0)                                    # :1066-1067
    c["bframes"] = clip(c["bframes"], 0, min(16, c["keyint_max"] - 1))       # :1074
    c["bframe_bias"] = clip(c["bframe_bias"], -90, 100)                      # :1075
    if c["bframes"] <= 1:                                                    # :1076-1077
        c["b_pyramid"] = 0
    c["b_pyramid"] = clip(c["b_pyramid"], 0, 2)
    c["b_adapt"] = clip(c["b_adapt"], 0, 2)
    if not c["bframes"]:                                                     # :1080-1086
        c["b_adapt"] = 0
        c["weighted_bipred"] = 0
        c["open_gop"] = 0
    if c["intra_refresh"]:                                                   # :1087-1102
        if c["b_pyramid"] == 2:
            c["b_pyramid"] = 1
        c["frame_refs"] = 1
        c["open_gop"] = 0
    if "fps" in over and "fps_num" not in over:                              # a plain frame rate: x264_param_parse's "fps" (base.c:1056-1065)
        c["fps_num"], c["fps_den"] = int(round(float(c["fps"]) * 1000)), 1000
    c["fps"] = float(np.float32(c["fps_num"]) / np.float32(c["fps_den"]))     # :1107 float fps
    if c["keyint_min"] <= 0:                                                 # :1109-1111 (0 = X264_KEYINT_MIN_AUTO)
        c["keyint_min"] = min(c["keyint_max"] // 10, int(c["fps"]))
    c["keyint_min"] = clip(c["keyint_min"], 1, c["keyint_max"] // 2 + 1)
    maxrate = max(c["vbv_maxrate"], c["bitrate"])                             # :1112-1117 (float arithmetic as there)
    bufsize = np.float32(c["vbv_bufsize"]) / np.float32(maxrate) if maxrate else np.float32(0)
    c["rc_lookahead"] = int(min(np.float32(clip(c["rc_lookahead"], 0, 250)), max(np.float32(c["keyint_max"]), bufsize * np.float32(c["fps_num"] / c["fps_den"]))))
    c["qcompress"] = clip(c["qcompress"], 0.0, 1.0)                          # :1125
    if c["keyint_max"] == 1 or c["qcompress"] == 1:                          # :1126-1127
        c["mb_tree"] = 0
    if not c["intra_refresh"] and c["keyint_max"] != 1 << 30 and not c["rc_lookahead"] and c["mb_tree"]:  # :1128-1133
        c["mb_tree"] = 0
    me = _ME.get(c["me"], 1)                                                 # :1156-1164
    c["me_range"] = clip(c["me_range"], 4, 1024)
    if c["me_range"] > 16 and me <= 1:
        c["me_range"] = 16
    if me == 4 and c["subme"] <= 1:
        me = 3
    c["aq_mode"] = clip(c["aq_mode"], 0, 3)                                  # :1177-1180
    c["aq_strength"] = clip(float(c["aq_strength"]), 0.0, 3.0)
    if c["aq_strength"] == 0:
        c["aq_mode"] = 0
    if not c["aq_mode"] and c["mb_tree"]:                                    # :1233-1237: MB-tree needs the AQ arrays
        c["aq_mode"] = 1
        c["aq_strength"] = 0.0
    if c["mv_range"] <= 0:                                                   # :1243-1268
        c["mv_range"] = mv_range_for(width, height, c["fps_num"], c["fps_den"], bit_depth, c["frame_refs"], c["bframes"],
                                     c["b_pyramid"], c["keyint_max"], c["transform_8x8"],
                                     0 if c["rc_is_cqp"] else c["bitrate"], c["vbv_maxrate"], c["vbv_bufsize"], c["chroma_format"])
    else:
        c["mv_range"] = clip(c["mv_range"], 32, 8192)
    c["weightp"] = clip(c["weightp"], 0, 2)                                  # :1271
    if not c["weightp"] and c["mb_tree"] and c["psy"]:
        c["weightp"] = -1  # X264_WEIGHTP_FAKE (:1316-1317): the lookahead still analyses and applies weights
    c["psy"] = int(bool(c["psy"]))
    c["mb_tree"] = int(bool(c["mb_tree"]))
    # lookahead bands (:567-583, :1273-1300)
    mb_h = (height + 15) // 16
    max_sliced = max(1, mb_h // 4)
    c["threads"] = clip(c["threads"], 1, 128)
    if c["threads"] == 1:
        c["lookahead_threads"] = 1
    elif c["lookahead_threads"] <= 0:  # X264_THREADS_AUTO, frame threads
        div = [[[6, 6, 6, 6], [3, 3, 3, 3], [4, 4, 4, 4], [6, 6, 6, 6], [12, 12, 12, 12]],
               [[3, 2, 1, 1], [2, 1, 1, 1], [4, 3, 2, 1], [6, 4, 3, 2], [12, 9, 6, 4]]]
        q_subme = min(c["subme"] // 3, 3) + int(c["subme"] > 1)
        q_b = min(int((c["bframes"] - 1) / 3), 3)  # C division truncates toward zero
        c["lookahead_threads"] = min(c["threads"] // div[int(c["b_adapt"] == 2)][q_subme][q_b], height // 128)
    c["lookahead_threads"] = clip(c["lookahead_threads"], 1, min(max_sliced, 16))
    # slicetype.c:823 (no VBV): whether the evaluations visit the outermost ring of blocks
    c["do_edges"] = int(c["mb_tree"] or c["vbv"] or (width + 15) // 16 <= 2 or mb_h <= 2)
    # lowres_context_init (slicetype.c:45-61) and mbcmp_init (encoder.c:1409-1427)
    if c["subme"] > 1:
        c["la_me_method"] = min(1, me)
        c["la_subpel_refine"] = 4
    else:
        c["la_me_method"] = 0
        c["la_subpel_refine"] = 2
    c["mbcmp_satd"] = int(c["subme"] > 1)
    c["fpelcmp_satd"] = int(me == 4 and c["subme"] > 1)
    c["width"], c["height"], c["bit_depth"] = width, height, bit_depth
    c["lam"] = 1 if bit_depth == 8 else 4
    return c


def make_la_params(cfg, cost_mv=None, max_frames=0):
    if cost_mv is None:
        cost_mv, centre = cost_mv_table(cfg["mv_range"], cfg["lam"])
    else:
        cost_mv = np.ascontiguousarray(cost_mv, np.uint16)
        centre = (cost_mv.size - 1) // 2
    dev = Params(cfg["bit_depth"], cfg["width"], cfg["height"], cfg["bframes"], cfg["lam"], cfg["la_me_method"],
                 cfg["la_subpel_refine"], cfg["me_range"], cfg["mv_range"], cfg["subme"], cfg["mbcmp_satd"],
                 cfg["fpelcmp_satd"], cfg["weighted_bipred"], cfg["aq_mode"], cfg["aq_strength"], cfg["bframe_bias"],
                 max_frames, int(not cfg["do_edges"]), cfg["lookahead_threads"], cfg["chroma_format"], cost_mv.ctypes.data + 2 * centre)
    p = LaParams(dev, cfg["keyint_max"], cfg["keyint_min"], cfg["scenecut"], cfg["b_adapt"], cfg["b_pyramid"],
                 cfg["rc_lookahead"], cfg["mb_tree"], cfg["weightp"], cfg["open_gop"], cfg["frame_refs"], cfg["psy"],
                 cfg["rc_is_cqp"], cfg["fps_num"], cfg["fps_den"], cfg["qcompress"], cfg["vbv"], cfg["vfr_input"], cfg["timebase_num"], cfg["timebase_den"], cfg["intra_refresh"])
    p._keep = cost_mv
    return p


def lookahead_classes(cfg):
    """x264hip_lookahead_classes: (cell_allowed[ns][ns] uint8, mask_l0, mask_l1) -- the cell / field classes the decisions of a lookahead
    with this configuration can ever ask for (pure host arithmetic, no device)"""
    L = load()
    p = make_la_params(cfg)
    ns = cfg["bframes"] + 2
    ok = np.zeros((ns, ns), np.uint8)
    m0, m1 = C.c_uint(0), C.c_uint(0)
    _ck(L.x264hip_lookahead_classes(C.byref(p), _p(ok), C.byref(m0), C.byref(m1)), "lookahead_classes")
    return ok, m0.value, m1.value


class Lookahead:
    """x264hip_lookahead: put frames in display order, get frames back in coded order with their types."""

    def __init__(self, cfg, device=0, backend=None, cost_mv=None, max_frames=0, prefetch_hook=None, mbtree_hook=None):
        """prefetch_hook( slots, frame_numbers ): device lookahead whose speculative submissions go through the caller
        (x264hip_lookahead_open_hooked; x264_amd/shard.py spreads them over several GPUs).
        mbtree_hook( cells ): called with the (slot_b, slot_p0, slot_p1, d0, d1) of the PROPAGATE steps right before every MB-tree call
        (x264hip_lookahead_set_mbtree_hook: the window shard fetches the per-block maps those steps read)."""
        L = load()
        self.L = L
        self.cfg = cfg
        self.params = make_la_params(cfg, cost_mv, max_frames)
        self.h = C.c_void_p()
        if prefetch_hook is not None:
            def _hook(user, slots, numbers, n):
                try:
                    prefetch_hook([slots[i] for i in range(n)], [numbers[i] for i in range(n)])
                    return 0
                except Exception as e:  # never let an exception cross the C ABI
                    import traceback
                    traceback.print_exc()
                    self._hook_error = e
                    return -4
            self._hook = PREFETCH_FN(_hook)
            L.x264hip_lookahead_open_hooked.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(LaParams), PREFETCH_FN, C.c_void_p]
            _ck(L.x264hip_lookahead_open_hooked(C.byref(self.h), device, C.byref(self.params), self._hook, None), "x264hip_lookahead_open_hooked")
        elif backend is None:
            L.x264hip_lookahead_open.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(LaParams)]
            _ck(L.x264hip_lookahead_open(C.byref(self.h), device, C.byref(self.params)), "x264hip_lookahead_open")
        else:
            self._backend = backend
            L.x264hip_lookahead_open_backend.argtypes = [C.POINTER(C.c_void_p), C.POINTER(LaParams), C.POINTER(Backend)]
            _ck(L.x264hip_lookahead_open_backend(C.byref(self.h), C.byref(self.params), C.byref(backend)),
                "x264hip_lookahead_open_backend")
        if mbtree_hook is not None:
            def _mhook(user, ops, n):
                try:
                    mbtree_hook([(ops[i].slot_b, ops[i].slot_p0, ops[i].slot_p1, ops[i].dist_p0, ops[i].dist_p1) for i in range(n) if ops[i].type == 1])
                    return 0
                except Exception as e:  # never let an exception cross the C ABI
                    import traceback
                    traceback.print_exc()
                    self._hook_error = e
                    return -4
            self._mhook = MBTREE_HOOK_FN(_mhook)
            L.x264hip_lookahead_set_mbtree_hook.argtypes = [C.c_void_p, MBTREE_HOOK_FN, C.c_void_p]
            _ck(L.x264hip_lookahead_set_mbtree_hook(self.h, self._mhook, None), "x264hip_lookahead_set_mbtree_hook")
        self.dtype = np.uint8 if cfg["bit_depth"] == 8 else np.uint16
        L.x264hip_lookahead_ctx.restype = C.c_void_p
        self.delay = L.x264hip_lookahead_delay(self.h)
        self._n_put = 0

    def close(self):
        if self.h:
            self.L.x264hip_lookahead_close(self.h)
            self.h = None

    def set_chunk(self, frames):
        _ck(self.L.x264hip_lookahead_set_chunk(self.h, int(frames)), "lookahead_set_chunk")

    def reset(self):
        _ck(self.L.x264hip_lookahead_reset(self.h), "lookahead_reset")
        self._n_put = 0

    def ctx_handle(self):
        return C.c_void_p(self.L.x264hip_lookahead_ctx(self.h))

    def put_pic(self, y, cb=None, cr=None, forced_type=0, pts=None, quant_offsets=None):
        """x264hip_lookahead_put: host planes (cb / cr optional) plus the picture's quant_offsets (float per macroblock)"""
        pic = Picture()
        keep = [np.ascontiguousarray(y, self.dtype)]
        pic.planes[0], pic.strides[0] = keep[0].ctypes.data, keep[0].shape[1]
        if cb is not None:
            keep += [np.ascontiguousarray(cb, self.dtype), np.ascontiguousarray(cr, self.dtype)]
            pic.planes[1], pic.planes[2] = keep[1].ctypes.data, keep[2].ctypes.data
            pic.strides[1] = pic.strides[2] = keep[1].shape[1]
        pic.i_type, pic.i_pts = forced_type, int(pts) if pts is not None else self._n_put
        if quant_offsets is not None:
            keep.append(np.ascontiguousarray(quant_offsets, np.float32))
            pic.quant_offsets = keep[-1].ctypes.data
        self.L.x264hip_lookahead_put.argtypes = [C.c_void_p, C.c_void_p]
        _ck(self.L.x264hip_lookahead_put(self.h, C.byref(pic)), "lookahead_put")
        self._n_put += 1

    def put_picture(self, y, cb, cr, forced_type=0, pts=None, device=False, strides=None):
        """the whole 4:2:0 picture (x264hip_lookahead_put_picture): numpy planes, or device addresses with strides=(ys, cs)"""
        self.L.x264hip_lookahead_put_picture.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64]
        if device:
            ptrs, st = (C.c_void_p * 3)(y, cb, cr), (C.c_int * 3)(strides[0], strides[1], strides[1])
        else:
            y, cb, cr = (np.ascontiguousarray(a, self.dtype) for a in (y, cb, cr))
            ptrs = (C.c_void_p * 3)(y.ctypes.data, cb.ctypes.data, cr.ctypes.data)
            st = (C.c_int * 3)(y.shape[1], cb.shape[1], cr.shape[1])
            self._keep_pic = (y, cb, cr)
        ts = int(pts) if pts is not None else self._n_put
        _ck(self.L.x264hip_lookahead_put_picture(self.h, ptrs, st, int(device), forced_type, ts), "lookahead_put_picture")
        self._n_put += 1

    def put(self, luma=None, device_ptr=None, stride=None, forced_type=0, pts=None):
        self._n_put += 1
        if pts is not None:
            self.L.x264hip_lookahead_put_frame_pts.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64]
            if device_ptr is not None:
                src, st, dev = C.c_void_p(device_ptr), stride or self.cfg["width"], 1
            else:
                luma = np.ascontiguousarray(luma, self.dtype)
                src, st, dev = _p(luma), luma.shape[1], 0
            _ck(self.L.x264hip_lookahead_put_frame_pts(self.h, src, st, dev, forced_type, int(pts)), "lookahead_put_frame_pts")
            return
        if device_ptr is not None:
            _ck(self.L.x264hip_lookahead_put_frame(self.h, C.c_void_p(device_ptr), stride or self.cfg["width"], 1,
                                                   forced_type), "lookahead_put_frame")
            return
        luma = np.ascontiguousarray(luma, self.dtype)
        _ck(self.L.x264hip_lookahead_put_frame(self.h, _p(luma), luma.shape[1], 0, forced_type), "lookahead_put_frame")

    def put_pictures(self, y_ptrs, stride, cb_ptrs=None, cr_ptrs=None, cstride=0, types=None, pts=None):
        """x264hip_lookahead_put_pictures: device-resident pictures in one call"""
        n = len(y_ptrs)
        arr = lambda p: (C.c_void_p * n)(*p) if p is not None else None  # noqa: E731
        ty = (C.c_int * n)(*[int(t) for t in types]) if types is not None else None
        ts = (C.c_int64 * n)(*[int(t) for t in pts]) if pts is not None else None
        self.L.x264hip_lookahead_put_pictures.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                          C.c_void_p]
        _ck(self.L.x264hip_lookahead_put_pictures(self.h, n, arr(y_ptrs), stride, arr(cb_ptrs), arr(cr_ptrs), cstride, ty, ts),
            "lookahead_put_pictures")
        self._n_put += n

    def put_batch(self, device_ptrs, stride=None):
        arr = (C.c_void_p * len(device_ptrs))(*device_ptrs)
        _ck(self.L.x264hip_lookahead_put_frames(self.h, len(device_ptrs), arr, stride or self.cfg["width"]), "lookahead_put_frames")

    def get(self, flush=False, qp_offsets=False, vbv=False):
        """vbv=True also returns what VBV rate control reads: out.planned (list of (type, satd)), out.own_cell, out.row_satds,
        out.row_satds_intra (x264hip_lookahead_get_frame_vbv)."""
        out = LaFrameOut()
        got = C.c_int(0)
        mb_h = (self.cfg["height"] + 15) // 16
        qp = np.zeros(((self.cfg["width"] + 15) // 16) * mb_h, np.float32) if qp_offsets else None
        if vbv:
            v = LaVbv()
            rows, rows_i = np.full(mb_h, -1, np.int32), np.full(mb_h, -1, np.int32)
            _ck(self.L.x264hip_lookahead_get_frame_vbv(self.h, int(flush), C.byref(out), C.byref(got), _p(qp), C.byref(v), _p(rows),
                                                       _p(rows_i)), "lookahead_get_frame_vbv")
            if got.value:
                out.planned = [(v.planned_type[i], v.planned_satd[i]) for i in range(v.n_planned)]
                out.own_cell = (v.dist_p0, v.dist_p1)
                out.rc_satd = v.satd
                out.row_satds, out.row_satds_intra = rows, rows_i
        else:
            _ck(self.L.x264hip_lookahead_get_frame_ex(self.h, int(flush), C.byref(out), C.byref(got), _p(qp)), "lookahead_get_frame")
        if got.value and qp_offsets:
            out.qp_offset = qp
        return out if got.value else None

    def stats(self):
        out = np.zeros(8, np.uint64)
        _ck(self.L.x264hip_lookahead_stats(self.h, _p(out), 8), "lookahead_stats")
        return out

    def class_requests(self):
        """(field_req[2][bframes+1], cell_req[ns][ns], cell_allowed[ns][ns], field_allowed[2]) of the lookahead's device context
        (x264hip_class_requests): the requests per class so far next to the statement x264hip_lookahead_open made about its flow"""
        bf = self.cfg["bframes"]
        ns = bf + 2
        fr, cr = np.zeros((2, bf + 1), np.uint32), np.zeros((ns, ns), np.uint32)
        ca, fa = np.zeros((ns, ns), np.uint8), np.zeros(2, np.uint32)
        _ck(self.L.x264hip_class_requests(self.ctx_handle(), _p(fr), _p(cr), _p(ca), _p(fa)), "class_requests")
        return fr, cr, ca, fa

    def run_frames(self, device_ptrs, stride=None, paced=True):
        """run() for device-resident luma frames with nothing but types and costs asked for, as ONE call (x264hip_lookahead_run_frames)"""
        n = len(device_ptrs)
        arr = (C.c_void_p * n)(*device_ptrs)
        outs = (LaFrameOut * n)()
        got = C.c_int(0)
        _ck(self.L.x264hip_lookahead_run_frames(self.h, n, arr, stride or self.cfg["width"], int(paced), outs, C.byref(got)), "lookahead_run_frames")
        return FrameOutList(outs, got.value)

    def run(self, frames=None, device_ptrs=None, stride=None, paced=True, qp_offsets=False, forced_types=None, vbv=False, pts=None, chroma=None, quant_offsets=None):
        """Feed a whole clip.  paced=True interleaves put/get exactly like x264_encoder_encode; paced=False puts
        every frame first (deep prefetch) -- results are identical, only the batching differs."""
        outs = []
        n = len(frames) if frames is not None else len(device_ptrs)
        if not paced and frames is None:
            self.put_batch(device_ptrs, stride)  # batch ingest: one launch per ingest kernel
            n_loop = 0
        else:
            n_loop = n
        for i in range(n_loop):
            ft = int(forced_types[i]) if forced_types is not None else 0
            ts = None if pts is None else int(pts[i])
            if quant_offsets is not None:
                self.put_pic(frames[i], None if chroma is None else chroma[0][i], None if chroma is None else chroma[1][i], forced_type=ft, pts=ts,
                             quant_offsets=quant_offsets[i])
            elif chroma is not None:
                self.put_picture(frames[i], chroma[0][i], chroma[1][i], forced_type=ft, pts=ts)
            elif frames is not None:
                self.put(frames[i], forced_type=ft, pts=ts)
            else:
                self.put(device_ptr=device_ptrs[i], stride=stride, forced_type=ft, pts=ts)
            if paced:
                o = self.get(False, qp_offsets, vbv)
                if o is not None:
                    outs.append(o)
        while len(outs) < n:
            o = self.get(True, qp_offsets, vbv)
            if o is None:
                break
            outs.append(o)
        return outs
