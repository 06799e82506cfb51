"""checkasm-style differential test of the oracle's vtable primitives against the reference C
functions (tools/checkasm.c:361-888 check_pixel, :890-1224 check_dct, :1226-1956 check_mc,
:2071-2517 check_quant): seeded random inputs plus max-difference patterns."""
import ctypes as C

import numpy as np
import pytest

from oracle import refharness
from oracle.oraclelib import Oracle, Weight

pytestmark = pytest.mark.skipif(not refharness.available(8), reason="oracle/_ref not built (no /root/reference)")

SIZES = [(16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)]


def _ptr(a, off=0):
    return C.c_void_p(a.ctypes.data + off * a.itemsize)


def _patterns(rng, shape, dtype, maxv):
    yield rng.integers(0, maxv + 1, size=shape).astype(dtype)
    yield rng.integers(0, 2, size=shape).astype(dtype) * maxv          # max-difference pattern
    a = np.zeros(shape, dtype); a[::2, ::2] = maxv; a[1::2, 1::2] = maxv   # checkerboard
    yield a
    yield np.full(shape, maxv, dtype)


@pytest.fixture(scope="module", params=[8, 10])
def env(request):
    d = request.param
    if not refharness.available(d):
        pytest.skip("no ref for depth %d" % d)
    r = refharness.Ref(64, 64, "medium", bit_depth=d)
    yield r, Oracle(d), d
    r.close()


def test_pixel_metrics(env):
    r, o, d = env
    rng = np.random.default_rng(1)
    maxv = (1 << d) - 1
    L = r.lib
    L.rh_pixel_cmp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long]
    for pa in _patterns(rng, (48, 64), o.dtype, maxv):
        for pb in _patterns(rng, (48, 64), o.dtype, maxv):
            for si, (w, h) in enumerate(SIZES):
                for (ox, oy) in [(0, 0), (3, 5), (17, 1)]:
                    a = _ptr(pa, 0)  # first arg aligned
                    b = _ptr(pb, oy * 64 + ox)
                    assert L.rh_pixel_cmp(r.ctx, 0, si, a, 64, b, 64) == o.f("sad", C.c_int)(a, 64, b, 64, w, h)
                    assert L.rh_pixel_cmp(r.ctx, 1, si, a, 64, b, 64) == o.f("satd", C.c_int)(a, 64, b, 64, w, h)
                    assert L.rh_pixel_cmp(r.ctx, 2, si, a, 64, b, 64) == o.f("ssd", C.c_int)(a, 64, b, 64, w, h)
                    if (w, h) in ((16, 16), (8, 8)):
                        sa = 0 if w == 16 else 3
                        assert L.rh_pixel_cmp(r.ctx, 3, sa, a, 64, b, 64) == o.f("sa8d", C.c_int)(a, 64, b, 64, w)
    L.rh_pixel_var.restype = C.c_uint64
    L.rh_pixel_var.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
    for pa in _patterns(rng, (48, 64), o.dtype, maxv):
        for si, (w, h) in [(0, (16, 16)), (2, (8, 16)), (3, (8, 8))]:
            assert L.rh_pixel_var(r.ctx, si, _ptr(pa), 64) == o.f("var", C.c_uint64)(_ptr(pa), 64, w, h)


def test_sad_satd_xn(env):
    r, o, d = env
    rng = np.random.default_rng(2)
    maxv = (1 << d) - 1
    fenc = rng.integers(0, maxv + 1, size=(16, 16)).astype(o.dtype)
    ref = rng.integers(0, maxv + 1, size=(64, 64)).astype(o.dtype)
    L = r.lib
    L.rh_pixel_cmp_xn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
    for satd in (0, 1):
        for n in (3, 4):
            for si, (w, h) in enumerate(SIZES):
                offs = np.array([20 * 64 + 20, 19 * 64 + 21, 22 * 64 + 17, 20 * 64 + 23], np.int32)
                out = np.zeros(4, np.int32)
                L.rh_pixel_cmp_xn(r.ctx, satd, n, si, _ptr(fenc), _ptr(ref), _ptr(offs), 64, _ptr(out))
                fn = o.f("satd" if satd else "sad", C.c_int)
                for k in range(n):
                    assert out[k] == fn(_ptr(fenc), 16, _ptr(ref, int(offs[k])), 64, w, h)


def test_intra_predictors(env):
    r, o, d = env
    rng = np.random.default_rng(3)
    maxv = (1 << d) - 1
    L = r.lib
    for it, buf0 in enumerate(_patterns(rng, (12, 32), o.dtype, maxv)):
        org = 1 * 32 + 8  # pixel (0,0): row 1, col 8 of an FDEC_STRIDE buffer
        for mode in range(4):
            a, b = buf0.copy(), buf0.copy()
            L.rh_predict_8x8c(r.ctx, mode if mode < 3 else 3, _ptr(a, org))
            # reference mode numbering (predict.h): I_PRED_CHROMA_DC=0, H=1, V=2, P=3
            o.f("predict_8x8c")(mode, _ptr(b, org))
            assert np.array_equal(a, b), ("8x8c", mode, it)
        ea = np.zeros(40, o.dtype); eb = np.zeros(40, o.dtype)
        L.rh_predict_8x8_filter(r.ctx, _ptr(buf0.copy(), org), _ptr(ea), 0xF, 0xF)  # ALL_NEIGHBORS
        o.f("predict_8x8_filter")(_ptr(buf0, org), _ptr(eb))
        assert np.array_equal(ea[6:33], eb[6:33]), ("filter", it)
        for mode in range(3, 9):
            a, b = buf0.copy(), buf0.copy()
            L.rh_predict_8x8(r.ctx, mode, _ptr(a, org), _ptr(ea))
            o.f("predict_8x8")(mode, _ptr(b, org), _ptr(eb))
            assert np.array_equal(a, b), ("8x8", mode, it)
        fenc = rng.integers(0, maxv + 1, size=(8, 16)).astype(o.dtype)
        for satd in (0, 1):
            ra = np.zeros(3, np.int32); rb = np.zeros(3, np.int32)
            L.rh_intra_x3_8x8c(r.ctx, satd, _ptr(fenc), _ptr(buf0.copy(), org), _ptr(ra))
            o.f("intra_x3_8x8c")(satd, _ptr(fenc), _ptr(buf0.copy(), org), _ptr(rb))
            assert np.array_equal(ra, rb)


def test_mc(env):
    r, o, d = env
    rng = np.random.default_rng(4)
    maxv = (1 << d) - 1
    planes = rng.integers(0, maxv + 1, size=(4, 64, 64)).astype(o.dtype)
    L = r.lib
    L.rh_mc_luma.argtypes = [C.c_void_p, C.c_void_p, C.c_long] + [C.c_void_p] * 4 + [C.c_long] + [C.c_int] * 8
    org = 24 * 64 + 24
    pp = (C.c_void_p * 4)(*[planes[i].ctypes.data + org * planes.itemsize for i in range(4)])
    for wt in [None, Weight(1, 55, 6, 3), Weight(1, 100, 5, -20), Weight(1, 3, 0, -1), Weight(1, 127, 7, 127)]:
        for mvy in range(-9, 10):
            for mvx in range(-9, 10):
                for (w, h) in ((8, 8), (8, 9), (16, 16), (4, 4)):
                    a = np.zeros((20, 32), o.dtype); b = np.zeros((20, 32), o.dtype)
                    wa = (wt.on, wt.scale, wt.denom, wt.offset) if wt else (0, 1, 0, 0)
                    L.rh_mc_luma(r.ctx, _ptr(a), 32, pp[0], pp[1], pp[2], pp[3], 64, mvx, mvy, w, h, *wa)
                    o.f("mc_luma")(_ptr(b), 32, pp, 64, mvx, mvy, w, h, C.byref(wt) if wt else None)
                    assert np.array_equal(a, b), (mvx, mvy, w, h, wa)
    L.rh_avg.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int]
    for si, (w, h) in enumerate(SIZES):
        for weight in (32, 1, 17, 63, -10, 70):
            a = np.zeros((20, 32), o.dtype); b = np.zeros((20, 32), o.dtype)
            L.rh_avg(r.ctx, si, _ptr(a), 32, _ptr(planes[0]), 64, _ptr(planes[1], 5), 64, weight)
            o.f("avg")(_ptr(b), 32, _ptr(planes[0]), 64, _ptr(planes[1], 5), 64, w, h, weight)
            assert np.array_equal(a, b), (w, h, weight)


def test_lowres_core(env):
    r, o, d = env
    rng = np.random.default_rng(5)
    maxv = (1 << d) - 1
    L = r.lib
    L.rh_lowres_core.argtypes = [C.c_void_p] * 6 + [C.c_long, C.c_long, C.c_int, C.c_int]
    for w in range(96, 121, 8):  # checkasm.c:1715-1744 uses widths 96..120 step 8
        h = 8
        src = rng.integers(0, maxv + 1, size=(2 * h + 2, 2 * w + 16)).astype(o.dtype)
        a = np.zeros((4, h, w), o.dtype); b = np.zeros((4, h, w), o.dtype)
        L.rh_lowres_core(r.ctx, _ptr(src), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), src.shape[1], w, w, h)
        o.f("lowres_core")(_ptr(src), _ptr(b[0]), _ptr(b[1]), _ptr(b[2]), _ptr(b[3]), src.shape[1], w, w, h)
        assert np.array_equal(a, b)


def test_hpel_filter(env):
    """oracle vs h->mc.hpel_filter incl. the five extra dstv columns (common/mc.c:172-196)"""
    r, o, d = env
    rng = np.random.default_rng(8)
    maxv = (1 << d) - 1
    L = r.lib
    L.rh_hpel_filter.argtypes = [C.c_void_p] * 5 + [C.c_long, C.c_int, C.c_int, C.c_void_p]
    f = o.f("hpel_filter")
    f.argtypes = [C.c_void_p] * 4 + [C.c_long, C.c_int, C.c_int, C.c_void_p]
    for w, h, kind in [(64, 16, "r"), (100, 9, "r"), (37, 5, "x"), (128, 32, "r"), (48, 48, "x")]:
        stride = w + 32
        src = rng.integers(0, maxv + 1, size=(h + 8, stride)).astype(o.dtype)
        if kind == "x":
            src[:] = np.where((np.indices(src.shape).sum(0) & 1) == 0, 0, maxv).astype(o.dtype)
        off = 3 * stride + 8
        a = [np.full((h + 8, stride), 7, o.dtype) for _ in range(3)]
        b = [np.full((h + 8, stride), 7, o.dtype) for _ in range(3)]
        buf = np.zeros(w + 64, np.int16)
        L.rh_hpel_filter(r.ctx, _ptr(a[0], off), _ptr(a[1], off), _ptr(a[2], off), _ptr(src, off), stride, w, h, _ptr(buf))
        f(_ptr(b[0], off), _ptr(b[1], off), _ptr(b[2], off), _ptr(src, off), stride, w, h, _ptr(buf))
        for k in range(3):
            assert np.array_equal(a[k], b[k]), (w, h, kind, k)


def test_var2_hadamard_ac_vsad_asd8(env):
    """the remaining P8 metrics: oracle vs pixf.var2 / hadamard_ac / vsad / asd8 on random and extreme data"""
    r, o, d = env
    rng = np.random.default_rng(12)
    maxv = (1 << d) - 1
    L = r.lib
    L.rh_var2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rh_hadamard_ac.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
    L.rh_hadamard_ac.restype = C.c_uint64
    L.rh_vsad.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int]
    L.rh_asd8.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int]
    f_var2 = o.f("var2", C.c_int); f_var2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    f_hac = o.f("hadamard_ac", C.c_uint64); f_hac.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    f_vsad = o.f("vsad", C.c_int); f_vsad.argtypes = [C.c_void_p, C.c_long, C.c_int]
    f_asd8 = o.f("asd8", C.c_int); f_asd8.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int]
    for trial in range(12):
        if trial == 0:
            fenc = np.full((16, 16), maxv, o.dtype); fdec = np.zeros((16, 32), o.dtype)
        elif trial == 1:
            yy, xx = np.mgrid[0:16, 0:16]
            fenc = (((yy ^ xx) & 1) * maxv).astype(o.dtype); fdec = np.full((16, 32), maxv // 2, o.dtype)
        else:
            fenc = rng.integers(0, maxv + 1, size=(16, 16)).astype(o.dtype); fdec = rng.integers(0, maxv + 1, size=(16, 32)).astype(o.dtype)
        for is16 in (0, 1):
            sa, sb = np.zeros(2, np.int32), np.zeros(2, np.int32)
            assert L.rh_var2(r.ctx, is16, _ptr(fenc), _ptr(fdec), _ptr(sa)) == f_var2(_ptr(fenc), _ptr(fdec), 16 if is16 else 8, _ptr(sb))
            assert np.array_equal(sa, sb)
        big = rng.integers(0, maxv + 1, size=(40, 64)).astype(o.dtype) if trial else np.full((40, 64), maxv, o.dtype)
        if trial == 1:
            yy, xx = np.mgrid[0:40, 0:64]
            big = (((yy ^ xx) & 1) * maxv).astype(o.dtype)
        for size, (w, h) in ((0, (16, 16)), (1, (16, 8)), (2, (8, 16)), (3, (8, 8))):
            assert L.rh_hadamard_ac(r.ctx, size, _ptr(big, 64 + 3), 64) == f_hac(_ptr(big, 64 + 3), 64, w, h), (trial, size)
        for height in (16, 32, 9):
            assert L.rh_vsad(r.ctx, _ptr(big, 5), 64, height) == f_vsad(_ptr(big, 5), 64, height)
        for height in (8, 16, 4):
            assert L.rh_asd8(r.ctx, _ptr(big, 2), 64, _ptr(big, 64 * 3 + 17), 64, height) == f_asd8(_ptr(big, 2), 64, _ptr(big, 64 * 3 + 17), 64, height)


def test_integral_init_and_ads(env):
    """oracle vs h->mc.integral_init{4h,8h,4v,8v} (checkasm.c:1745-1770 geometry: stride 96) and pixf.ads[] (checkasm.c:836-885:
    saturating multiples of 8*PIXEL_MAX and random 14/16-bit sums)"""
    r, o, d = env
    rng = np.random.default_rng(9)
    maxv = (1 << d) - 1
    L = r.lib
    L.rh_integral_init.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    stride = 96
    for trial in range(4):
        pix = (rng.integers(0, maxv + 1, size=(4, stride)) if trial else np.full((4, stride), maxv)).astype(o.dtype)
        base = rng.integers(0, 65536, size=(24, stride)).astype(np.uint16)
        for kind, name in ((0, "integral_init4h"), (1, "integral_init8h")):
            a, b = base.copy(), base.copy()
            L.rh_integral_init(r.ctx, kind, _ptr(a, stride), None, _ptr(pix), stride)
            f = o.f(name); f.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
            f(_ptr(b, stride), _ptr(pix), stride)
            assert np.array_equal(a, b), name
        a, b = base.copy(), base.copy()
        L.rh_integral_init(r.ctx, 2, _ptr(a), _ptr(a, 14 * stride), None, stride)
        f = o.f("integral_init4v"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        f(_ptr(b), _ptr(b, 14 * stride), stride)
        assert np.array_equal(a, b)
        a, b = base.copy(), base.copy()
        L.rh_integral_init(r.ctx, 3, _ptr(a), None, None, stride)
        f = o.f("integral_init8v"); f.argtypes = [C.c_void_p, C.c_long]
        f(_ptr(b), stride)
        assert np.array_equal(a, b)
    L.rh_ads.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.rh_ads.restype = C.c_int
    f = o.f("ads", C.c_int)
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    cost = rng.integers(0, 65536, size=32).astype(np.uint16)
    for i in range(100):
        size_idx, n_dc = ((0, 4), (1, 2), (3, 1))[i % 3]
        if i < 40:
            sums = (rng.integers(0, 9, size=72) * 8 * maxv).astype(np.uint16)
            dc = (rng.integers(0, 9, size=4) * 8 * maxv).astype(np.int32)
        else:
            sums = rng.integers(0, 1 << (d + 6), size=72).astype(np.uint16)
            dc = rng.integers(0, 1 << (d + 6), size=4).astype(np.int32)
        thresh = int(rng.integers(0, 257)) * maxv + int(rng.integers(0, 65536))
        ma, mb = np.zeros(48, np.int16), np.zeros(48, np.int16)
        na = L.rh_ads(r.ctx, size_idx, _ptr(dc), _ptr(sums), 32, _ptr(cost), _ptr(ma), 28, thresh)
        nb = f(n_dc, _ptr(dc), _ptr(sums), 32, _ptr(cost), _ptr(mb), 28, thresh)
        assert na == nb and np.array_equal(ma[:na], mb[:nb]), (i, n_dc)


def test_dct_quant(env):
    r, o, d = env
    rng = np.random.default_rng(6)
    maxv = (1 << d) - 1
    L = r.lib
    nco = {0: 16, 1: 64, 2: 256, 3: 64, 4: 256, 5: 4, 6: 8}
    for fenc in _patterns(rng, (16, 16), o.dtype, maxv):
        for fdec in _patterns(rng, (16, 32), o.dtype, maxv):
            for kind, n in nco.items():
                a = np.zeros(n, o.coef_dtype); b = np.zeros(n, o.coef_dtype)
                L.rh_dct(r.ctx, kind, _ptr(a), _ptr(fenc), _ptr(fdec))
                o.f("dct")(kind, _ptr(b), _ptr(fenc), _ptr(fdec))
                assert np.array_equal(a, b), ("dct", kind)
            a = rng.integers(-2000, 2000, size=16).astype(o.coef_dtype); b = a.copy()
            L.rh_dct(r.ctx, 7, _ptr(a), None, None)
            o.f("dct")(7, _ptr(b), None, None)
            assert np.array_equal(a, b)
            a = rng.integers(-4000, 4000, size=8).astype(o.coef_dtype); b = a.copy()  # dct2x4dc (4:2:2 chroma DC)
            L.rh_dct(r.ctx, 8, _ptr(a), None, None)
            o.f("dct")(8, _ptr(b), None, None)
            assert np.array_equal(a, b)
    L.rh_quant.restype = C.c_int
    for i_list in range(4):
        for qp in range(0, 52 + 6 * (d - 8), 3):
            for kind, n, is8 in ((0, 16, 0), (1, 64, 1), (2, 64, 0), (3, 16, 0), (4, 4, 0)):
                if is8 and i_list > 1:
                    continue
                il = i_list if not is8 else i_list
                mf = np.zeros(64 if is8 else 16, o.ucoef_dtype); bias = np.zeros_like(mf)
                L.rh_quant_tables(r.ctx, is8, il, qp, _ptr(mf), _ptr(bias))
                lim = 30000 if d == 8 else 1 << 20
                for amp in (lim, 300, 3):
                    a = rng.integers(-amp, amp + 1, size=n).astype(o.coef_dtype); b = a.copy()
                    ra = L.rh_quant(r.ctx, kind, _ptr(a), il, qp, 0)
                    rb = o.f("quant", C.c_int)(kind, _ptr(b), _ptr(mf), _ptr(bias), int(mf[0]) >> 1, int(bias[0]) << 1)
                    assert ra == rb and np.array_equal(a, b), ("quant", kind, i_list, qp, amp)
