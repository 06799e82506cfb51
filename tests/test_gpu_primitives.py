"""GPU parity of the batched vtable primitives against the oracle, through the C ABI.

Mirrors how the reference's own checkasm exercises them (tools/checkasm.c:340-420 pixel metrics on random and
extreme blocks, :1715-1744 lowres widths 96..120, :560-640 dct, :1950-2050 quant): random data plus the
saturating patterns, every block size the device entry offers, both bit depths."""
import ctypes as C

import numpy as np
import pytest

from oracle.oraclelib import Oracle
from x264_amd import lib

pytestmark = pytest.mark.gpu


def _ptr(a, off=0):
    return C.c_void_p(a.ctypes.data + off * a.itemsize)


def _planes(rng, h, w, dtype, maxv, kind):
    if kind == "random":
        return rng.integers(0, maxv + 1, size=(h, w)).astype(dtype), rng.integers(0, maxv + 1, size=(h, w)).astype(dtype)
    if kind == "extreme":  # |diff| = max everywhere in a checkerboard: largest Hadamard sums
        yy, xx = np.mgrid[0:h, 0:w]
        a = (((yy ^ xx) & 1) * maxv).astype(dtype)
        return a, (maxv - a).astype(dtype)
    a = rng.integers(0, maxv + 1, size=(h, w)).astype(dtype)
    return a, np.clip(a.astype(np.int32) + rng.integers(-3, 4, size=(h, w)), 0, maxv).astype(dtype)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("dims", [(64, 48), (1920, 1072), (112, 16), (16, 16)])
def test_pixel_cmp_batch(depth, dims):
    import torch
    W, H = dims
    o = Oracle(depth)
    maxv = (1 << depth) - 1
    rng = np.random.default_rng(W * 31 + H + depth)
    PAD = 32
    stride = W + 2 * PAD + 5  # deliberately not a multiple of 4: every ref row is misaligned differently
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        kinds = ["random", "extreme", "near"] if W <= 112 else ["random"]
        for kind in kinds:
            f, r = _planes(rng, H + 2 * PAD, stride, o.dtype, maxv, kind)
            tdt = torch.uint8 if depth == 8 else torch.int16
            fd = torch.from_numpy(f.view(np.uint8 if depth == 8 else np.int16)).cuda()
            rd = torch.from_numpy(r.view(np.uint8 if depth == 8 else np.int16)).cuda()
            assert fd.dtype == tdt
            org = (PAD * stride + PAD) * f.itemsize
            for size_idx, (sw, sh) in enumerate(((16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4))):  # PIXEL_16x16 .. PIXEL_4x4
                bw, bh = W // sw, H // sh
                mv = rng.integers(-PAD, PAD - 15, size=(bw * bh, 2)).astype(np.int16)
                mvd = torch.from_numpy(mv).cuda()
                for satd in (0, 1):
                    out = torch.full((bw * bh,), -1, dtype=torch.int32, device="cuda")
                    torch.cuda.synchronize()  # torch fills on its own stream; the context launches on another one
                    ctx.pixel_cmp_batch(satd, size_idx, fd.data_ptr() + org, rd.data_ptr() + org, stride, bw, bh, mvd.data_ptr(), out.data_ptr())
                    ctx.synchronize()
                    got = out.cpu().numpy()
                    fn = o.f("satd" if satd else "sad", C.c_int)
                    # the oracle is a scalar C loop: check every block of small planes, a strided sample of the large one
                    step = 1 if bw * bh <= 4096 else 37
                    for bi in range(0, bw * bh, step):
                        by, bx = divmod(bi, bw)
                        a = _ptr(f, (PAD + by * sh) * stride + PAD + bx * sw)
                        b = _ptr(r, (PAD + by * sh + int(mv[bi, 1])) * stride + PAD + bx * sw + int(mv[bi, 0]))
                        assert got[bi] == fn(a, stride, b, stride, sw, sh), (kind, sw, sh, satd, bi)
                    assert (got >= 0).all()
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_frame_init_lowres_core(depth):
    import torch
    o = Oracle(depth)
    maxv = (1 << depth) - 1
    rng = np.random.default_rng(7 + depth)
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        for w, h in [(96, 8), (104, 8), (120, 8), (960, 540), (33, 5)]:
            src = rng.integers(0, maxv + 1, size=(2 * h + 2, 2 * w + 16)).astype(o.dtype)
            exp = np.zeros((4, h, w), o.dtype)
            o.f("lowres_core")(_ptr(src), _ptr(exp[0]), _ptr(exp[1]), _ptr(exp[2]), _ptr(exp[3]), src.shape[1], w, w, h)
            vdt = np.uint8 if depth == 8 else np.int16
            sd = torch.from_numpy(src.view(vdt)).cuda()
            dd = torch.zeros((4, h, w), dtype=sd.dtype, device="cuda")
            torch.cuda.synchronize()  # torch fills on its own stream; the context launches on another one
            ctx.frame_init_lowres_core(sd.data_ptr(), [dd[i].data_ptr() for i in range(4)], src.shape[1], w, w, h)
            ctx.synchronize()
            got = dd.cpu().numpy().view(o.dtype)
            assert np.array_equal(got, exp), (w, h)
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("is8", [0, 1])
def test_dct_quant_batch(depth, is8):
    o = Oracle(depth)
    maxv = (1 << depth) - 1
    rng = np.random.default_rng(11 + depth + is8)
    N = 8 if is8 else 4
    n = 777
    fenc = rng.integers(0, maxv + 1, size=(n, N, 16)).astype(o.dtype)
    fdec = rng.integers(0, maxv + 1, size=(n, N, 32)).astype(o.dtype)
    fenc[0] = maxv; fdec[0] = 0; fenc[1] = 0; fdec[1] = maxv          # saturating blocks
    fdec[2, :, :16] = fenc[2]                                         # all-zero residual
    mf = rng.integers(500, 14000, size=N * N).astype(o.ucoef_dtype)
    bias = rng.integers(0, 30000, size=N * N).astype(o.ucoef_dtype)
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        coefs, nz = ctx.dct_quant_batch(is8, fenc, fdec, mf, bias)
    finally:
        ctx.close()
    dct, quant = o.f("dct"), o.f("quant", C.c_int)
    for i in range(n):
        c = np.zeros(N * N, o.coef_dtype)
        dct(3 if is8 else 0, _ptr(c), _ptr(fenc[i]), _ptr(fdec[i]))
        rnz = quant(1 if is8 else 0, _ptr(c), _ptr(mf), _ptr(bias), 0, 0)
        assert np.array_equal(coefs[i], c), i
        assert int(nz[i]) == rnz, i


@pytest.mark.parametrize("depth", [8, 10])
def test_hpel_filter(depth):
    """x264_mc_functions_t.hpel_filter (mc.c:172-196) on whole planes, incl. the extra dstv columns; checkasm.c:1600-1640
    exercises it with random data, here also the saturating checkerboard."""
    import torch
    o = Oracle(depth)
    maxv = (1 << depth) - 1
    rng = np.random.default_rng(21 + depth)
    f = o.f("hpel_filter")
    f.argtypes = [C.c_void_p] * 4 + [C.c_long, C.c_int, C.c_int, C.c_void_p]
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        for w, h, kind in [(64, 16, "r"), (100, 9, "r"), (37, 5, "x"), (1920, 1080, "r"), (130, 33, "x"), (248, 8, "r"), (251, 10, "x"),
                           (499, 20, "r"), (745, 3, "r")]:  # (the 8-bit kernel works in strips of 248 columns x 8 rows)
            stride = w + 32
            src = rng.integers(0, maxv + 1, size=(h + 8, stride)).astype(o.dtype)
            if kind == "x":
                src[:] = np.where((np.indices(src.shape).sum(0) & 1) == 0, 0, maxv).astype(o.dtype)
            off = 3 * stride + 8
            exp = [np.full((h + 8, stride), 7, o.dtype) for _ in range(3)]
            buf = np.zeros(w + 64, np.int16)
            f(_ptr(exp[0], off), _ptr(exp[1], off), _ptr(exp[2], off), _ptr(src, off), stride, w, h, _ptr(buf))
            vdt = np.uint8 if depth == 8 else np.int16
            sd = torch.from_numpy(src.view(vdt)).cuda()
            dd = torch.full((3, h + 8, stride), 7, dtype=sd.dtype, device="cuda")
            torch.cuda.synchronize()
            ob = off * src.itemsize
            ctx.hpel_filter(dd[0].data_ptr() + ob, dd[1].data_ptr() + ob, dd[2].data_ptr() + ob, sd.data_ptr() + ob, stride, w, h)
            ctx.synchronize()
            got = dd.cpu().numpy().view(o.dtype)
            for k in range(3):
                assert np.array_equal(got[k], exp[k]), (w, h, kind, "hvc"[k])
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_frame_dct_quant4x4(depth):
    """sub4x4_dct + quant_4x4 of every block of a plane pair against the oracle's per-block functions."""
    import torch
    o = Oracle(depth)
    maxv = (1 << depth) - 1
    rng = np.random.default_rng(31 + depth)
    W, H = 1036, 68  # 259 x 17 blocks: a ragged last wave
    fs, ds = W + 12, W + 40
    fenc = rng.integers(0, maxv + 1, size=(H, fs)).astype(o.dtype)
    fdec = rng.integers(0, maxv + 1, size=(H, ds)).astype(o.dtype)
    fenc[:4, :8] = maxv; fdec[:4, :8] = 0          # saturating blocks
    fdec[4:8, :4] = fenc[4:8, :4]                  # zero residual
    mf = rng.integers(500, 14000, size=16).astype(o.ucoef_dtype)
    bias = rng.integers(0, 30000, size=16).astype(o.ucoef_dtype)
    vdt = np.uint8 if depth == 8 else np.int16
    fd, dd = torch.from_numpy(fenc.view(vdt)).cuda(), torch.from_numpy(fdec.view(vdt)).cuda()
    bw, bh = W // 4, H // 4
    cdt = torch.int16 if depth == 8 else torch.int32
    coefs = torch.full((bh, bw, 16), 77, dtype=cdt, device="cuda")
    nz = torch.full((bh, bw), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        ctx.frame_dct_quant4x4(fd.data_ptr(), fs, dd.data_ptr(), ds, W, H, mf, bias, coefs.data_ptr(), nz.data_ptr())
        ctx.synchronize()
    finally:
        ctx.close()
    got, gnz = coefs.cpu().numpy(), nz.cpu().numpy()
    dct, quant = o.f("dct"), o.f("quant", C.c_int)
    fe16 = np.zeros((4, 16), o.dtype); fd32 = np.zeros((4, 32), o.dtype)
    for by in range(bh):
        for bx in range(0, bw, 3 if by else 1):
            fe16[:, :4] = fenc[4 * by:4 * by + 4, 4 * bx:4 * bx + 4]
            fd32[:, :4] = fdec[4 * by:4 * by + 4, 4 * bx:4 * bx + 4]
            c = np.zeros(16, o.coef_dtype)
            dct(0, _ptr(c), _ptr(fe16), _ptr(fd32))
            rnz = quant(0, _ptr(c), _ptr(mf), _ptr(bias), 0, 0)
            assert np.array_equal(got[by, bx], c), (by, bx)
            assert int(gnz[by, bx]) == rnz, (by, bx)


@pytest.mark.parametrize("depth", [8, 10])
def test_vtable_batches(depth):
    """x264hip_dct_batch / quant_batch / var2_batch / ads_batch: every remaining entry of x264_dct_function_t and
    x264_quant_function_t, var2 and ads of x264_pixel_function_t, on checkasm-style inputs (tools/checkasm.c:361-888, 890-1224)
    against the oracle (pinned against the reference vtables in tests/test_primitives_vs_ref.py)."""
    from tests.test_vtable_blocks_host import DCT_COEFS, QUANT_COEFS
    o = Oracle(depth)
    rng = np.random.default_rng(77 + depth)
    maxv = (1 << depth) - 1
    n = 300
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        fenc = rng.integers(0, maxv + 1, size=(n, 16, 16)).astype(o.dtype)
        fdec = rng.integers(0, maxv + 1, size=(n, 16, 32)).astype(o.dtype)
        fenc[0] = maxv; fdec[0] = 0; fenc[1] = 0; fdec[1] = maxv  # largest differences of either sign
        fdec[2, :, :16] = fenc[2]
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        for kind in range(7):
            got = ctx.dct_batch(kind, fenc, fdec)
            for i in range(n):
                want = np.zeros(DCT_COEFS[kind], o.coef_dtype)
                o.f("dct")(kind, p(want), p(fenc[i]), p(fdec[i]))
                assert np.array_equal(got[i], want), ("dct", kind, i)
        for kind in (7, 8):
            co = rng.integers(-(40000 if depth == 10 else 8000), 8000, size=(n, DCT_COEFS[kind])).astype(o.coef_dtype)
            got = ctx.dct_batch(kind, coefs=co)
            for i in range(n):
                want = co[i].copy()
                o.f("dct")(kind, p(want), None, None)
                assert np.array_equal(got[i], want), ("dct", kind, i)
        lim = 30000 if depth == 8 else 1 << 20
        for kind, nc in QUANT_COEFS.items():
            nt = 64 if kind == 1 else 16
            mf = rng.integers(1, 30000 if depth == 8 else 1 << 18, size=nt).astype(o.ucoef_dtype)
            bias = rng.integers(0, 30000, size=nt).astype(o.ucoef_dtype)
            co = rng.integers(-lim, lim + 1, size=(n, nc)).astype(o.coef_dtype)
            co[: n // 3] = rng.integers(-3, 4, size=(n // 3, nc))  # many all-zero results: exercises the nz flags / masks
            got, nz = ctx.quant_batch(kind, co, mf, bias, int(mf[0]) >> 1, int(bias[0]) << 1)
            for i in range(n):
                want = co[i].copy()
                r = o.f("quant", C.c_int)(kind, p(want), p(mf), p(bias), int(mf[0]) >> 1, int(bias[0]) << 1)
                assert np.array_equal(got[i], want) and nz[i] == r, ("quant", kind, i)
        for h in (8, 16):
            var, ssd = ctx.var2_batch(h, fenc, fdec)
            for i in range(n):
                s = np.zeros(2, np.int32)
                r = o.f("var2", C.c_int)(p(fenc[i]), p(fdec[i]), h, p(s))
                assert var[i] == r and np.array_equal(ssd[i], s), ("var2", h, i)
        # ads: rows of different widths (more and less than one wave), the three forms, thresholds from "none" to "all"
        sums = rng.integers(0, 1 << 16, size=5000).astype(np.uint16)
        cost = rng.integers(0, 200, size=2000).astype(np.uint16)
        calls, off = [], 0
        for i in range(90):
            width = int(rng.integers(1, 200))
            calls.append(dict(n_dc=(1, 2, 4)[i % 3], delta=32, width=width, thresh=int(rng.integers(0, 140000)),
                              enc_dc=rng.integers(0, 1 << 16, size=4), sums_off=int(rng.integers(0, 4000)), cost_off=int(rng.integers(0, 1700)), mvs_off=off))
            off += width
        mvs, counts = ctx.ads_batch(calls, sums, cost, off)
        f = o.f("ads", C.c_int)
        for c, cnt in zip(calls, counts):
            want = np.zeros(c["width"], np.int16)
            dc = np.ascontiguousarray(c["enc_dc"], np.int32)
            r = f(c["n_dc"], p(dc), p(sums[c["sums_off"]:]), c["delta"], p(cost[c["cost_off"]:]), p(want), c["width"], c["thresh"])
            assert cnt == r and np.array_equal(mvs[c["mvs_off"]:c["mvs_off"] + r], want[:r]), c
    finally:
        ctx.close()


class McFunctions(C.Structure):
    """x264hip_mc_functions: the coarse members of x264_mc_functions_t with the reference's exact signatures"""
    _fields_ = [
        ("plane_copy", C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int)),
        ("hpel_filter", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_void_p)),
        ("frame_init_lowres_core", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int)),
        ("mbtree_propagate_cost", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int)),
        ("mbtree_propagate_list", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)),
    ]


@pytest.mark.parametrize("depth", [8, 10])
def test_mc_fill_exact_signature_members(depth):
    """x264hip_mc_fill: the frame- and row-granular members of x264_mc_functions_t (mc.h:292,306-307,326-327,333-337) called
    through the function pointers with HOST buffers, exactly like the encoder calls h->mc.*, against the oracle."""
    o = Oracle(depth)
    maxv = (1 << depth) - 1
    rng = np.random.default_rng(91 + depth)
    W, H = 352, 288
    mb_w, mb_h = W // 16, H // 16
    ctx = lib.Context(W, H, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        pf = McFunctions()
        ctx.L.x264hip_mc_fill.argtypes = [C.c_void_p, C.POINTER(McFunctions)]
        assert ctx.L.x264hip_mc_fill(ctx.h, C.byref(pf)) == 0
        # plane_copy
        src = rng.integers(0, maxv + 1, size=(37, 120)).astype(o.dtype)
        dst = np.full((37, 140), 9, o.dtype)
        pf.plane_copy(_ptr(dst), 140, _ptr(src), 120, 101, 37)
        assert np.array_equal(dst[:, :101], src[:, :101]) and (dst[:, 101:] == 9).all()
        # frame_init_lowres_core
        w, h = 104, 21
        src = rng.integers(0, maxv + 1, size=(2 * h + 2, 2 * w + 16)).astype(o.dtype)
        exp = np.zeros((4, h, w + 8), o.dtype); got = np.zeros_like(exp)
        o.f("lowres_core")(_ptr(src), _ptr(exp[0]), _ptr(exp[1]), _ptr(exp[2]), _ptr(exp[3]), src.shape[1], w + 8, w, h)
        pf.frame_init_lowres_core(_ptr(src), _ptr(got[0]), _ptr(got[1]), _ptr(got[2]), _ptr(got[3]), src.shape[1], w + 8, w, h)
        assert np.array_equal(got, exp)
        # hpel_filter, incl. the extra dstv columns
        f = o.f("hpel_filter")
        f.argtypes = [C.c_void_p] * 4 + [C.c_long, C.c_int, C.c_int, C.c_void_p]
        w, h = 100, 19
        stride = w + 32
        src = rng.integers(0, maxv + 1, size=(h + 8, stride)).astype(o.dtype)
        off = 3 * stride + 8
        exp = [np.full((h + 8, stride), 7, o.dtype) for _ in range(3)]
        got = [np.full((h + 8, stride), 7, o.dtype) for _ in range(3)]
        buf = np.zeros(w + 64, np.int16)
        f(_ptr(exp[0], off), _ptr(exp[1], off), _ptr(exp[2], off), _ptr(src, off), stride, w, h, _ptr(buf))
        pf.hpel_filter(_ptr(got[0], off), _ptr(got[1], off), _ptr(got[2], off), _ptr(src, off), stride, w, h, _ptr(buf))
        for k in range(3):
            assert np.array_equal(got[k], exp[k]), "hvc"[k]
        # mbtree_propagate_cost + mbtree_propagate_list, row by row like macroblock_tree_propagate (slicetype.c:1051-1089),
        # against the oracle's whole-frame propagation
        n = mb_w * mb_h
        intra = rng.integers(1, 0x3FFF, size=n).astype(np.uint16)
        lists = rng.integers(1, 4, size=n).astype(np.uint16)
        lc = (np.minimum(rng.integers(0, 0x3FFF, size=n), intra) | (lists << 14)).astype(np.uint16)
        inv = rng.integers(64, 1024, size=n).astype(np.uint16)
        pin = rng.integers(0, 20000, size=n).astype(np.uint16)
        mv0 = rng.integers(-200, 200, size=(n, 2)).astype(np.int16); mv1 = rng.integers(-200, 200, size=(n, 2)).astype(np.int16)
        mv0[::7] = 0
        fps = np.float32(0.04 / (0.04 * 256.0) * 0.5)
        want = [rng.integers(0, 30000, size=n).astype(np.uint16) for _ in range(2)]
        have = [w_.copy() for w_ in want]
        L = o.lib
        L.or_mbtree_propagate.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_float]
        L.or_mbtree_propagate(mb_w, mb_h, _ptr(intra), _ptr(lc), _ptr(inv), _ptr(pin), _ptr(mv0), _ptr(mv1), _ptr(want[0]), _ptr(want[1]), 40, fps)
        fpsf = C.c_float(float(fps))
        for y in range(mb_h):
            s = slice(y * mb_w, (y + 1) * mb_w)
            amount = np.zeros(mb_w, np.int16)
            row = [np.ascontiguousarray(a[s]) for a in (pin, intra, lc, inv)]
            pf.mbtree_propagate_cost(_ptr(amount), _ptr(row[0]), _ptr(row[1]), _ptr(row[2]), _ptr(row[3]), C.byref(fpsf), mb_w)
            for lst, mv in ((0, mv0), (1, mv1)):
                mvr = np.ascontiguousarray(mv[s])
                pf.mbtree_propagate_list(None, _ptr(have[lst]), _ptr(mvr), _ptr(amount), _ptr(row[2]), 40 if lst == 0 else 24, y, mb_w, lst)
        assert np.array_equal(have[0], want[0]) and np.array_equal(have[1], want[1])
    finally:
        ctx.close()


_CMP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t)
_ONE = C.CFUNCTYPE(C.c_uint64, C.c_void_p, C.c_ssize_t)


_X3 = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p)
_X4 = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p)


class PixelFunctions(C.Structure):
    """x264hip_pixel_functions (member names and signatures of x264_pixel_function_t, common/pixel.h:78-146)"""
    _fields_ = [("sad", _CMP * 8), ("ssd", _CMP * 8), ("satd", _CMP * 8), ("sa8d", _CMP * 4), ("var", _ONE * 4), ("hadamard_ac", _ONE * 4),
                ("sad_x3", _X3 * 7), ("sad_x4", _X4 * 7), ("satd_x3", _X3 * 7), ("satd_x4", _X4 * 7),
                ("vsad", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_ssize_t, C.c_int)),
                ("asd8", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int)),
                ("var2", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p) * 4),
                ("ads", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int) * 7),
                ("mbcmp", _CMP * 8), ("mbcmp_unaligned", _CMP * 8), ("fpelcmp", _CMP * 8), ("fpelcmp_x3", _X3 * 7), ("fpelcmp_x4", _X4 * 7),
                ("sad_aligned", _CMP * 8)]


_SUB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)


class DctFunctions(C.Structure):
    """x264hip_dct_functions (x264_dct_function_t, common/dct.h:29-59)"""
    _fields_ = [("sub4x4_dct", _SUB), ("sub8x8_dct", _SUB), ("sub8x8_dct_dc", _SUB), ("sub8x16_dct_dc", _SUB), ("sub16x16_dct", _SUB), ("sub8x8_dct8", _SUB),
                ("sub16x16_dct8", _SUB), ("dct4x4dc", C.CFUNCTYPE(None, C.c_void_p)), ("dct2x4dc", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p))]


_Q = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
_QDC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)


class QuantFunctions(C.Structure):
    """x264hip_quant_functions (x264_quant_function_t, common/quant.h:30-45)"""
    _fields_ = [("quant_8x8", _Q), ("quant_4x4", _Q), ("quant_4x4x4", _Q), ("quant_4x4_dc", _QDC), ("quant_2x2_dc", _QDC)]


@pytest.mark.parametrize("depth", [8, 10])
def test_table_fillers_exact_signature_members(depth):
    """x264hip_pixel_fill / x264hip_dct_fill / x264hip_quant_fill: members with the reference's exact signatures, called the way the
    encoder calls h->pixf.* / h->dctf.* / h->quantf.* -- host pointers INTO macroblock buffers (fenc stride 16, fdec stride 32, any block
    offset), one block per call -- against the oracle.  Also: two contexts of different bit depth stay bound side by side, and the
    members can be called from several threads at once."""
    from tests.test_vtable_blocks_host import DCT_COEFS
    o = Oracle(depth)
    maxv = (1 << depth) - 1
    rng = np.random.default_rng(17 + depth)
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    other = lib.Context(64, 64, bit_depth=18 - depth, max_frames=2, mv_range=32)  # the other bit depth: its binding must not disturb ours
    try:
        L = ctx.L
        pf, df, qf, opf = PixelFunctions(), DctFunctions(), QuantFunctions(), PixelFunctions()
        for name, st in (("pixel", pf), ("dct", df), ("quant", qf)):
            fn = getattr(L, "x264hip_%s_fill" % name)
            fn.argtypes = [C.c_void_p, C.c_void_p]
            assert fn(ctx.h, C.byref(st)) == 0
        assert L.x264hip_pixel_fill(other.h, C.byref(opf)) == 0
        sizes = [(16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)]
        # the encoder's macroblock buffers: 48 rows of FENC_STRIDE, 54 rows of FDEC_STRIDE (common/common.h), blocks at interior offsets
        fenc = rng.integers(0, maxv + 1, size=(48, 16)).astype(o.dtype)
        fdec = rng.integers(0, maxv + 1, size=(54, 32)).astype(o.dtype)
        fenc[:4, :4] = maxv; fdec[:4, :4] = 0
        isz = fenc.itemsize
        for idx, (w, h) in enumerate(sizes):
            for (ye, xe, yd, xd) in ((0, 0, 0, 0), (16 - h, 16 - w, 2 + 16 - h, 16 - w), (4 * (h < 16), 4 * (w < 16), 7, 13)):
                a, b = _ptr(fenc, ye * 16 + xe), _ptr(fdec, yd * 32 + xd)
                for name, fo in (("sad", "sad"), ("satd", "satd"), ("ssd", "ssd")):
                    want = o.f(fo, C.c_int)(a, 16, b, 32, w, h)
                    assert getattr(pf, name)[idx](a, 16, b, 32) == want, (name, w, h, ye, xe)
            if (w, h) in ((16, 16), (8, 8)):
                assert pf.sa8d[idx](_ptr(fenc), 16, _ptr(fdec, 64), 32) == o.f("sa8d", C.c_int)(_ptr(fenc), 16, _ptr(fdec, 64), 32, w)
            if (w, h) in ((16, 16), (8, 16), (8, 8)):
                assert pf.var[idx](_ptr(fdec, 37), 32) == o.f("var", C.c_uint64)(_ptr(fdec, 37), 32, w, h), ("var", w, h)
            if idx < 4:
                assert pf.hadamard_ac[idx](_ptr(fdec, 70), 32) == o.f("hadamard_ac", C.c_uint64)(_ptr(fdec, 70), 32, w, h), ("hadamard_ac", w, h)
        assert not pf.sa8d[1] and not pf.var[1] and not pf.sad[7]  # entries the reference leaves empty stay NULL
        # sad_x3 / sad_x4 / satd_x3 / satd_x4 (common/pixel.c:441-516): one source block (FENC_STRIDE) against three / four candidates of a
        # reference plane with its own stride, every size
        plane = rng.integers(0, maxv + 1, size=(40, 72)).astype(o.dtype)
        for idx, (w, h) in enumerate(sizes):
            a = _ptr(fenc, 16 * (16 - h) + (16 - w))
            offs = [5 * 72 + 9, 6 * 72 + 10, 4 * 72 + 8, 5 * 72 + 11]
            for name, fo in (("sad", "sad"), ("satd", "satd")):
                want = [o.f(fo, C.c_int)(a, 16, _ptr(plane, off), 72, w, h) for off in offs]
                s3, s4 = np.full(3, -1, np.int32), np.full(4, -1, np.int32)
                getattr(pf, name + "_x3")[idx](a, *[_ptr(plane, off) for off in offs[:3]], 72, _ptr(s3))
                getattr(pf, name + "_x4")[idx](a, *[_ptr(plane, off) for off in offs], 72, _ptr(s4))
                assert s3.tolist() == want[:3] and s4.tolist() == want, (name, w, h)
        # vsad / asd8 (pixel.c:716-754)
        for hgt in (16, 8):
            assert pf.vsad(_ptr(fdec, 3 * 32 + 5), 32, hgt) == o.f("vsad", C.c_int)(_ptr(fdec, 3 * 32 + 5), 32, hgt)
            assert pf.asd8(_ptr(fenc, 4), 16, _ptr(fdec, 2 * 32 + 7), 32, hgt) == o.f("asd8", C.c_int)(_ptr(fenc, 4), 16, _ptr(fdec, 2 * 32 + 7), 32, hgt)
        # var2[PIXEL_8x16] / [PIXEL_8x8] (pixel.c:206-231): the chroma halves of the encoder's fenc / fdec buffers
        cfe = rng.integers(0, maxv + 1, size=(16, 16)).astype(o.dtype); cfd = rng.integers(0, maxv + 1, size=(16, 32)).astype(o.dtype)
        for slot, hgt in ((2, 16), (3, 8)):
            ws, gs = np.zeros(2, np.int32), np.zeros(2, np.int32)
            want = o.f("var2", C.c_int)(_ptr(cfe), _ptr(cfd), hgt, _ptr(ws))
            assert pf.var2[slot](_ptr(cfe), _ptr(cfd), _ptr(gs)) == want and np.array_equal(gs, ws), ("var2", hgt)
        assert not pf.var2[0] and not pf.ads[2]
        # ads[PIXEL_16x16] = ads4, [PIXEL_16x8] = ads2, [PIXEL_8x8] = ads1 (pixel.c:756-803)
        sums = rng.integers(0, 1 << 16, size=400).astype(np.uint16); cost = rng.integers(0, 200, size=200).astype(np.uint16)
        for slot, ndc in ((0, 4), (1, 2), (3, 1)):
            for width, thresh in ((37, 60000), (150, 90000), (8, 0)):
                dc = rng.integers(0, 1 << 16, size=4).astype(np.int32)
                want = np.zeros(width, np.int16); got = np.full(width, -7, np.int16)
                r = o.f("ads", C.c_int)(ndc, _ptr(dc), _ptr(sums), 32, _ptr(cost), _ptr(want), width, thresh)
                assert pf.ads[slot](_ptr(dc), _ptr(sums), 32, _ptr(cost), _ptr(got), width, thresh) == r and np.array_equal(got[:r], want[:r]), ("ads", ndc, width)
        # mbcmp / fpelcmp (mbcmp_init, encoder/encoder.c:1409-1427): this context has subme 7 and me dia -> mbcmp = satd, fpelcmp = sad
        a, b = _ptr(fenc), _ptr(fdec, 64)
        assert pf.mbcmp[0](a, 16, b, 32) == pf.satd[0](a, 16, b, 32) == pf.mbcmp_unaligned[0](a, 16, b, 32)
        assert pf.fpelcmp[3](a, 16, b, 32) == pf.sad[3](a, 16, b, 32) == pf.sad_aligned[3](a, 16, b, 32)
        s3a, s3b = np.zeros(3, np.int32), np.zeros(3, np.int32)
        pf.fpelcmp_x3[0](a, _ptr(plane, 80), _ptr(plane, 81), _ptr(plane, 152), 72, _ptr(s3a)); pf.sad_x3[0](a, _ptr(plane, 80), _ptr(plane, 81), _ptr(plane, 152), 72, _ptr(s3b))
        assert np.array_equal(s3a, s3b)
        # dct members: (kind of x264hip_dct_batch, member)
        for kind, name in ((0, "sub4x4_dct"), (1, "sub8x8_dct"), (2, "sub16x16_dct"), (3, "sub8x8_dct8"), (4, "sub16x16_dct8"), (5, "sub8x8_dct_dc"), (6, "sub8x16_dct_dc")):
            bs = 16 if "16x16" in name else 8 if "8x" in name else 4
            ye = xe = 0 if bs == 16 else 8 if bs == 8 else 12
            a, b = _ptr(fenc, ye * 16 + xe), _ptr(fdec, (2 + ye) * 32 + xe)
            want = np.zeros(DCT_COEFS[kind], o.coef_dtype); got = np.full(DCT_COEFS[kind], 7, o.coef_dtype)
            o.f("dct")(kind, _ptr(want), a, b)
            getattr(df, name)(_ptr(got), a, b)
            assert np.array_equal(got, want), name
        co = rng.integers(-4000, 4000, size=16).astype(o.coef_dtype); want = co.copy()
        o.f("dct")(7, _ptr(want), None, None); df.dct4x4dc(_ptr(co))
        assert np.array_equal(co, want)
        blk = rng.integers(-4000, 4000, size=(8, 16)).astype(o.coef_dtype)
        want = np.ascontiguousarray(blk[:, 0]); o.f("dct")(8, _ptr(want), None, None)
        got = np.zeros(8, o.coef_dtype); keep = blk.copy()
        df.dct2x4dc(_ptr(got), _ptr(blk))
        assert np.array_equal(got, want) and (blk[:, 0] == 0).all() and np.array_equal(blk[:, 1:], keep[:, 1:])
        # quant members
        lim = 30000 if depth == 8 else 1 << 20
        for kind, name, nc in ((1, "quant_8x8", 64), (0, "quant_4x4", 16), (2, "quant_4x4x4", 64)):
            nt = 64 if kind == 1 else 16
            mf = rng.integers(1, 30000 if depth == 8 else 1 << 18, size=nt).astype(o.ucoef_dtype)
            bias = rng.integers(0, 30000, size=nt).astype(o.ucoef_dtype)
            co = rng.integers(-lim, lim + 1, size=nc).astype(o.coef_dtype); want = co.copy()
            r = o.f("quant", C.c_int)(kind, _ptr(want), _ptr(mf), _ptr(bias), 0, 0)
            assert getattr(qf, name)(_ptr(co), _ptr(mf), _ptr(bias)) == r and np.array_equal(co, want), name
        for kind, name, nc in ((3, "quant_4x4_dc", 16), (4, "quant_2x2_dc", 4)):
            co = rng.integers(-lim, lim + 1, size=nc).astype(o.coef_dtype); want = co.copy()
            r = o.f("quant", C.c_int)(kind, _ptr(want), None, None, 11000, 20000)
            assert getattr(qf, name)(_ptr(co), 11000, 20000) == r and np.array_equal(co, want), name
        # re-entrancy: the same members from four threads at once (the reference calls table members from its frame threads)
        import threading
        errs = []

        def hammer(seed):
            r2 = np.random.default_rng(seed)
            fe = r2.integers(0, maxv + 1, size=(16, 16)).astype(o.dtype); fd = r2.integers(0, maxv + 1, size=(16, 32)).astype(o.dtype)
            for _ in range(40):
                if pf.satd[3](_ptr(fe), 16, _ptr(fd), 32) != o.f("satd", C.c_int)(_ptr(fe), 16, _ptr(fd), 32, 8, 8):
                    errs.append(seed)
        th = [threading.Thread(target=hammer, args=(s,)) for s in range(4)]
        [t.start() for t in th]; [t.join() for t in th]
        assert not errs
    finally:
        ctx.close(); other.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_multi_plane_forms_equal_single_calls(depth):
    """x264hip_pixel_cmp_batch_multi / _hpel_filter_multi / _frame_dct_quant4x4_multi: several independent planes / plane pairs in ONE launch
    (blockIdx.z picks the set) give exactly what one call per set gives (the single forms are checked against the oracle above)."""
    import torch
    o = Oracle(depth)
    maxv = (1 << depth) - 1
    rng = np.random.default_rng(77 + depth)
    vdt = np.uint8 if depth == 8 else np.int16
    N, PAD = 5, 32
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        # --- SAD / SATD of displaced blocks, all seven sizes
        W, H = 208, 112
        stride = W + 2 * PAD + 3
        planes = [torch.from_numpy(rng.integers(0, maxv + 1, size=(2, H + 2 * PAD, stride)).astype(o.dtype).view(vdt)).cuda() for _ in range(N)]
        org = (PAD * stride + PAD) * (1 if depth == 8 else 2)
        for size_idx, (sw, sh) in enumerate(((16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4))):
            bw, bh = W // sw, H // sh
            mvs = [torch.from_numpy(rng.integers(-PAD, PAD - 15, size=(bw * bh, 2)).astype(np.int16)).cuda() for _ in range(N)]
            for satd in (0, 1):
                one = [torch.full((bw * bh,), -1, dtype=torch.int32, device="cuda") for _ in range(N)]
                many = [torch.full((bw * bh,), -2, dtype=torch.int32, device="cuda") for _ in range(N)]
                torch.cuda.synchronize()
                for k in range(N):
                    ctx.pixel_cmp_batch(satd, size_idx, planes[k][0].data_ptr() + org, planes[k][1].data_ptr() + org, stride, bw, bh, mvs[k].data_ptr(), one[k].data_ptr())
                ctx.pixel_cmp_batch_multi(satd, size_idx, [p[0].data_ptr() + org for p in planes], [p[1].data_ptr() + org for p in planes], stride, bw, bh,
                                          [m.data_ptr() for m in mvs], [t.data_ptr() for t in many])
                ctx.synchronize()
                for k in range(N):
                    assert torch.equal(one[k], many[k]) and int(one[k].min()) >= 0, (size_idx, satd, k)
        # --- hpel_filter
        w, h = 499, 37
        st = w + 32
        off = (3 * st + 8) * (1 if depth == 8 else 2)
        srcs = [torch.from_numpy(rng.integers(0, maxv + 1, size=(h + 8, st)).astype(o.dtype).view(vdt)).cuda() for _ in range(N)]
        one = [torch.full((3, h + 8, st), 7, dtype=srcs[0].dtype, device="cuda") for _ in range(N)]
        many = [torch.full((3, h + 8, st), 7, dtype=srcs[0].dtype, device="cuda") for _ in range(N)]
        torch.cuda.synchronize()
        for k in range(N):
            ctx.hpel_filter(one[k][0].data_ptr() + off, one[k][1].data_ptr() + off, one[k][2].data_ptr() + off, srcs[k].data_ptr() + off, st, w, h)
        ctx.hpel_filter_multi([m[0].data_ptr() + off for m in many], [m[1].data_ptr() + off for m in many], [m[2].data_ptr() + off for m in many],
                              [s_.data_ptr() + off for s_ in srcs], st, w, h)
        ctx.synchronize()
        for k in range(N):
            assert torch.equal(one[k], many[k]), k
        assert not torch.equal(one[0], one[1])
        # --- sub4x4_dct + quant_4x4 of whole planes
        W2, H2 = 1036, 68
        fs, ds = W2 + 12, W2 + 40
        mf = rng.integers(500, 14000, size=16).astype(o.ucoef_dtype)
        bias = rng.integers(0, 30000, size=16).astype(o.ucoef_dtype)
        fe = [torch.from_numpy(rng.integers(0, maxv + 1, size=(H2, fs)).astype(o.dtype).view(vdt)).cuda() for _ in range(N)]
        fd = [torch.from_numpy(rng.integers(0, maxv + 1, size=(H2, ds)).astype(o.dtype).view(vdt)).cuda() for _ in range(N)]
        cdt = torch.int16 if depth == 8 else torch.int32
        mk = lambda fill: ([torch.full((H2 // 4, W2 // 4, 16), fill, dtype=cdt, device="cuda") for _ in range(N)],  # noqa: E731
                           [torch.full((H2 // 4, W2 // 4), 9, dtype=torch.uint8, device="cuda") for _ in range(N)])
        (c1, z1), (c2, z2) = mk(77), mk(55)
        torch.cuda.synchronize()
        for k in range(N):
            ctx.frame_dct_quant4x4(fe[k].data_ptr(), fs, fd[k].data_ptr(), ds, W2, H2, mf, bias, c1[k].data_ptr(), z1[k].data_ptr())
        ctx.frame_dct_quant4x4_multi([t.data_ptr() for t in fe], fs, [t.data_ptr() for t in fd], ds, W2, H2, mf, bias, [t.data_ptr() for t in c2], [t.data_ptr() for t in z2])
        ctx.synchronize()
        for k in range(N):
            assert torch.equal(c1[k], c2[k]) and torch.equal(z1[k], z2[k]), k
    finally:
        ctx.close()
