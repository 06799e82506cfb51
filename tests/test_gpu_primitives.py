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
            for size_idx, size in ((0, 16), (3, 8), (6, 4)):
                bw, bh = W // size, H // size
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
                        a = _ptr(f, (PAD + by * size) * stride + PAD + bx * size)
                        b = _ptr(r, (PAD + by * size + int(mv[bi, 1])) * stride + PAD + bx * size + int(mv[bi, 0]))
                        assert got[bi] == fn(a, stride, b, stride, size, size), (kind, size, satd, bi)
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
        for w, h, kind in [(64, 16, "r"), (100, 9, "r"), (37, 5, "x"), (1920, 1080, "r"), (130, 33, "x")]:
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
