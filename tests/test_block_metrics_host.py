"""The arithmetic of x264_amd/csrc/block_metrics.h (the functions the batched device kernels call for ssd / sa8d / var /
hadamard_ac / vsad / asd8) compiled for the host by tests/tools/block_metrics_host.cpp and checked against the oracle, which is
itself pinned against the reference vtables (tests/test_primitives_vs_ref.py).  The GPU test of the same entry points is in
tests/test_gpu_configs.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle.oraclelib import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "tools", "block_metrics_host.cpp")
OUT = os.path.join(HERE, "tools", "_build", "libbm_host.so")
METRICS = {  # name: (id, sizes, needs second plane)
    "ssd": (0, [(16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)], True),
    "sa8d": (1, [(16, 16), (8, 8)], True),
    "var": (2, [(16, 16), (8, 16), (8, 8)], False),
    "hadamard_ac": (3, [(16, 16), (16, 8), (8, 16), (8, 8)], False),
    "vsad": (4, [(16, 16), (16, 8)], False),
    "asd8": (5, [(8, 16), (8, 8)], True),
}


def _lib():
    hdrs = [os.path.join(HERE, "..", "x264_amd", "csrc", h) for h in ("block_metrics.h", "dct_quant_block.h", "me_full.h", "integral.h")]
    if not os.path.exists(OUT) or max([os.path.getmtime(SRC)] + [os.path.getmtime(h) for h in hdrs]) > os.path.getmtime(OUT):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-o", OUT, SRC])
    return C.CDLL(OUT)


def oracle_metric(o, name, w, h, a, sa, b, sb):
    p = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
    if name == "ssd":
        return o.f("ssd", C.c_int)(p(a), sa, p(b), sb, w, h)
    if name == "sa8d":
        return o.f("sa8d", C.c_int)(p(a), sa, p(b), sb, w)
    if name == "var":
        return o.f("var", C.c_uint64)(p(a), sa, w, h)
    if name == "hadamard_ac":
        fn = o.f("hadamard_ac", C.c_uint64)
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        return fn(p(a), sa, w, h)
    if name == "vsad":
        fn = o.f("vsad", C.c_int)
        fn.argtypes = [C.c_void_p, C.c_long, C.c_int]
        return fn(p(a), sa, h)
    fn = o.f("asd8", C.c_int)
    fn.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int]
    return fn(p(a), sa, p(b), sb, h)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("name", list(METRICS))
def test_block_metric_arithmetic(name, depth):
    L = _lib()
    fn = L.bm_host_u8 if depth == 8 else L.bm_host_u16
    fn.restype = C.c_uint64
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long]
    o = Oracle(depth)
    rng = np.random.default_rng(7 + depth)
    maxv = (1 << depth) - 1
    mid, sizes, two = METRICS[name]
    for kind in range(4):
        if kind == 0:
            a = rng.integers(0, maxv + 1, size=(40, 64)); b = rng.integers(0, maxv + 1, size=(40, 64))
        elif kind == 1:   # extremes: the largest differences every accumulator has to hold
            a = np.full((40, 64), maxv); b = np.zeros((40, 64), np.int64)
        elif kind == 2:
            a = (rng.integers(0, 2, size=(40, 64)) * maxv); b = maxv - a
        else:
            a = rng.integers(0, maxv + 1, size=(40, 64)); b = np.clip(a + rng.integers(-3, 4, size=(40, 64)), 0, maxv)
        a = np.ascontiguousarray(a, o.dtype); b = np.ascontiguousarray(b, o.dtype)
        for (w, h) in sizes:
            for (ox, oy) in ((0, 0), (5, 3), (17, 9)):
                pa, pb = a[oy:, ox:], b[oy + 1:, ox + 2:]
                got = fn(mid, w, h, pa.ctypes.data, 64, pb.ctypes.data if two else None, 64)
                want = oracle_metric(o, name, w, h, pa, 64, pb, 64)
                assert got == (want & 0xFFFFFFFFFFFFFFFF if name in ("var", "hadamard_ac") else want & 0xFFFFFFFF), (name, w, h, kind, ox, oy)


@pytest.mark.parametrize("depth", [8, 10])
def test_dct_quant8x8_block_arithmetic(depth):
    """dq_block8x8 (x264_amd/csrc/dct_quant_block.h, the body of frame_dct_quant8x8_kernel) against the oracle's sub8x8_dct8 +
    quant_8x8 (pinned against the reference vtables in tests/test_primitives_vs_ref.py)."""
    L = _lib()
    fn = L.dq8x8_host_u8 if depth == 8 else L.dq8x8_host_u16
    fn.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]
    o = Oracle(depth)
    dct, quant = o.f("dct"), o.f("quant", C.c_int)
    rng = np.random.default_rng(3 + depth)
    maxv = (1 << depth) - 1
    p = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
    for trial in range(60):
        fe = rng.integers(0, maxv + 1, size=(8, 24)).astype(o.dtype)
        fd = rng.integers(0, maxv + 1, size=(8, 40)).astype(o.dtype)
        if trial == 0:
            fe[:] = maxv; fd[:] = 0
        elif trial == 1:
            fd[:, :8] = fe[:, :8]
        elif trial == 2:
            fe[:, :8] = (np.indices((8, 8)).sum(0) % 2) * maxv; fd[:] = maxv // 2
        mf = rng.integers(300, 14000, size=64).astype(np.uint32)
        bias = rng.integers(0, 30000, size=64).astype(np.uint32)
        out = np.zeros(64, o.coef_dtype)
        nz = fn(p(fe), 24, p(fd), 40, p(mf), p(bias), p(out))
        fe16 = np.zeros((8, 16), o.dtype); fd32 = np.zeros((8, 32), o.dtype)
        fe16[:, :8] = fe[:, :8]; fd32[:, :8] = fd[:, :8]
        c = np.zeros(64, o.coef_dtype)
        dct(3, p(c), p(fe16), p(fd32))
        rnz = quant(1, p(c), p(mf.astype(o.ucoef_dtype)), p(bias.astype(o.ucoef_dtype)), 0, 0)
        assert np.array_equal(out, c), trial
        assert nz == rnz, trial


@pytest.mark.parametrize("depth", [8, 10])
def test_integral_planes_vs_reference_recording(depth):
    """integral.h (the two passes of integral_rows_kernel / integral_cols_kernel) against the integral planes the reference built
    for the recorded searches (tests/golden/me_full_d*.npz, x264_frame_filter): every entry whose whole box lies inside the padded
    plane below the first row (row 0 of the reference's buffer is its zero row, tests/test_me_full_vs_ref.py)."""
    L = _lib()
    fn = L.ii_host_u8 if depth == 8 else L.ii_host_u16
    fn.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    z = np.load(os.path.join(HERE, "golden", "me_full_d%d.npz" % depth))
    plane = np.ascontiguousarray(z["planes"][0])
    ph, pw = plane.shape
    ref = np.ascontiguousarray(z["integral"])          # [2*ph][pw]: 8x8 sums, then 4x4 sums
    s8 = np.zeros((ph, pw), np.uint16); s4 = np.zeros((ph, pw), np.uint16)
    fn(plane.ctypes.data, pw, pw, ph, s8.ctypes.data, s4.ctypes.data)
    assert np.array_equal(s8[1:ph - 8, :pw - 8], ref[1:ph - 8, :pw - 8])
    assert np.array_equal(s4[1:ph - 8, :pw - 8], ref[ph + 1:2 * ph - 8, :pw - 8])
    # and against plain box sums everywhere a box fits
    p = plane.astype(np.int64)
    c = np.zeros((ph + 1, pw + 1), np.int64); c[1:, 1:] = p.cumsum(0).cumsum(1)
    for n, got in ((8, s8), (4, s4)):
        box = (c[n:, n:] - c[:-n, n:] - c[n:, :-n] + c[:-n, :-n]) & 0xFFFF
        assert np.array_equal(got[:ph - n + 1, :pw - n + 1], box)


@pytest.mark.parametrize("depth", [8, 10])
def test_whole_frame_filter_order_vs_reference_recording(depth):
    """The composition x264hip_frame_filter runs on the device -- replicate the picture's border, hpel_filter over the picture plus
    8 samples all round, then every filtered sample outside [-4, W+3] x [-8, H+7] from the nearest one inside -- gives exactly the
    four padded planes the reference produced row by row (x264_frame_expand_border / x264_frame_filter /
    x264_frame_expand_border_filtered, recorded in the golden file).  hpel_filter itself: the oracle's (pinned against the vtable)."""
    z = np.load(os.path.join(HERE, "golden", "me_full_d%d.npz" % depth))
    W, H, pw, ph, padh, padv, _ = (int(v) for v in z["geom"])
    o = Oracle(depth)
    gold = z["planes"]
    pic = gold[0][padv:padv + H, padh:padh + W]
    luma = np.ascontiguousarray(np.pad(pic, ((padv, padv), (padh, padh)), mode="edge"))
    assert np.array_equal(luma, gold[0])
    f = o.f("hpel_filter")
    f.argtypes = [C.c_void_p] * 4 + [C.c_long, C.c_int, C.c_int, C.c_void_p]
    outs = [np.zeros_like(luma) for _ in range(3)]
    off = ((padv - 8) * pw + padh - 8) * luma.itemsize
    buf = np.zeros(W + 16 + 64, np.int16)
    f(outs[0].ctypes.data + off, outs[1].ctypes.data + off, outs[2].ctypes.data + off, luma.ctypes.data + off, pw, W + 16, H + 16, buf.ctypes.data)
    ys = np.clip(np.arange(-padv, H + padv), -8, H + 7) + padv
    xs = np.clip(np.arange(-padh, W + padh), -4, W + 3) + padh
    for k in range(3):
        assert np.array_equal(outs[k][np.ix_(ys, xs)], gold[k + 1]), k + 1
