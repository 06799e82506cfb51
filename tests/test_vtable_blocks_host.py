"""The arithmetic of x264_amd/csrc/vtable_blocks.h -- the device code behind x264hip_dct_batch / quant_batch / var2_batch / ads_batch
(every remaining entry of x264_dct_function_t and x264_quant_function_t, var2 and ads of x264_pixel_function_t) -- compiled for the
host (tests/tools/vtable_blocks_host.cpp) and compared with the oracle, which tests/test_primitives_vs_ref.py pins against the
reference's own vtables on checkasm-style inputs (tools/checkasm.c:890-1224).  The GPU test of the same entry points is
tests/test_gpu_primitives.py::test_vtable_batches."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle.oraclelib import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "tools", "vtable_blocks_host.cpp")
OUT = os.path.join(HERE, "tools", "_build", "libvtb_host.so")
DCT_COEFS = {0: 16, 1: 64, 2: 256, 3: 64, 4: 256, 5: 4, 6: 8, 7: 16, 8: 8}
QUANT_COEFS = {0: 16, 1: 64, 2: 64, 3: 16, 4: 4}


def _lib():
    hdrs = [os.path.join(ROOT, "x264_amd", "csrc", h) for h in ("vtable_blocks.h", "dct_quant_block.h")]
    if not os.path.exists(OUT) or max([os.path.getmtime(SRC)] + [os.path.getmtime(h) for h in hdrs]) > os.path.getmtime(OUT):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "x264_amd", "csrc"), "-o", OUT, SRC])
    return C.CDLL(OUT)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def patterns(rng, shape, dtype, maxv):
    """checkasm-style inputs: random, all-max against all-zero (largest differences), near-equal"""
    yield rng.integers(0, maxv + 1, size=shape).astype(dtype)
    yield np.full(shape, maxv, dtype)
    yield np.zeros(shape, dtype)


@pytest.mark.parametrize("depth", [8, 10])
def test_dct_kinds(depth):
    L = _lib(); o = Oracle(depth)
    rng = np.random.default_rng(depth)
    maxv = (1 << depth) - 1
    fn = L.vtb_dct_u8 if depth == 8 else L.vtb_dct_u16
    for fenc in patterns(rng, (16, 16), o.dtype, maxv):
        for fdec in patterns(rng, (16, 32), o.dtype, maxv):
            for kind in range(7):
                a = np.zeros(DCT_COEFS[kind], o.coef_dtype); b = np.zeros_like(a)
                fn(kind, _p(a), _p(fenc), _p(fdec))
                o.f("dct")(kind, _p(b), _p(fenc), _p(fdec))
                assert np.array_equal(a, b), kind
    for kind in (7, 8):
        for amp in (40000 if depth == 10 else 8000, 300, 2):
            a = rng.integers(-amp, amp + 1, size=DCT_COEFS[kind]).astype(o.coef_dtype); b = a.copy()
            fn(kind, _p(a), None, None)
            o.f("dct")(kind, _p(b), None, None)
            assert np.array_equal(a, b), kind


@pytest.mark.parametrize("depth", [8, 10])
def test_quant_kinds(depth):
    L = _lib(); o = Oracle(depth)
    rng = np.random.default_rng(10 + depth)
    fn = L.vtb_quant_u8 if depth == 8 else L.vtb_quant_u16
    lim = 30000 if depth == 8 else 1 << 20
    for kind, n in QUANT_COEFS.items():
        nt = 64 if kind == 1 else 16
        for trial in range(20):
            mf = rng.integers(1, 30000 if depth == 8 else 1 << 18, size=nt).astype(o.ucoef_dtype)
            bias = rng.integers(0, 30000, size=nt).astype(o.ucoef_dtype)
            for amp in (lim, 300, 3):
                a = rng.integers(-amp, amp + 1, size=n).astype(o.coef_dtype); b = a.copy()
                ra = fn(kind, _p(a), _p(mf), _p(bias), int(mf[0]) >> 1, int(bias[0]) << 1)
                rb = o.f("quant", C.c_int)(kind, _p(b), _p(mf), _p(bias), int(mf[0]) >> 1, int(bias[0]) << 1)
                assert ra == rb and np.array_equal(a, b), (kind, trial, amp)


@pytest.mark.parametrize("depth", [8, 10])
def test_var2_and_ads(depth):
    L = _lib(); o = Oracle(depth)
    rng = np.random.default_rng(20 + depth)
    maxv = (1 << depth) - 1
    fn = L.vtb_var2_u8 if depth == 8 else L.vtb_var2_u16
    for fenc in patterns(rng, (16, 16), o.dtype, maxv):
        for fdec in patterns(rng, (16, 32), o.dtype, maxv):
            for h in (8, 16):
                sa = np.zeros(2, np.int32); sb = np.zeros(2, np.int32)
                ra = fn(_p(fenc), _p(fdec), h, _p(sa))
                rb = o.f("var2", C.c_int)(_p(fenc), _p(fdec), h, _p(sb))
                assert ra == rb and np.array_equal(sa, sb), h
    f = o.f("ads", C.c_int)
    for i in range(60):
        n_dc = (1, 2, 4)[i % 3]
        sums = rng.integers(0, 1 << 16, size=600).astype(np.uint16)
        dc = rng.integers(0, 1 << 16, size=4).astype(np.int32)
        cost = rng.integers(0, 200, size=64).astype(np.uint16)
        thresh = int(rng.integers(1000, 120000))
        ma = np.zeros(64, np.int16); mb = np.zeros(64, np.int16)
        na = L.vtb_ads(n_dc, _p(dc), _p(sums), 32, _p(cost), _p(ma), 60, thresh)
        nb = f(n_dc, _p(dc), _p(sums), 32, _p(cost), _p(mb), 60, thresh)
        assert na == nb and np.array_equal(ma[:na], mb[:nb]), (i, n_dc)
