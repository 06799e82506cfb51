"""The block-search logic the device search kernel runs per 8-lane group (x264_amd/csrc/me_logic.h: x264_me_search_ref DIA / HEX +
refine_subpel as slicetype_mb_cost drives them, candidates costed one after the other and applied in the reference's order)
compiled for the host with a scalar evaluator (tests/tools/me_logic_host.cpp) and compared, vector for vector and cost for cost,
with the oracle's whole-field search, which is pinned against the reference.  Covers the neighbour/predictor list, the limits,
weights, lookahead bands and the unvisited edge ring; the device evaluator itself is covered by the -m gpu parity tests.
Every case also runs with an evaluator that reads the reference out of the STRIP copy of the planes through
x264_amd/csrc/strip_layout.h, the index arithmetic the kernels share: the layout is checked here as well."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle.oraclelib import Oracle, Weight
from tests.common import clip

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "tools", "me_logic_host.cpp")
OUT = os.path.join(HERE, "tools", "_build", "libme_logic_host.so")


_FUSED = [0]  # the build under test: me_logic.h's ME_FUSED_START (the default, 0, and the fused start set)


@pytest.fixture(params=[0, 1], ids=["plain-start", "fused-start"], autouse=True)
def _start_form(request):
    _FUSED[0] = request.param
    yield


def _lib():
    global OUT
    OUT = os.path.join(HERE, "tools", "_build", "libme_logic_host%s.so" % ("_fused" if _FUSED[0] else ""))
    hdr = os.path.join(ROOT, "x264_amd", "csrc", "me_logic.h")
    hdr2 = os.path.join(ROOT, "x264_amd", "csrc", "strip_layout.h")
    from oracle import oraclelib
    oraclelib.build()
    olib = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(OUT) or max(os.path.getmtime(SRC), os.path.getmtime(hdr), os.path.getmtime(hdr2), os.path.getmtime(olib)) > os.path.getmtime(OUT):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-DME_FUSED_START=%d" % _FUSED[0], "-I" + os.path.join(ROOT, "x264_amd", "csrc"), "-I" + os.path.join(ROOT, "oracle"),
                               "-o", OUT, SRC, olib, "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return C.CDLL(OUT)


CONFIGS = {
    # name: (depth, me_method, subpel_refine, me_range, subme, mbcmp_satd, fpelcmp_satd)
    "hex_r4": (8, 1, 4, 16, 7, 1, 0),
    "dia_r4": (8, 0, 4, 16, 8, 1, 0),
    "hex_r32": (8, 1, 4, 32, 9, 1, 0),
    "dia_r2_sad": (8, 0, 2, 16, 1, 0, 0),
    "dia_r4_subme2": (8, 0, 4, 16, 2, 1, 0),
    "hex_10bit_tesa": (10, 1, 4, 24, 10, 1, 1),
    "hex_10bit": (10, 1, 4, 16, 7, 1, 0),
    "hex_r2_sad": (8, 1, 2, 16, 1, 0, 0),
}


def _field(L, o, cfg, fenc, ref, wt=None, wplane=None, strips=False):
    n = cfg.mb_w * cfg.mb_h
    mvs = np.zeros((n, 2), np.int16)
    costs = np.zeros(n, np.int32)
    evals = np.zeros(2, np.int64)
    fn = getattr(L, "mel%d_search_field%s" % (o.d, "_strips" if strips else ""))  # _strips: reference samples out of the strip copy
    fn.restype = None
    fn(C.byref(cfg), C.c_void_p(o.origin(fenc[0])), o._plane_ptrs(ref), C.c_void_p(o.origin(wplane)) if wplane is not None else None,
       C.byref(wt) if wt is not None else None, mvs.ctypes.data_as(C.c_void_p), costs.ctypes.data_as(C.c_void_p), evals.ctypes.data_as(C.c_void_p))
    return mvs, costs, evals


@pytest.mark.parametrize("cfgname", list(CONFIGS))
@pytest.mark.parametrize("clipname", ["pan", "fastpan", "noise", "static", "fade"])
def test_logic_matches_oracle_field(cfgname, clipname):
    depth, me, refine, me_range, subme, mbcmp, fpelcmp = CONFIGS[cfgname]
    L = _lib()
    o = Oracle(depth)
    W, H = (352, 288) if clipname == "pan" else (176, 144)
    frames = clip(clipname, W, H, 3, depth)
    cfg = o.make_cfg((W + 15) // 16, (H + 15) // 16, me_method=me, subpel_refine=refine, me_range=me_range, mv_range=128, subme=subme,
                     mbcmp_satd=mbcmp, fpelcmp_satd=fpelcmp)
    pl = [o.lowres_init(cfg, f) for f in frames]
    for (b, r) in ((1, 0), (2, 0), (0, 2)):
        wm, wc = o.search_field(cfg, pl[b], pl[r])
        for strips in (False, True):
            m, c, ev = _field(L, o, cfg, pl[b], pl[r], strips=strips)
            assert np.array_equal(m, wm), (b, r, strips, int((m != wm).any(1).sum()))
            assert np.array_equal(c, wc), (b, r, strips)
            assert ev.sum() > 0 or clipname == "static"


@pytest.mark.parametrize("cfgname", ["hex_r4", "dia_r2_sad", "hex_10bit"])
def test_logic_with_weights(cfgname):
    depth, me, refine, me_range, subme, mbcmp, fpelcmp = CONFIGS[cfgname]
    L = _lib()
    o = Oracle(depth)
    frames = clip("fade", 176, 144, 3, depth)
    cfg = o.make_cfg(11, 9, me_method=me, subpel_refine=refine, me_range=me_range, mv_range=128, subme=subme, mbcmp_satd=mbcmp, fpelcmp_satd=fpelcmp)
    pl = [o.lowres_init(cfg, f) for f in frames]
    for wt in ((1, 55, 6, 3), (1, 100, 7, -4), (1, 3, 0, -2)):
        w = Weight(*wt)
        wp = o.weight_plane(cfg, pl[0][0], w)
        wm, wc = o.search_field(cfg, pl[2], pl[0], w, wp)
        for strips in (False, True):
            m, c, _ = _field(L, o, cfg, pl[2], pl[0], w, wp, strips=strips)
            assert np.array_equal(m, wm) and np.array_equal(c, wc), (wt, strips)


@pytest.mark.parametrize("n_slices,do_edges", [(3, 1), (1, 0), (4, 0), (16, 1)])
def test_logic_bands_and_edge_ring(n_slices, do_edges):
    L = _lib()
    o = Oracle(8)
    for (W, H) in ((352, 288), (48, 32), (32, 48)):
        frames = clip("fastpan", W, H, 2)
        cfg = o.make_cfg((W + 15) // 16, (H + 15) // 16, me_method=1, subpel_refine=4, me_range=16, mv_range=128, subme=7, mbcmp_satd=1, fpelcmp_satd=0,
                         n_slices=n_slices, do_edges=do_edges)
        pl = [o.lowres_init(cfg, f) for f in frames]
        wm, wc = o.search_field(cfg, pl[1], pl[0])
        for strips in (False, True):
            m, c, _ = _field(L, o, cfg, pl[1], pl[0], strips=strips)
            assert np.array_equal(m, wm) and np.array_equal(c, wc), (W, H, strips)
