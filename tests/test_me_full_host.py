"""The main-encode motion search of x264_amd/csrc/me_full.h (the body of me_full_kernel: x264_me_search_ref with DIA / HEX / UMH /
ESA / TESA + refine_subpel, one thread per request) compiled for the host by tests/tools/block_metrics_host.cpp, against the
results recorded from the reference (tests/golden/me_full_d{8,10}.npz, 2 x 600 calls over all methods, partition sizes and subme
levels).  The GPU test of the same code is tests/test_gpu_configs.py::test_me_search_batch."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.common import ME_METHODS, ME_SIZES
from tests.test_block_metrics_host import _lib

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class MfHostReq(C.Structure):
    _fields_ = [("i_pixel", C.c_int), ("me_method", C.c_int), ("subpel_refine", C.c_int), ("me_range", C.c_int), ("mbcmp_satd", C.c_int),
                ("fpelcmp_satd", C.c_int), ("fenc", C.c_void_p), ("fenc_stride", C.c_int), ("ref", C.c_void_p * 4), ("stride", C.c_int),
                ("integral", C.c_void_p), ("integral_lower", C.c_long), ("mvp", C.c_int * 2), ("lim_min", C.c_int * 2),
                ("lim_max", C.c_int * 2), ("spel_min", C.c_int * 2), ("spel_max", C.c_int * 2), ("cost_mv", C.c_void_p)]


def request_geometry(geom, call):
    """limits of one recorded call exactly as tests/common.py:oracle_me_search derives them (analyse.c:333,348-349)"""
    W, H, pw, ph, padh, padv, mv_range = (int(v) for v in geom)
    i_pixel, mb_x, mb_y, xoff, yoff = (int(v) for v in call[:5])
    mbw, mbh = W // 16, H // 16
    fm = 4 * mv_range
    smin = [max(4 * (-16 * mb_x - 24), -fm), max(4 * (-16 * mb_y - 24), -fm)]
    smax = [min(4 * (16 * (mbw - mb_x - 1) + 24), fm - 1), min(4 * (16 * (mbh - mb_y - 1) + 24), fm - 1)]
    lim_min = [(smin[k] >> 2) + 6 for k in range(2)]
    lim_max = [(smax[k] >> 2) - 6 for k in range(2)]
    sy, sx = 16 * mb_y + yoff, 16 * mb_x + xoff
    return smin, smax, lim_min, lim_max, sx, sy, (padv + sy) * pw + padh + sx


@pytest.mark.parametrize("depth", [8, 10])
def test_me_search_full_host_vs_golden(depth):
    L = _lib()
    fn = L.mf_host_u8 if depth == 8 else L.mf_host_u16
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    z = np.load(os.path.join(GOLD, "me_full_d%d.npz" % depth))
    dt = np.uint8 if depth == 8 else np.uint16
    planes = [np.ascontiguousarray(z["planes"][p]) for p in range(4)]
    integral = np.ascontiguousarray(z["integral"])
    cost_mv = np.ascontiguousarray(z["cost_mv"])
    centre = (cost_mv.size - 1) // 2
    frame = np.ascontiguousarray(z["fenc_frame"], dt)
    W, H, pw, ph, padh, padv, mv_range = (int(v) for v in z["geom"])
    n = 0
    for me in ME_METHODS:
        for call in z["calls_%s" % me]:
            i_pixel, mb_x, mb_y, xoff, yoff, subme, me_range, mvpx, mvpy, n_mvc = (int(v) for v in call[:10])
            smin, smax, lim_min, lim_max, sx, sy, org = request_geometry(z["geom"], call)
            m = MfHostReq()
            m.i_pixel, m.me_method, m.subpel_refine, m.me_range = i_pixel, ME_METHODS[me], subme, me_range
            m.mbcmp_satd, m.fpelcmp_satd = 1, int(me == "tesa")
            m.fenc = frame.ctypes.data + (sy * frame.shape[1] + sx) * frame.itemsize   # straight from the source plane
            m.fenc_stride = frame.shape[1]
            for p in range(4):
                m.ref[p] = planes[p].ctypes.data + org * planes[p].itemsize
            m.stride = pw
            for k in range(2):
                m.spel_min[k], m.spel_max[k], m.lim_min[k], m.lim_max[k] = smin[k], smax[k], lim_min[k], lim_max[k]
            m.mvp[0], m.mvp[1] = mvpx, mvpy
            m.cost_mv = cost_mv.ctypes.data + 2 * centre
            if ME_METHODS[me] >= 3:
                m.integral = integral.ctypes.data + org * 2
                m.integral_lower = ph * pw
            mvc = np.ascontiguousarray(np.array(call[10:18], np.int16).reshape(4, 2))
            out = np.zeros(4, np.int32)
            fn(C.byref(m), mvc.ctypes.data, n_mvc, out.ctypes.data)
            want = call[18:22]
            k = 4 if subme >= 2 else 3
            assert np.array_equal(out[:k], want[:k]), (me, call.tolist(), out.tolist())
            n += 1
    assert n >= 500


@pytest.mark.parametrize("depth", [8, 10])
def test_thread_group_form_equals_one_thread_form(depth):
    """me_full.h's cooperative code -- the chunked exhaustive scans, the ordered compaction of the ads survivors, the SAD stage's running
    thresholds as a prefix minimum, the survivor thinning -- run by EIGHT host threads standing for the lanes of a wave (collectives through
    a shared slot array between barriers, tests/tools/block_metrics_host.cpp) against the one-thread form, on the recorded ESA / TESA calls
    and on random requests of every method; all eight threads must also end in the same state.  (The device's 64-lane form is held against
    the one-thread form by tests/test_gpu_configs.py::test_searches_wave_form_equals_scalar_form.)"""
    L = _lib()
    one = L.mf_host_u8 if depth == 8 else L.mf_host_u16
    grp = L.mf_host_lanes_u8 if depth == 8 else L.mf_host_lanes_u16
    one.argtypes = grp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    grp.restype = C.c_int
    z = np.load(os.path.join(GOLD, "me_full_d%d.npz" % depth))
    dt = np.uint8 if depth == 8 else np.uint16
    planes = [np.ascontiguousarray(z["planes"][p]) for p in range(4)]
    integral = np.ascontiguousarray(z["integral"])
    cost_mv = np.ascontiguousarray(z["cost_mv"])
    centre = (cost_mv.size - 1) // 2
    frame = np.ascontiguousarray(z["fenc_frame"], dt)
    W, H, pw, ph, padh, padv, mv_range = (int(v) for v in z["geom"])
    rng = np.random.default_rng(77 + depth)
    calls = [(ME_METHODS[me], [int(v) for v in call[:10]], np.array(call[10:18], np.int16).reshape(4, 2)) for me in ("esa", "tesa") for call in z["calls_%s" % me]]
    for t in range(180):
        i_pixel = int(rng.integers(0, 7))
        bw, bh = ME_SIZES[i_pixel]
        mb_x, mb_y = int(rng.integers(0, W // 16)), int(rng.integers(0, H // 16))
        xoff, yoff = int(rng.integers(0, 16 // bw)) * bw, int(rng.integers(0, 16 // bh)) * bh
        smin, smax, _, _, _, _, _ = request_geometry(z["geom"], [i_pixel, mb_x, mb_y, xoff, yoff])
        mvp = [int(rng.integers(smin[k], smax[k] + 1)) for k in range(2)]
        n_mvc = int(rng.integers(0, 5))
        mvc = np.zeros((4, 2), np.int16)
        for i in range(n_mvc):
            mvc[i] = [int(rng.integers(smin[k] - 8, smax[k] + 9)) for k in range(2)]
        me = (3, 4, 4, 3, 2, 1, 0)[t % 7]
        calls.append((me, [i_pixel, mb_x, mb_y, xoff, yoff, int(rng.choice([1, 2, 5, 7])), int(rng.choice([8, 16, 24])), mvp[0], mvp[1], n_mvc], mvc))
    for me, call, mvc in calls:
        i_pixel, mb_x, mb_y, xoff, yoff, subme, me_range, mvpx, mvpy, n_mvc = call
        smin, smax, lim_min, lim_max, sx, sy, org = request_geometry(z["geom"], call)
        m = MfHostReq()
        m.i_pixel, m.me_method, m.subpel_refine, m.me_range = i_pixel, me, subme, me_range
        m.mbcmp_satd, m.fpelcmp_satd = 1, int(me == 4)
        m.fenc = frame.ctypes.data + (sy * frame.shape[1] + sx) * frame.itemsize
        m.fenc_stride = frame.shape[1]
        for p in range(4):
            m.ref[p] = planes[p].ctypes.data + org * planes[p].itemsize
        m.stride = pw
        for k in range(2):
            m.spel_min[k], m.spel_max[k], m.lim_min[k], m.lim_max[k] = smin[k], smax[k], lim_min[k], lim_max[k]
        m.mvp[0], m.mvp[1] = mvpx, mvpy
        m.cost_mv = cost_mv.ctypes.data + 2 * centre
        m.integral = integral.ctypes.data + org * 2
        m.integral_lower = ph * pw
        mvc = np.ascontiguousarray(mvc)
        a, b = np.zeros(4, np.int32), np.zeros(4, np.int32)
        one(C.byref(m), mvc.ctypes.data, n_mvc, a.ctypes.data)
        same = grp(C.byref(m), mvc.ctypes.data, n_mvc, b.ctypes.data)
        assert same == 1, ("the threads of the group ended in different states", me, call)
        k = 4 if subme >= 2 else 3
        assert np.array_equal(a[:k], b[:k]), (me, call, a.tolist(), b.tolist())
