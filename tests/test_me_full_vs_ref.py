"""SURVEY 8(f) rank 3 groundwork: the oracle's restatement of the main-encode motion search (x264_me_search_ref with DIA / HEX /
UMH / ESA / TESA + refine_subpel, any partition size) against the real reference, driven block by block through
oracle/ref_harness.c:rh_me_search on reference frames prepared like reconstructed frames (x264_frame_filter)."""
import ctypes as C

import numpy as np
import pytest

from oracle import refharness
from oracle.oraclelib import Oracle
from x264_amd.synth import make_clip

pytestmark = pytest.mark.skipif(not refharness.available(8), reason="oracle/_ref not built (no /root/reference)")

from tests.common import ME_METHODS as METHODS, ME_SIZES as SIZES, oracle_me_search  # noqa: E402


def _box_sums(plane, n):
    """(sum over an n x n box with top-left at each sample) mod 2^16, zero where the box leaves the array"""
    p = plane.astype(np.int64)
    c = np.zeros((p.shape[0] + 1, p.shape[1] + 1), np.int64)
    c[1:, 1:] = p.cumsum(0).cumsum(1)
    out = np.zeros(p.shape, np.int64)
    hh, ww = p.shape[0] - n + 1, p.shape[1] - n + 1
    out[:hh, :ww] = c[n:n + hh, n:n + ww] - c[:hh, n:n + ww] - c[n:n + hh, :ww] + c[:hh, :ww]
    return (out & 0xFFFF).astype(np.uint16)


CLIPS = {"pan": dict(pan=(7, -5), noise=6, texture=0.5), "noise": dict(pan=(1, 0), noise=40, texture=0.9),
         "fastpan": dict(pan=(23, 11), noise=2, texture=0.3)}


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("clipname", list(CLIPS))
@pytest.mark.parametrize("me", list(METHODS))
def test_me_search_full(me, clipname, depth):
    W, H = 176, 144
    o = Oracle(depth)
    fr = make_clip(W, H, 2, seed=31 + depth, bit_depth=depth, **CLIPS[clipname])
    r = refharness.Ref(W, H, "medium", opts="me=%s,partitions=all,merange=24" % me, bit_depth=depth)
    try:
        L = r.lib
        L.rh_add_ref_frame.argtypes = [C.c_void_p, C.c_void_p]
        L.rh_me_search.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.rh_get_ref_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.rh_get_integral.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        ref = np.ascontiguousarray(fr[0])
        idx = L.rh_add_ref_frame(r.ctx, ref.ctypes.data)
        assert idx == 0
        geo = (C.c_int * 8)()
        L.rh_ref_geometry(r.ctx, geo)
        w, lines, rstride, padh, padv, has_int, padh_align, sub8 = list(geo)
        pw, ph = w + 2 * padh, lines + 2 * padv
        planes = []
        for p in range(4):
            a = np.zeros((ph, pw), o.dtype)
            L.rh_get_ref_plane(r.ctx, 0, p, a.ctypes.data)
            planes.append(a)
        # the half-pel planes are what the oracle's hpel_filter gives for the padded luma (inside the picture area)
        cost_mv = np.ascontiguousarray(r.cost_mv())
        centre = (cost_mv.size - 1) // 2
        integral = None
        if has_int:
            # the reference's integral planes against plain box sums of the padded luma plane
            rows = 2 * ph
            raw = np.zeros(rows * rstride, np.uint16)
            assert L.rh_get_integral(r.ctx, 0, raw.ctypes.data, raw.size) > 0
            raw = raw.reshape(rows, rstride)
            # row y of the upper plane is picture row y - padv; column x is picture column x - padh_align
            box8 = _box_sums(planes[0], 8)
            x0 = padh_align - padh
            # (row 0 of the buffer is the zero row the running sums start from: the first box row is row 1)
            assert np.array_equal(raw[1:ph - 8, x0:x0 + pw - 8], box8[1:ph - 8, :pw - 8])
            if sub8:
                box4 = _box_sums(planes[0], 4)
                assert np.array_equal(raw[ph + 1:ph + ph - 8, x0:x0 + pw - 8], box4[1:ph - 8, :pw - 8])
            integral = raw
        rng = np.random.default_rng(5)
        mv_range = r.cfg["mv_range"]
        mbw, mbh = W // 16, H // 16
        itg = None
        if integral is not None:
            # the oracle walks the integral plane with the reference plane's stride: hand it a view with the same geometry
            x0 = padh_align - padh
            itg = np.ascontiguousarray(integral[:, x0:x0 + pw])
        n_checked = 0
        for trial in range(150):
            i_pixel = int(rng.integers(0, 7)) if sub8 or METHODS[me] < 3 else int(rng.integers(0, 4))
            bw, bh = SIZES[i_pixel]
            mb_x, mb_y = int(rng.integers(0, mbw)), int(rng.integers(0, mbh))
            xoff = int(rng.integers(0, 16 // bw)) * bw
            yoff = int(rng.integers(0, 16 // bh)) * bh
            subme = int(rng.choice([1, 2, 3, 5, 7, 9]))
            me_range = int(rng.choice([8, 16, 24]))
            fenc = np.zeros((16, 16), o.dtype)
            sy, sx = 16 * mb_y + yoff, 16 * mb_x + xoff
            blk = fr[1][sy:sy + bh, sx:sx + bw]
            fenc[:blk.shape[0], :blk.shape[1]] = blk
            mvp = np.array(rng.integers(-100, 101, size=2) if trial % 3 else [0, 0], np.int16)
            n_mvc = int(rng.integers(0, 5))
            mvc = np.ascontiguousarray(rng.integers(-120, 121, size=(max(n_mvc, 1), 2)).astype(np.int16))
            if n_mvc and trial % 5 == 1:
                mvc[0] = mvp  # a candidate equal to the predictor is dropped
            if n_mvc > 1 and trial % 4 == 0:
                mvc[1] = 0
            out_r = np.zeros(4, np.int32)
            assert L.rh_me_search(r.ctx, 0, fenc.ctypes.data, mb_x, mb_y, xoff, yoff, i_pixel, subme, me_range, mvp.ctypes.data,
                                  mvc.ctypes.data, n_mvc, out_r.ctypes.data) == 0
            call = [i_pixel, mb_x, mb_y, xoff, yoff, subme, me_range, int(mvp[0]), int(mvp[1]), n_mvc] + \
                   (list(mvc.reshape(-1)) + [0] * 8)[:8]
            out_o = oracle_me_search(o, me, planes, itg, cost_mv, (W, H, pw, ph, padh, padv, mv_range), fenc, call)
            if subme < 2:
                out_r[3] = out_o[3]  # cost_mv is only defined by refine_subpel / the subme < 3 exit
            assert np.array_equal(out_r[:3], out_o[:3]), "%s trial %d pix %d subme %d range %d mvp %s nmvc %d mb %d,%d off %d,%d ref %s oracle %s" % (me, trial, i_pixel, subme, me_range, mvp.tolist(), n_mvc, mb_x, mb_y, xoff, yoff, out_r.tolist(), out_o.tolist())
            if subme >= 2:
                assert out_r[3] == out_o[3]
            n_checked += 1
        assert n_checked == 150
    finally:
        r.close()
