"""Numpy models of two pieces of device arithmetic whose correctness does not follow from a parity run on random pictures:

* the SATD tail of x264_amd/csrc/device_common.h satd_partial_px4: coefficients carry a bias of 0x8000 so that the last butterfly, the
  absolute values and their sum are one v_sad_u16 (unsigned 16-bit absolute differences) -- checked against a plain 4x4 Hadamard at the
  extremes of 8- and 10-bit differences, where a wrap-around of the 16-bit lanes would show;
* the workgroup renumbering of device_common.h xcd_band_block (each XCD gets one contiguous run of the row-major order): it must be a
  bijection for every grid shape, or blocks of a field would be costed twice and others never."""
import numpy as np
import pytest


def _sad_u16(a, b):
    a = np.asarray(a, np.int64) & 0xFFFF
    b = np.asarray(b, np.int64) & 0xFFFF
    return np.abs(a - b)


def _satd4x4_device_model(d):
    """d: 4x4 differences (rows = the four lanes of a quad).  Returns the per-lane partial sums as the kernel forms them."""
    d = np.asarray(d, np.int64)
    rows = []
    for lane in range(4):
        r = d[lane].copy()
        if lane % 2 == 0:
            r[0] = (r[0] + 0x8000) & 0xFFFF  # the bias planted in sample 0 of the even rows (f.a ^ 0x8000 before the subtraction)
        # horizontal 4-point Hadamard, 16-bit wrap-around arithmetic
        s = [(r[0] + r[2]) & 0xFFFF, (r[1] + r[3]) & 0xFFFF]
        t = [(r[0] - r[2]) & 0xFFFF, (r[1] - r[3]) & 0xFFFF]
        rows.append(np.array([(s[0] + s[1]) & 0xFFFF, (s[0] - s[1]) & 0xFFFF, (t[0] + t[1]) & 0xFFFF, (t[0] - t[1]) & 0xFFFF], np.int64))
    rows = np.array(rows)
    # first vertical step: v' = partner + sign * v over lane ^ 1
    st1 = np.empty_like(rows)
    for lane in range(4):
        sign = -1 if lane & 1 else 1
        st1[lane] = (rows[lane ^ 1] + sign * rows[lane]) & 0xFFFF
    # second step + abs + sum: the lanes that would subtract send their value negated, v_sad_u16 of ( partner's message, own value )
    out = np.zeros(4, np.int64)
    for lane in range(4):
        partner = lane ^ 2
        sign_p = -1 if partner & 2 else 1
        msg = (sign_p * st1[partner]) & 0xFFFF
        out[lane] = _sad_u16(msg, st1[lane]).sum()
    return out


def _satd4x4_plain(d):
    h = np.array([[1, 1, 1, 1], [1, 1, -1, -1], [1, -1, -1, 1], [1, -1, 1, -1]], np.int64)
    return np.abs(h @ np.asarray(d, np.int64) @ h.T).sum()


@pytest.mark.parametrize("maxdiff", [255, 1023])
def test_biased_satd_tail_equals_the_hadamard_sum_at_the_extremes(maxdiff):
    rng = np.random.default_rng(5)
    cases = [np.full((4, 4), maxdiff), np.full((4, 4), -maxdiff), np.zeros((4, 4), np.int64)]
    signs = np.array([[1, 1, 1, 1], [1, 1, -1, -1], [1, -1, -1, 1], [1, -1, 1, -1]])
    for a in signs:       # every +-maxdiff pattern that drives one coefficient to its largest magnitude
        for b in signs:
            cases.append(np.outer(a, b) * maxdiff)
    cases += [rng.integers(-maxdiff, maxdiff + 1, (4, 4)) for _ in range(300)]
    for d in cases:
        assert _satd4x4_device_model(d).sum() == _satd4x4_plain(d), d


def _xcd_band_block(gx, gy, bx, by, bz=0):
    G, idx = gx * gy, by * gx + bx
    shift = (bz * G) & 7
    xcd = (idx + shift) & 7
    start = sum((G - ((j - shift) & 7) + 7) >> 3 for j in range(xcd))
    id2 = start + ((idx - ((xcd - shift) & 7)) >> 3)
    return id2 % gx, id2 // gx


@pytest.mark.parametrize("gx,gy", [(1, 1), (1, 7), (3, 5), (8, 8), (60, 135), (16, 271), (7, 9), (5, 1), (31, 2), (2, 1000)])
@pytest.mark.parametrize("bz", [0, 1, 2, 5, 13])
def test_xcd_band_renumbering_is_a_bijection_and_keeps_an_xcd_in_one_run(gx, gy, bz):
    """bz: the slice of a 3-D grid (multi-plane launches): the hardware deals workgroups to XCDs by their index over all three dimensions"""
    seen = {}
    for by in range(gy):
        for bx in range(gx):
            x, y = _xcd_band_block(gx, gy, bx, by, bz)
            assert 0 <= x < gx and 0 <= y < gy
            seen[(x, y)] = (bz * gx * gy + by * gx + bx) & 7   # the XCD this workgroup really runs on
    assert len(seen) == gx * gy
    # row-major order of the new positions: the XCD index never decreases (one contiguous run per XCD)
    order = [seen[(x, y)] for y in range(gy) for x in range(gx)]
    assert order == sorted(order)
