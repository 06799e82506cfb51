"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

from oracle.oraclelib import Oracle
from x264_amd.synth import make_clip

CLIPS = {
    # name: kwargs for make_clip
    "pan": dict(seed=3),
    "fastpan": dict(seed=5, pan=(17, -9), noise=9, texture=0.35),
    "noise": dict(seed=7, pan=(0, 0), noise=60, texture=0.9),
    "static": dict(seed=9, pan=(0, 0), noise=0, texture=0.05),
    "fade": dict(seed=11, fade=(1, 3, 0.55, 12.0)),
}


def clip(name, w, h, n, depth=8):
    return make_clip(w, h, n, bit_depth=depth, **CLIPS[name])


def oracle_cfg(o, rcfg, cost_mv=None):
    """Build the oracle's lookahead config from the reference's validated parameters."""
    return o.make_cfg(rcfg["mb_w"], rcfg["mb_h"], me_method=rcfg["me_method"], subpel_refine=rcfg["subpel_refine"],
                      me_range=rcfg["me_range"], mv_range=rcfg["mv_range"], subme=rcfg["subme"],
                      mbcmp_satd=rcfg["mbcmp_satd"], fpelcmp_satd=rcfg["fpelcmp_satd"],
                      weighted_bipred=rcfg["weighted_bipred"], aq_mode=rcfg["aq_mode"], lam=rcfg["lambda"],
                      bframe_bias=rcfg["b_bias"], cost_mv=cost_mv, n_slices=rcfg.get("lookahead_threads", 1),
                      do_edges=int(bool(rcfg["mb_tree"] or rcfg["vbv"])))


# ---- main-encode motion search (SURVEY 8f rank 3 groundwork): shared between the reference test and the golden test ----
import ctypes as _C

ME_SIZES = [(16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)]
ME_METHODS = {"dia": 0, "hex": 1, "umh": 2, "esa": 3, "tesa": 4}


class MeFull(_C.Structure):
    """or{8,10}_me_full of oracle/x264_oracle.h"""
    _fields_ = [("i_pixel", _C.c_int), ("me_method", _C.c_int), ("subpel_refine", _C.c_int), ("me_range", _C.c_int),
                ("mbcmp_satd", _C.c_int), ("fpelcmp_satd", _C.c_int), ("fenc", _C.c_void_p), ("ref", _C.c_void_p * 4), ("stride", _C.c_int),
                ("integral", _C.c_void_p), ("integral_lower", _C.c_long), ("mvp", _C.c_int * 2), ("lim_min", _C.c_int * 2),
                ("lim_max", _C.c_int * 2), ("spel_min", _C.c_int * 2), ("spel_max", _C.c_int * 2), ("cost_mv", _C.c_void_p)]


def oracle_me_search(o, me, planes, integral, cost_mv, geom, fenc, call):
    """One x264_me_search_ref-style call through the oracle.  planes: [4][ph][pw] padded half-pel planes, integral: [2*ph][pw]
    uint16 or None, cost_mv: centred table, geom: (W, H, pw, ph, padh, padv, mv_range), fenc: 16x16 block buffer,
    call: (i_pixel, mb_x, mb_y, xoff, yoff, subme, me_range, mvpx, mvpy, n_mvc, mvc[8]).  Returns [mvx, mvy, cost, cost_mv]."""
    import numpy as np
    W, H, pw, ph, padh, padv, mv_range = (int(v) for v in geom)
    i_pixel, mb_x, mb_y, xoff, yoff, subme, me_range, mvpx, mvpy, n_mvc = (int(v) for v in call[:10])
    mvc = np.ascontiguousarray(np.array(call[10:18], np.int16).reshape(4, 2))
    mbw, mbh = W // 16, H // 16
    m = MeFull()
    m.i_pixel, m.me_method, m.subpel_refine, m.me_range = i_pixel, ME_METHODS[me], subme, me_range
    m.mbcmp_satd, m.fpelcmp_satd = 1, int(me == "tesa")
    m.fenc = fenc.ctypes.data
    sy, sx = 16 * mb_y + yoff, 16 * mb_x + xoff
    org = (padv + sy) * pw + padh + sx
    for p in range(4):
        m.ref[p] = planes[p].ctypes.data + org * planes[p].itemsize
    m.stride = pw
    fm = 4 * mv_range
    smin = [max(4 * (-16 * mb_x - 24), -fm), max(4 * (-16 * mb_y - 24), -fm)]
    smax = [min(4 * (16 * (mbw - mb_x - 1) + 24), fm - 1), min(4 * (16 * (mbh - mb_y - 1) + 24), fm - 1)]
    for k in range(2):
        m.spel_min[k], m.spel_max[k] = smin[k], smax[k]
        m.lim_min[k], m.lim_max[k] = (smin[k] >> 2) + 6, (smax[k] >> 2) - 6   # i_fpel_border, analyse.c:333,348-349
    m.mvp[0], m.mvp[1] = mvpx, mvpy
    centre = (cost_mv.size - 1) // 2
    m.cost_mv = cost_mv.ctypes.data + 2 * centre
    if integral is not None:
        m.integral = integral.ctypes.data + org * 2
        m.integral_lower = ph * pw
    out = np.zeros(4, np.int32)
    f = o.f("me_search_full")
    f.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_int, _C.c_void_p]
    f(_C.byref(m), mvc.ctypes.data, n_mvc, out.ctypes.data)
    return out


def nearest_ref_cells(idx, typ):
    """(b-p0, p1-b) of every frame of a coded-order sequence as the encoder's fref_nearest gives them (x264_rc_analyse_slice,
    slicetype.c:1982-1991): distances to the nearest already coded reference before and after the frame in display order.
    Independent of the host logic's own bookkeeping (which is checked against it)."""
    cells, refs = [], []
    for f, t in zip(idx, typ):
        f, t = int(f), int(t)
        if t in (4, 5):
            cells.append((f - max(r for r in refs if r < f), min(r for r in refs if r > f) - f))
        elif t in (1, 2):
            cells.append((0, 0))
        else:
            cells.append((f - max(r for r in refs if r < f), 0))
        if t != 5:
            refs.append(f)
    return cells
