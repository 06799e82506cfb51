"""GPU parity: the HIP path (through the C ABI, x264_amd/libx264hip.so) against the CPU oracle on the
same seeded inputs.  Integer work: bit-exact."""
import numpy as np
import pytest

from oracle.oraclelib import Oracle, PAD, Weight
from tests.common import clip
from x264_amd import lib

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: (depth, me_method, subpel_refine, me_range, subme, mbcmp_satd, fpelcmp_satd, bframes)
    "hex_r4": (8, 1, 4, 16, 7, 1, 0, 3),
    "dia_r4": (8, 0, 4, 16, 8, 1, 0, 3),
    "hex_r32": (8, 1, 4, 32, 9, 1, 0, 3),
    "dia_r2_sad": (8, 0, 2, 16, 1, 0, 0, 3),
    "dia_r4_subme2": (8, 0, 4, 16, 2, 1, 0, 3),
    "hex_10bit_tesa": (10, 1, 4, 24, 10, 1, 1, 8),
    "hex_10bit": (10, 1, 4, 16, 7, 1, 0, 3),
}
SEQ = [(0, 0, 0), (0, 1, 1), (0, 2, 2), (0, 2, 1), (1, 1, 1), (0, 3, 3), (0, 3, 1), (0, 3, 2), (1, 3, 2), (2, 3, 3), (3, 3, 3),
       (1, 2, 2)]


def _mk(depth, me_method, subpel_refine, me_range, subme, mbcmp_satd, fpelcmp_satd, bframes, W, H, mv_range=128):
    o = Oracle(depth)
    cfg = o.make_cfg((W + 15) // 16, (H + 15) // 16, me_method=me_method, subpel_refine=subpel_refine, me_range=me_range,
                     mv_range=mv_range, subme=subme, mbcmp_satd=mbcmp_satd, fpelcmp_satd=fpelcmp_satd)
    ctx = lib.Context(W, H, bit_depth=depth, bframes=bframes, me_method=me_method, subpel_refine=subpel_refine,
                      me_range=me_range, mv_range=mv_range, subme=subme, mbcmp_satd=mbcmp_satd, fpelcmp_satd=fpelcmp_satd,
                      max_frames=8, cost_mv=o._cost_mv)
    return o, cfg, ctx


def _run_sequence(o, cfg, ctx, frames, weights=None, prefetch=False):
    """Replays SEQ with reference first-trigger semantics on the HIP context and through the oracle's pure
    functions, comparing every produced array."""
    nf = len(frames)
    planes, inv, intra = [], [], []
    for i in range(nf):
        ctx.frame_put(i, frames[i])
        pl = o.lowres_init(cfg, frames[i])
        for p in range(4):
            assert np.array_equal(ctx.lowres(i, p), pl[p][:, :8 * cfg.mb_w + 2 * PAD]), ("lowres", i, p)
        iq, qp, s, ssd = o.aq_frame(frames[i], cfg.mb_w, cfg.mb_h, 1, 1.0)
        assert np.array_equal(ctx.inv_qscale(i), iq), ("inv_qscale", i)
        assert np.array_equal(ctx.qp_offsets(i), qp), ("f_qp_offset_aq", i)  # FP32, same operation order
        assert ctx.frame_stats(i) == (s, ssd)
        ic = o.intra_costs(cfg, pl)
        assert np.array_equal(ctx.intra_costs(i), ic), ("intra", i, int((ctx.intra_costs(i) != ic).sum()))
        planes.append(pl); inv.append(iq); intra.append(ic)
    if prefetch:
        ctx.prefetch(list(range(nf)), list(range(nf)))
    fields = {}
    intra_done = set()
    for (p0, p1, b) in SEQ:
        if max(p0, p1, b) >= nf:
            continue
        d0, d1 = b - p0, p1 - b
        with_intra = b not in intra_done
        if p0 == p1:
            out = ctx.frame_cost(p0, p1, b, 0, 0, (0, 0), None, with_intra, False)
            lc, rows, rows_i, oo = o.cell(cfg, planes[b], None, None, 128, None, None, None, None, None, intra[b], inv[b], with_intra)
            glc, grows = ctx.lowres_costs(b, 0, 0)
            assert np.array_equal(glc, lc)
            if with_intra:  # the intra sums are only defined when the caller asked for them
                assert (out.intra_cost_est, out.intra_cost_est_aq) == (oo.intra_cost_est, oo.intra_cost_est_aq)
                assert np.array_equal(grows, rows_i)
            intra_done.add(b)
            continue
        do0 = (b, 0, d0 - 1) not in fields
        do1 = d1 > 0 and (b, 1, d1 - 1) not in fields
        wt = None
        if do0 and d1 == 0 and weights and (b, p0) in weights:
            wt = weights[(b, p0)]
        if do0:
            w = Weight(*wt) if wt else None
            wplane = o.weight_plane(cfg, planes[p0][0], w) if wt else None
            fields[(b, 0, d0 - 1)] = o.search_field(cfg, planes[b], planes[p0], w, wplane)
        if do1:
            fields[(b, 1, d1 - 1)] = o.search_field(cfg, planes[b], planes[p1])
        ref1_valid = d1 > 0 and (p1, 0, d0 + d1 - 1) in fields
        out = ctx.frame_cost(p0, p1, b, d0, d1, (do0, do1), wt, with_intra, ref1_valid)
        m0, c0 = fields[(b, 0, d0 - 1)]
        gm, gc = ctx.mvs(b, 0, d0 - 1)
        assert np.array_equal(gm, m0), ("L0 mvs", p0, p1, b, int((gm != m0).any(1).sum()))
        assert np.array_equal(gc, c0), ("L0 costs", p0, p1, b)
        dsf = (d0 * 256 + (d0 + d1) // 2) // (d0 + d1)
        if d1 > 0:
            m1, c1 = fields[(b, 1, d1 - 1)]
            gm1, gc1 = ctx.mvs(b, 1, d1 - 1)
            assert np.array_equal(gm1, m1) and np.array_equal(gc1, c1), ("L1", p0, p1, b)
            r1 = fields[(p1, 0, d0 + d1 - 1)][0] if ref1_valid else None
            lc, rows, rows_i, oo = o.cell(cfg, planes[b], planes[p0], planes[p1], dsf, m0, c0, m1, c1, r1, intra[b], inv[b], with_intra)
        else:
            lc, rows, rows_i, oo = o.cell(cfg, planes[b], planes[p0], None, dsf, m0, c0, None, None, None, intra[b], inv[b], with_intra)
            intra_done.add(b)
        glc, grows = ctx.lowres_costs(b, d0, d1)
        assert np.array_equal(glc, lc), ("lowres_costs", p0, p1, b, int((glc != lc).sum()))
        assert np.array_equal(grows, rows), ("row_satds", p0, p1, b)
        assert (out.cost_est, out.cost_est_aq) == (oo.cost_est, oo.cost_est_aq), ("sums", p0, p1, b)
        if d1 == 0:
            assert out.intra_mbs == oo.intra_mbs
        if with_intra:
            assert (out.intra_cost_est, out.intra_cost_est_aq) == (oo.intra_cost_est, oo.intra_cost_est_aq)
    return len(fields)


@pytest.mark.parametrize("cfgname", list(CONFIGS))
@pytest.mark.parametrize("clipname", ["pan", "fastpan", "noise", "static"])
def test_eval_sequence(cfgname, clipname):
    depth = CONFIGS[cfgname][0]
    W, H, nf = (352, 288, 4) if clipname == "pan" else (176, 144, 4)
    frames = clip(clipname, W, H, nf, depth)
    o, cfg, ctx = _mk(*CONFIGS[cfgname], W, H)
    try:
        assert _run_sequence(o, cfg, ctx, frames) >= 4
    finally:
        ctx.close()


@pytest.mark.parametrize("cfgname", ["hex_r4", "hex_10bit"])
def test_weighted_search(cfgname):
    depth = CONFIGS[cfgname][0]
    frames = clip("fade", 176, 144, 4, depth)
    o, cfg, ctx = _mk(*CONFIGS[cfgname], 176, 144)
    try:
        weights = {(1, 0): (1, 55, 6, 3), (2, 0): (1, 100, 7, -4), (3, 2): (1, 3, 0, -2)}
        _run_sequence(o, cfg, ctx, frames, weights)
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_mc_luma_all_phases_and_edges(depth):
    """x264_mc_functions_t.mc_luma / get_ref on the lowres planes (common/mc.c:198-249), swept the way tools/checkasm.c:1226-1290 sweeps
    the reference's: all 16 quarter-pel phases x full-pel displacements up to the furthest the 32-sample padding allows, for blocks in
    the picture's corners, on its edges and inside, unweighted and under four weights (denominator 0, rounding, negative offset, clip at
    both ends) -- the device's tap arithmetic over the strip copy (what every candidate of the search and cell kernels reads) against
    the oracle's mc_luma over the oracle's own planes."""
    import ctypes as C
    W, H = 208, 112  # lowres 104 x 56 -> 13 x 7 blocks: the strip copy's last strips and the bottom rows are reached
    frames = clip("fastpan", W, H, 2, depth)
    cfgname = "hex_10bit" if depth == 10 else "hex_r4"
    o, cfg, ctx = _mk(*CONFIGS[cfgname], W, H)
    try:
        ctx.frame_put(0, frames[1])
        pl = o.lowres_init(cfg, frames[1])
        W8, H8 = 8 * cfg.mb_w, 8 * cfg.mb_h
        reqs = []
        for (x, y) in ((0, 0), (W8 - 8, 0), (0, H8 - 8), (W8 - 8, H8 - 8), (W8 // 2 - 3, H8 // 2 + 1), (8, H8 - 8), (W8 - 8, 16), (5, 3)):
            # full-pel displacements: none, one either way, the reference's own limit (12 samples outside, slicetype.c:585-588) and the
            # last position inside the padding (the quarter-pel partner reads one sample / one row further)
            xs = sorted({0, -1, 1, -x - 12, W8 - 8 - x + 12, -x - PAD, W8 + PAD - 9 - x})
            ys = sorted({0, -1, 1, -y - 12, H8 - 8 - y + 12, -y - PAD, H8 + PAD - 9 - y})
            for fy in ys:
                for fx in xs:
                    for ph in range(16):
                        reqs.append((x, y, 4 * fx + (ph & 3), 4 * fy + (ph >> 2)))
        reqs = np.array(reqs, np.int32)
        pp = o._plane_ptrs(pl)
        mc = o.f("mc_luma")
        for wt in (None, (1, 55, 6, 3), (1, 100, 5, -20), (1, 3, 0, -1), (1, 127, 7, 127)):
            got = ctx.mc_luma_probe(0, reqs, wt)
            w = Weight(*wt) if wt else None
            want = np.zeros_like(got)
            blk = np.zeros((8, 8), o.dtype)
            for i, (x, y, mvx, mvy) in enumerate(reqs):
                org = (C.c_void_p * 4)(*[pp[k] + (int(y) * cfg.stride + int(x)) * o.isz for k in range(4)])
                mc(C.c_void_p(blk.ctypes.data), 8, org, cfg.stride, int(mvx), int(mvy), 8, 8, C.byref(w) if w else None)
                want[i] = blk
            bad = np.nonzero((got != want).reshape(len(reqs), -1).any(axis=1))[0]
            assert not len(bad), (wt, len(bad), reqs[bad[:4]].tolist())
        # a block that leaves the padded planes is refused
        with pytest.raises(Exception):
            ctx.mc_luma_probe(0, np.array([[0, 0, 4 * (-PAD - 1), 0]], np.int32))
    finally:
        ctx.close()


def test_prefetch_does_not_change_results():
    frames = clip("fastpan", 176, 144, 4)
    o, cfg, ctx = _mk(*CONFIGS["hex_r4"], 176, 144)
    try:
        _run_sequence(o, cfg, ctx, frames, prefetch=True)
        assert ctx.counters()[2] > 0  # speculative fields were actually used
    finally:
        ctx.close()


def test_non_mod16_and_tiny():
    for (W, H) in ((200, 120), (32, 32), (48, 16)):
        frames = clip("fastpan", W, H, 3)
        o, cfg, ctx = _mk(*CONFIGS["hex_r4"], W, H)
        try:
            _run_sequence(o, cfg, ctx, frames)
        finally:
            ctx.close()


def test_weight_cost():
    frames = clip("fade", 176, 144, 3)
    o, cfg, ctx = _mk(*CONFIGS["hex_r4"], 176, 144)
    try:
        pls = []
        for i in range(3):
            ctx.frame_put(i, frames[i])
            pls.append(o.lowres_init(cfg, frames[i]))
        ic = o.intra_costs(cfg, pls[1])
        for wt in (None, (1, 55, 6, 3), (1, 127, 7, -128), (1, 2, 0, 5)):
            w = Weight(*wt) if wt else None
            assert ctx.weight_cost(1, 0, wt) == o.weight_cost(cfg, pls[1], pls[0], w, ic)
    finally:
        ctx.close()


def test_1080p_search_matches_oracle():
    """Full BASELINE config-1 geometry (1920x1080, dia): one P and one B evaluation."""
    W, H = 1920, 1080
    frames = clip("fastpan", W, H, 3)
    o, cfg, ctx = _mk(8, 0, 4, 16, 8, 1, 0, 3, W, H, mv_range=512)
    try:
        SEQ2 = [(0, 2, 2), (0, 2, 1)]
        global SEQ
        old, SEQ = SEQ, SEQ2
        try:
            _run_sequence(o, cfg, ctx, frames)
        finally:
            SEQ = old
    finally:
        ctx.close()


def test_mbtree_steps_match_oracle():
    """x264hip_mbtree on an explicit step list (zero / propagate P and B / finish) against the oracle's
    or_mbtree_propagate / or_mbtree_finish on the same fields and maps."""
    import ctypes as C
    frames = clip("fastpan", 176, 144, 3)
    o, cfg, ctx = _mk(*CONFIGS["hex_r4"], 176, 144)
    try:
        n = cfg.mb_w * cfg.mb_h
        planes, inv, intra, qp_aq = [], [], [], []
        for i in range(3):
            ctx.frame_put(i, frames[i])
            planes.append(o.lowres_init(cfg, frames[i]))
            iq, qp, _, _ = o.aq_frame(frames[i], cfg.mb_w, cfg.mb_h, 1, 1.0)
            inv.append(iq); qp_aq.append(qp); intra.append(o.intra_costs(cfg, planes[i]))
        # evaluations: P (0,2,2) then B (0,2,1)
        ctx.frame_cost(0, 2, 2, 2, 0, (1, 0), None, True, False)
        ctx.frame_cost(0, 2, 1, 1, 1, (1, 1), None, True, True)
        f20 = o.search_field(cfg, planes[2], planes[0])
        f10 = o.search_field(cfg, planes[1], planes[0]); f11 = o.search_field(cfg, planes[1], planes[2])
        lcP = o.cell(cfg, planes[2], planes[0], None, 128, f20[0], f20[1], None, None, None, intra[2], inv[2], True)[0]
        lcB = o.cell(cfg, planes[1], planes[0], planes[2], 128, f10[0], f10[1], f11[0], f11[1], f20[0], intra[1], inv[1], True)[0]
        fps = np.float32(0.04 / (0.04 * 256.0) * 0.5)
        ops = [lib.MbtreeOp(0, 2, 2, 2, 0, 0, 0, 0, 0.0, 0, 0.0, 0.0), lib.MbtreeOp(0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0, 0.0, 0.0),
               lib.MbtreeOp(1, 1, 0, 2, 1, 1, 0, 32, fps, 0, 0.0, 0.0),      # B frame 1, not referenced
               lib.MbtreeOp(1, 2, 0, 2, 2, 0, 1, 32, fps, 0, 0.0, 0.0),      # P frame 2, referenced
               lib.MbtreeOp(2, 0, 0, 0, 0, 0, 0, 0, 0.0, 512, 0.0, 2.0)]     # finish frame 0
        ctx.mbtree(ops)
        prop = [np.zeros(n, np.uint16) for _ in range(3)]
        L = o.lib
        L.or_mbtree_propagate.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_float]
        L.or_mbtree_propagate(cfg.mb_w, cfg.mb_h, intra[1].ctypes.data, lcB.ctypes.data, inv[1].ctypes.data, None,
                              f10[0].ctypes.data, f11[0].ctypes.data, prop[0].ctypes.data, prop[2].ctypes.data, 32, fps)
        pin = prop[2].copy()
        L.or_mbtree_propagate(cfg.mb_w, cfg.mb_h, intra[2].ctypes.data, lcP.ctypes.data, inv[2].ctypes.data, pin.ctypes.data,
                              f20[0].ctypes.data, None, prop[0].ctypes.data, None, 32, fps)
        qp = qp_aq[0].copy()
        L.or_mbtree_finish.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_float, C.c_float]
        L.or_mbtree_finish(n, intra[0].ctypes.data, inv[0].ctypes.data, prop[0].ctypes.data, qp_aq[0].ctypes.data, qp.ctypes.data, 512, 0.0, 2.0)
        assert prop[0].max() > 0
        for i in (0, 2):
            assert np.array_equal(ctx.propagate_cost(i), prop[i]), ("i_propagate_cost", i)
        assert np.array_equal(ctx.qp_offsets(0), qp), ("f_qp_offset", float(np.abs(ctx.qp_offsets(0) - qp).max()))
    finally:
        ctx.close()


def test_mbtree_queued_and_immediate_lists_interleave():
    """x264hip_mbtree queues step lists that clear every accumulator they read and runs them side by side on accumulator banks of
    their own; lists that continue from what an earlier list left run at once on the frames' own accumulators.  A caller must not be
    able to tell: two queued lists (the second lands in bank 1), then a list that adds to the accumulators without clearing them,
    every accumulator and offset map against the oracle applied in the caller's order."""
    import ctypes as C
    frames = clip("fastpan", 176, 144, 3)
    o, cfg, ctx = _mk(*CONFIGS["hex_r4"], 176, 144)
    try:
        n = cfg.mb_w * cfg.mb_h
        planes, inv, intra, qp_aq = [], [], [], []
        for i in range(3):
            ctx.frame_put(i, frames[i])
            planes.append(o.lowres_init(cfg, frames[i]))
            iq, qp, _, _ = o.aq_frame(frames[i], cfg.mb_w, cfg.mb_h, 1, 1.0)
            inv.append(iq); qp_aq.append(qp); intra.append(o.intra_costs(cfg, planes[i]))
        ctx.frame_cost(0, 2, 2, 2, 0, (1, 0), None, True, False)
        ctx.frame_cost(0, 2, 1, 1, 1, (1, 1), None, True, True)
        f20 = o.search_field(cfg, planes[2], planes[0])
        f10 = o.search_field(cfg, planes[1], planes[0]); f11 = o.search_field(cfg, planes[1], planes[2])
        lcP = o.cell(cfg, planes[2], planes[0], None, 128, f20[0], f20[1], None, None, None, intra[2], inv[2], True)[0]
        lcB = o.cell(cfg, planes[1], planes[0], planes[2], 128, f10[0], f10[1], f11[0], f11[1], f20[0], intra[1], inv[1], True)[0]
        L = o.lib
        L.or_mbtree_propagate.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_float]
        L.or_mbtree_finish.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_float, C.c_float]
        prop = [np.zeros(n, np.uint16) for _ in range(3)]
        qp_now = [q.copy() for q in qp_aq]

        def oracle_b(fps):
            L.or_mbtree_propagate(cfg.mb_w, cfg.mb_h, intra[1].ctypes.data, lcB.ctypes.data, inv[1].ctypes.data, None,
                                  f10[0].ctypes.data, f11[0].ctypes.data, prop[0].ctypes.data, prop[2].ctypes.data, 32, fps)

        def oracle_p(fps):
            pin = prop[2].copy()
            L.or_mbtree_propagate(cfg.mb_w, cfg.mb_h, intra[2].ctypes.data, lcP.ctypes.data, inv[2].ctypes.data, pin.ctypes.data,
                                  f20[0].ctypes.data, None, prop[0].ctypes.data, None, 32, fps)

        def oracle_finish(i, strength):
            L.or_mbtree_finish(n, intra[i].ctypes.data, inv[i].ctypes.data, prop[i].ctypes.data, qp_aq[i].ctypes.data, qp_now[i].ctypes.data, 512, 0.0, strength)

        def zero(i): return lib.MbtreeOp(0, i, i, i, 0, 0, 0, 0, 0.0, 0, 0.0, 0.0)
        def prop_b(fps): return lib.MbtreeOp(1, 1, 0, 2, 1, 1, 0, 32, fps, 0, 0.0, 0.0)
        def prop_p(fps): return lib.MbtreeOp(1, 2, 0, 2, 2, 0, 1, 32, fps, 0, 0.0, 0.0)
        def finish(i, strength): return lib.MbtreeOp(2, i, i, i, 0, 0, 0, 0, 0.0, 512, 0.0, strength)
        fps1 = np.float32(0.04 / (0.04 * 256.0) * 0.5); fps2 = np.float32(fps1 * 0.5)
        # list 1 (queued, bank 0): finishes frame 0
        ctx.mbtree([zero(2), zero(0), prop_b(fps1), prop_p(fps1), finish(0, 2.0)])
        prop[0][:] = 0; prop[2][:] = 0; oracle_b(fps1); oracle_p(fps1); oracle_finish(0, 2.0)
        # list 2 (queued beside it, bank 1): other factors, finishes frame 2
        ctx.mbtree([zero(2), zero(0), prop_b(fps2), prop_p(fps2), finish(2, 1.5)])
        prop[0][:] = 0; prop[2][:] = 0; oracle_b(fps2); oracle_p(fps2); oracle_finish(2, 1.5)
        # list 3 continues from there (clears nothing): runs at once, behind the queue, on what list 2 left
        ctx.mbtree([prop_p(fps1), finish(0, 1.0)])
        oracle_p(fps1); oracle_finish(0, 1.0)
        for i in (0, 2):
            assert np.array_equal(ctx.propagate_cost(i), prop[i]), ("i_propagate_cost", i)
            assert np.array_equal(ctx.qp_offsets(i), qp_now[i]), ("f_qp_offset", i)
        # and a queued list after the immediate one: the accumulators it clears are the frames' own again
        ctx.mbtree([zero(2), zero(0), prop_b(fps2), prop_p(fps1), finish(0, 2.0)])
        prop[0][:] = 0; prop[2][:] = 0; oracle_b(fps2); oracle_p(fps1); oracle_finish(0, 2.0)
        for i in (0, 2):
            assert np.array_equal(ctx.propagate_cost(i), prop[i]), ("i_propagate_cost after the last list", i)
        assert np.array_equal(ctx.qp_offsets(0), qp_now[0])
    finally:
        ctx.close()


def test_frame_cost_recalculate():
    """x264hip_frame_cost_recalculate (slicetype_frame_cost_recalculate, slicetype.c:999-1024) against the oracle on the
    device's own maps: a P cell under f_qp_offset, a B cell under f_qp_offset_aq, and the I cell after an MB-tree finish."""
    import ctypes as C
    frames = clip("fastpan", 176, 144, 3)
    o, cfg, ctx = _mk(*CONFIGS["hex_r4"], 176, 144)
    try:
        qp_aq = []
        for i in range(3):
            ctx.frame_put(i, frames[i])
            qp_aq.append(o.aq_frame(frames[i], cfg.mb_w, cfg.mb_h, 1, 1.0)[1])
        ctx.frame_cost(0, 0, 0, 0, 0, (0, 0), None, True, False)
        ctx.frame_cost(0, 2, 2, 2, 0, (1, 0), None, True, False)
        ctx.frame_cost(0, 2, 1, 1, 1, (1, 1), None, True, True)
        fps = np.float32(0.04 / (0.04 * 256.0) * 0.5)
        ctx.mbtree([lib.MbtreeOp(0, 2, 2, 2, 0, 0, 0, 0, 0.0, 0, 0.0, 0.0), lib.MbtreeOp(0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0, 0.0, 0.0),
                    lib.MbtreeOp(1, 2, 0, 2, 2, 0, 1, 32, fps, 0, 0.0, 0.0), lib.MbtreeOp(2, 0, 0, 0, 0, 0, 0, 0, 0.0, 512, 0.0, 2.0)])
        f = o.f("frame_cost_recalculate", C.c_int)
        f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        for slot, d0, d1, use_aq in ((2, 2, 0, False), (1, 1, 1, True), (0, 0, 0, False)):
            lc, _ = ctx.lowres_costs(slot, d0, d1)
            qp = np.ascontiguousarray(qp_aq[slot] if use_aq else ctx.qp_offsets(slot), np.float32)
            rows = np.zeros(cfg.mb_h, np.int32)
            want = f(cfg.mb_w, cfg.mb_h, lc.ctypes.data, qp.ctypes.data, rows.ctypes.data)
            got = ctx.frame_cost_recalculate(slot, d0, d1, use_aq)
            _, rows_dev = ctx.lowres_costs(slot, d0, d1)
            assert got == want and np.array_equal(rows_dev, rows), (slot, d0, d1, use_aq, got, want)
        assert not np.array_equal(ctx.qp_offsets(0), qp_aq[0])  # the finish step really changed frame 0's offsets
    finally:
        ctx.close()


from x264_amd.synth import upscaled_clip  # noqa: E402,F401  (imported from here by other test modules)


def test_8k_10bit_search_matches_oracle():
    """BASELINE configs[4] geometry (7680x4320 10-bit, veryslow + tesa: HEX range 24, fpelcmp = SATD, bframes 8, level 6.x mv range):
    lowres planes, AQ, intra costs, one P and one B evaluation, every array against the oracle.  8.7 M samples per padded lowres
    plane: the size where 24-bit offset arithmetic would first go wrong."""
    W, H = 7680, 4320
    frames = upscaled_clip(W, H, 3, 10, seed=5, pan=(17, -9), noise=9, texture=0.35)
    o, cfg, ctx = _mk(10, 1, 4, 24, 10, 1, 1, 8, W, H, mv_range=8192)
    try:
        global SEQ
        old, SEQ = SEQ, [(0, 2, 2), (0, 2, 1)]
        try:
            assert _run_sequence(o, cfg, ctx, frames) == 3
        finally:
            SEQ = old
    finally:
        ctx.close()


def test_bright_10bit_4k_frame_stats():
    """Luma total above 2^32: i_pixel_sum wraps like the reference's uint32_t before the mean is removed from the ssd
    (common/frame.h:140, ratecontrol.c:405-414; pinned against the reference in tests/test_oracle_vs_ref.py)."""
    W, H = 3840, 2160
    rng = np.random.default_rng(3)
    y = (900 + rng.integers(0, 100, (H, W))).astype(np.uint16)
    o = Oracle(10)
    ctx = lib.Context(W, H, bit_depth=10, max_frames=2)
    try:
        ctx.frame_put(0, y)
        got = ctx.frame_stats(0)
        inv = ctx.inv_qscale(0)
    finally:
        ctx.close()
    iq, _, s, ssd = o.aq_frame(y, (W + 15) // 16, (H + 15) // 16, 1, 1.0)
    assert int(y.astype(np.uint64).sum()) > 1 << 32
    assert got == (s, ssd)
    assert np.array_equal(inv, iq)
