"""Pins the CPU restatement (oracle/) against the REAL reference C path (oracle/_ref, built from
/root/reference by oracle/build_ref.sh).  Skipped where the reference build is not present; the
golden-fixture tests (test_golden.py) cover those boxes."""
import numpy as np
import pytest

from oracle import refharness
from oracle.oraclelib import Oracle, PAD, Weight
from tests.common import clip, oracle_cfg

pytestmark = pytest.mark.skipif(not refharness.available(8), reason="oracle/_ref not built (no /root/reference)")

CONFIGS = [
    # (preset, opts, depth)
    ("medium", "", 8),
    ("slow", "me=dia", 8),
    ("slower", "me=umh,merange=32", 8),
    ("medium", "subme=1", 8),
    ("medium", "subme=2,me=dia", 8),
    ("veryslow", "me=tesa", 10),
    ("medium", "bframes=8,rc-lookahead=60", 10),
]


def _eval_sequence(r, o, cfg, planes, inv, nf):
    """Run the same (p0,p1,b) sequence in the reference and through the oracle's pure functions."""
    fields = {}   # (b, list, dist) -> (mvs, costs, weight)
    intra = {}

    def get_intra(b):
        if b not in intra:
            intra[b] = o.intra_costs(cfg, planes[b])
        return intra[b]

    def search(b, lst, ref, wt=None):
        key = (b, lst, abs(ref - b) - 1)
        if key not in fields:
            wplane = None
            if wt is not None and wt.on:
                wplane = o.weight_plane(cfg, planes[ref][0], wt)
            fields[key] = o.search_field(cfg, planes[b], planes[ref], wt, wplane)
        return fields[key]

    seq = [(0, 0, 0), (0, 1, 1), (0, 2, 2), (0, 2, 1), (1, 1, 1), (0, 3, 3), (0, 3, 1), (0, 3, 2), (1, 3, 2), (2, 3, 3),
           (3, 3, 3)]
    for (p0, p1, b) in seq:
        if max(p0, p1, b) >= nf:
            continue
        score = r.frame_cost(p0, p1, b)
        ic = get_intra(b)
        if p0 == p1:
            lc, rows, rows_i, out = o.cell(cfg, planes[b], None, None, 128, None, None, None, None, None, ic, inv[b], True)
            lc_ref, _, summ = r.cell(b, 0, 0)
            assert np.array_equal(lc, lc_ref)
            assert out.intra_cost_est == summ[0] == score and out.intra_cost_est_aq == summ[1]
            continue
        wt = None
        if b == p1:
            w = r.weight(b)
            # weights only apply when this call triggered the search (slicetype.c:857-866)
            if (b, 0, b - p0 - 1) not in fields and w[0]:
                wt = Weight(*w)
        m0, c0 = search(b, 0, p0, wt)
        mr, cr = r.mvs(b, 0, b - p0 - 1)
        assert np.array_equal(m0, mr), ("L0 mvs", p0, p1, b, int((m0 != mr).any(1).sum()))
        assert np.array_equal(c0, cr), ("L0 costs", p0, p1, b)
        dsf = ((b - p0) * 256 + (p1 - p0) // 2) // (p1 - p0)
        if b < p1:
            m1, c1 = search(b, 1, p1)
            mr1, cr1 = r.mvs(b, 1, p1 - b - 1)
            assert np.array_equal(m1, mr1) and np.array_equal(c1, cr1), ("L1", p0, p1, b)
            ref1_l0 = fields.get((p1, 0, p1 - p0 - 1), (None,))[0]
            lc, rows, _, out = o.cell(cfg, planes[b], planes[p0], planes[p1], dsf, m0, c0, m1, c1, ref1_l0, ic, inv[b], False)
            expect = out.cost_est * 100 // (120 + cfg.bframe_bias)
        else:
            lc, rows, _, out = o.cell(cfg, planes[b], planes[p0], None, dsf, m0, c0, None, None, None, ic, inv[b], False)
            expect = out.cost_est
        lc_ref, rows_ref, summ = r.cell(b, b - p0, p1 - b)
        assert np.array_equal(lc, lc_ref), ("lowres_costs", p0, p1, b)
        assert expect == summ[0] == score, ("cost_est", p0, p1, b)
        assert out.cost_est_aq == summ[1], ("cost_est_aq", p0, p1, b)
        if b == p1:
            assert out.intra_mbs == summ[2]
    return len(fields)


@pytest.mark.parametrize("preset,opts,depth", CONFIGS)
@pytest.mark.parametrize("clipname", ["pan", "fastpan", "noise", "static", "fade"])
def test_eval_sequence(preset, opts, depth, clipname):
    W, H, nf = (176, 144, 4) if clipname != "pan" else (352, 288, 4)
    frames = clip(clipname, W, H, nf, depth)
    r = refharness.Ref(W, H, preset, opts=opts, bit_depth=depth)
    try:
        o = Oracle(depth)
        cfg = oracle_cfg(o, r.cfg)
        assert np.array_equal(r.cost_mv(), o._cost_mv)
        planes, inv = [], []
        g = None
        for i in range(nf):
            r.add_frame(frames[i])
            pl = o.lowres_init(cfg, frames[i])
            g = g or r.lowres_geometry()
            for p in range(4):
                assert np.array_equal(r.lowres(i, p), pl[p][:, :g["width"] + 2 * PAD]), ("lowres", i, p)
            planes.append(pl)
            iq, _, ss = r.frame_stats(i)
            iq_o, _, s, ssd = o.aq_frame(frames[i], cfg.mb_w, cfg.mb_h, r.cfg["aq_mode"], r.cfg["aq_strength_q16"] / 65536.0)
            assert np.array_equal(iq, iq_o) and (s, ssd) == ss
            inv.append(iq)
        n = _eval_sequence(r, o, cfg, planes, inv, nf)
        assert n >= 4
    finally:
        r.close()


def test_non_mod16_size():
    W, H = 200, 120  # lowres runs over the mod16 size 208x128 with replicated edges
    frames = clip("fastpan", W, H, 3)
    r = refharness.Ref(W, H, "medium")
    try:
        o = Oracle(8)
        cfg = oracle_cfg(o, r.cfg)
        planes, inv = [], []
        for i in range(3):
            r.add_frame(frames[i])
            planes.append(o.lowres_init(cfg, frames[i]))
            inv.append(r.frame_stats(i)[0])
            g = r.lowres_geometry()
            for p in range(4):
                assert np.array_equal(r.lowres(i, p), planes[i][p][:, :g["width"] + 2 * PAD])
        _eval_sequence(r, o, cfg, planes, inv, 3)
    finally:
        r.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_frame_cost_recalculate(depth):
    """slicetype_frame_cost_recalculate (slicetype.c:999-1024): oracle vs the reference's static function on evaluated cells,
    with the AQ offsets (B-frame form) and with arbitrary MB-tree-like offsets incl. values that saturate exp2fix8."""
    import ctypes as C
    W, H = 176, 144
    frames = clip("pan", W, H, 4, depth)
    r = refharness.Ref(W, H, "medium", bit_depth=depth)
    o = Oracle(depth)
    try:
        for f in frames:
            r.add_frame(f)
        L = r.lib
        L.rh_frame_cost_recalculate.argtypes = [C.c_void_p] + [C.c_int] * 4
        L.rh_get_qp_offsets.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rh_set_qp_offsets.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        f_or = o.f("frame_cost_recalculate", C.c_int)
        f_or.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        rng = np.random.default_rng(3)
        for (p0, p1, b) in [(0, 1, 1), (0, 2, 1), (0, 3, 3), (1, 3, 2)]:
            r.frame_cost(p0, p1, b)
            lc, _, _ = r.cell(b, b - p0, p1 - b)
            qp = np.zeros(r.n_mb, np.float32)
            qp_aq = np.zeros(r.n_mb, np.float32)
            for variant in range(3):
                if variant == 2:
                    mt = rng.uniform(-60, 60, size=r.n_mb).astype(np.float32)   # beyond both exp2fix8 clamps
                    mt[::7] = 0
                    L.rh_set_qp_offsets(r.ctx, b, mt.ctypes.data)
                L.rh_get_qp_offsets(r.ctx, b, qp.ctypes.data, qp_aq.ctypes.data)
                is_b = variant == 0
                want = L.rh_frame_cost_recalculate(r.ctx, p0, p1, b, int(is_b))
                _, rows_ref, _ = r.cell(b, b - p0, p1 - b)
                rows = np.zeros(r.mb_h, np.int32)
                src = qp_aq if is_b else qp
                got = f_or(r.mb_w, r.mb_h, lc.ctypes.data, src.ctypes.data, rows.ctypes.data)
                assert got == want and np.array_equal(rows, rows_ref), (p0, p1, b, variant)
    finally:
        r.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_adaptive_quant_with_chroma(depth):
    """x264_adaptive_quant_frame on whole 4:2:0 pictures (luma + Cb + Cr energy, ratecontrol.c:258-276; aq-mode 1, 2, 3; non-mod16
    sizes): i_inv_qscale_factor and the luma sum / ssd the weight analysis reads."""
    if not refharness.available(depth):
        pytest.skip("no reference build for this depth")
    from x264_amd.synth import make_chroma, make_clip
    o = Oracle(depth)
    for (W, H) in ((176, 144), (100, 70)):
        y = make_clip(W, H, 1, seed=5, bit_depth=depth, noise=20)[0]
        cb, cr = (c[0] for c in make_chroma(W, H, 1, seed=5, bit_depth=depth))
        for mode, strength in ((1, 1.0), (2, 1.0), (3, 0.6)):
            r = refharness.Ref(W, H, "medium", opts="aq-mode=%d,aq-strength=%g" % (mode, strength), bit_depth=depth)
            try:
                r.add_frame(y, cb, cr)
                iq, _, ss = r.frame_stats(0)
            finally:
                r.close()
            iq_o, _, s, ssd = o.aq_frame(y, (W + 15) // 16, (H + 15) // 16, mode, strength, cb, cr)
            assert np.array_equal(iq, iq_o), (W, H, mode)
            assert (s, ssd) == ss


def test_pixel_sum_wraps_like_the_reference():
    """i_pixel_sum is a uint32_t (common/frame.h:140): on a bright 10-bit 4K picture the luma total exceeds 2^32 and the
    reference squares the WRAPPED value when it removes the mean from i_pixel_ssd (ratecontrol.c:405-414).  The oracle (and
    x264hip_frame_stats, which the GPU suite compares with the oracle) must reproduce exactly that."""
    if not refharness.available(10):
        pytest.skip("no reference build for this depth")
    W, H = 3840, 2160
    rng = np.random.default_rng(3)
    y = (900 + rng.integers(0, 100, (H, W))).astype(np.uint16)
    assert int(y.astype(np.uint64).sum()) > 1 << 32
    r = refharness.Ref(W, H, "medium", bit_depth=10)
    try:
        r.add_frame(y)
        iq, _, ss = r.frame_stats(0)
    finally:
        r.close()
    o = Oracle(10)
    iq_o, _, s, ssd = o.aq_frame(y, (W + 15) // 16, (H + 15) // 16, 1, 1.0)
    assert np.array_equal(iq, iq_o)
    assert (s, ssd) == ss
