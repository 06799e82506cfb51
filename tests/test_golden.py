"""CPU tests against the committed golden fixtures (tests/golden/, generated from the real reference by
tests/golden/make_golden.py): the oracle restatement and the product's host lookahead logic must
reproduce the reference's outputs without /root/reference being present."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from oracle.oraclelib import Oracle, PAD, Weight
from tests.common import clip
from tests.golden.make_golden import EVAL_SEQ, LOOKAHEAD_CASES, LOOKAHEAD_CASES_R2
from tests.oracle_backend import OracleBackend
from x264_amd import lib
from x264_amd.synth import make_chroma, make_clip

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cfgdict(z):
    return {str(k): int(v) for k, v in zip(z["cfg_keys"], z["cfg"])}


@pytest.mark.parametrize("mv_range,depth", [(128, 8), (512, 8), (128, 10), (512, 10)])
def test_cost_mv_tables(mv_range, depth):
    gold = np.load(os.path.join(GOLD, "cost_mv_r%d_d%d.npy" % (mv_range, depth)))
    lam = 1 if depth == 8 else 4
    tab, centre = lib.cost_mv_table(mv_range, lam)
    assert np.array_equal(tab, gold)
    o = Oracle(depth)
    o.make_cfg(4, 4, me_method=1, subpel_refine=4, me_range=16, mv_range=mv_range, subme=7, mbcmp_satd=1)
    assert np.array_equal(o._cost_mv, gold)


@pytest.mark.parametrize("depth", [8, 10])
def test_primitive_known_answers(depth):
    z = np.load(os.path.join(GOLD, "primitives_d%d.npz" % depth))
    o = Oracle(depth)
    a, b = np.ascontiguousarray(z["a"]), np.ascontiguousarray(z["b"])
    sizes = [(16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)]
    for oi, (ox, oy) in enumerate(z["offs"]):
        pb = C.c_void_p(b.ctypes.data + (int(oy) * 64 + int(ox)) * b.itemsize)
        pa = C.c_void_p(a.ctypes.data)
        for si, (w, h) in enumerate(sizes):
            assert o.f("sad", C.c_int)(pa, 64, pb, 64, w, h) == z["cmp"][0, si, oi]
            assert o.f("satd", C.c_int)(pa, 64, pb, 64, w, h) == z["cmp"][1, si, oi]
            assert o.f("ssd", C.c_int)(pa, 64, pb, 64, w, h) == z["cmp"][2, si, oi]
        assert o.f("sa8d", C.c_int)(pa, 64, pb, 64, 16) == z["cmp"][3, 0, oi]
        assert o.f("sa8d", C.c_int)(pa, 64, pb, 64, 8) == z["cmp"][3, 3, oi]
    fenc, fdec = np.ascontiguousarray(z["fenc"]), np.ascontiguousarray(z["fdec"])
    for kind, n in {0: 16, 1: 64, 2: 256, 3: 64, 4: 256, 5: 4, 6: 8}.items():
        out = np.zeros(n, o.coef_dtype)
        o.f("dct")(kind, C.c_void_p(out.ctypes.data), C.c_void_p(fenc.ctypes.data), C.c_void_p(fdec.ctypes.data))
        assert np.array_equal(out, z["dct%d" % kind]), kind
    for kind, src, mf, bias, gold, nz in ((0, "dct0", "mf4", "bias4", "q4", 0), (1, "dct3", "mf8", "bias8", "q8", 1)):
        c = z[src].copy()
        mfa, ba = np.ascontiguousarray(z[mf]), np.ascontiguousarray(z[bias])
        r = o.f("quant", C.c_int)(kind, C.c_void_p(c.ctypes.data), C.c_void_p(mfa.ctypes.data), C.c_void_p(ba.ctypes.data), 0, 0)
        assert np.array_equal(c, z[gold]) and r == z["nz"][nz]
    # hpel_filter, all three planes incl. the extra dstv columns and the untouched surroundings
    hw, hh, hs = (int(v) for v in z["hpel_dims"])
    hsrc = np.ascontiguousarray(z["hpel_src"])
    out = np.full((3, hh + 8, hs), 7, o.dtype)
    buf = np.zeros(hw + 64, np.int16)
    off = (3 * hs + 8) * hsrc.itemsize
    f = o.f("hpel_filter")
    f.argtypes = [C.c_void_p] * 4 + [C.c_long, C.c_int, C.c_int, C.c_void_p]
    f(out[0].ctypes.data + off, out[1].ctypes.data + off, out[2].ctypes.data + off, hsrc.ctypes.data + off, hs, hw, hh, buf.ctypes.data)
    assert np.array_equal(out, z["hpel_out"])


EVALSEQ_FILES = sorted(glob.glob(os.path.join(GOLD, "evalseq_*.npz")))


@pytest.mark.parametrize("path", EVALSEQ_FILES, ids=[os.path.basename(p)[8:-4] for p in EVALSEQ_FILES])
def test_oracle_eval_sequence_vs_golden(path):
    z = np.load(path)
    rc = _cfgdict(z)
    depth = 10 if "_10_" in os.path.basename(path) else 8
    clipname = os.path.basename(path)[:-4].split("_")[-1]
    W, H, nf = 176, 144, 4
    frames = clip(clipname, W, H, nf, depth)
    o = Oracle(depth)
    cfg = o.make_cfg(rc["mb_w"], rc["mb_h"], me_method=rc["me_method"], subpel_refine=rc["subpel_refine"], me_range=rc["me_range"],
                     mv_range=rc["mv_range"], subme=rc["subme"], mbcmp_satd=rc["mbcmp_satd"], fpelcmp_satd=rc["fpelcmp_satd"],
                     weighted_bipred=rc["weighted_bipred"], aq_mode=rc["aq_mode"], lam=rc["lambda"], bframe_bias=rc["b_bias"])
    planes, inv = [], []
    for i in range(nf):
        pl = o.lowres_init(cfg, frames[i])
        assert np.array_equal(pl[0][:, :8 * cfg.mb_w + 2 * PAD], z["lowres0_%d" % i])
        crc = [int(pl[p][:, :8 * cfg.mb_w + 2 * PAD].astype(np.uint64).sum()) for p in range(4)]
        assert crc == [int(v) for v in z["lowres_crc_%d" % i]]
        iq, _, s, ssd = o.aq_frame(frames[i], cfg.mb_w, cfg.mb_h, rc["aq_mode"], rc["aq_strength_q16"] / 65536.0)
        assert np.array_equal(iq, z["inv_%d" % i]) and [s, ssd] == [int(v) for v in z["sums_%d" % i]]
        planes.append(pl); inv.append(iq)
        assert np.array_equal(o.intra_costs(cfg, pl), z["intra_%d" % i])
    fields = {}
    for k, (p0, p1, b) in enumerate(EVAL_SEQ):
        ic = o.intra_costs(cfg, planes[b])
        summ = [int(v) for v in z["summ_%d" % k]]
        if p0 == p1:
            lc, _, _, out = o.cell(cfg, planes[b], None, None, 128, None, None, None, None, None, ic, inv[b], True)
            assert np.array_equal(lc, z["lc_%d" % k]) and out.intra_cost_est == summ[0] == summ[3]
            continue
        wt = None
        if b == p1 and (b, 0, b - p0 - 1) not in fields:
            w = [int(v) for v in z["weight_%d" % k]]
            if w[0]:
                wt = Weight(*w)
        if (b, 0, b - p0 - 1) not in fields:
            wplane = o.weight_plane(cfg, planes[p0][0], wt) if wt else None
            fields[(b, 0, b - p0 - 1)] = o.search_field(cfg, planes[b], planes[p0], wt, wplane)
        m0, c0 = fields[(b, 0, b - p0 - 1)]
        assert np.array_equal(m0, z["mv0_%d" % k]) and np.array_equal(c0, z["c0_%d" % k]), ("L0", k)
        dsf = ((b - p0) * 256 + (p1 - p0) // 2) // (p1 - p0)
        if b < p1:
            if (b, 1, p1 - b - 1) not in fields:
                fields[(b, 1, p1 - b - 1)] = o.search_field(cfg, planes[b], planes[p1])
            m1, c1 = fields[(b, 1, p1 - b - 1)]
            assert np.array_equal(m1, z["mv1_%d" % k]) and np.array_equal(c1, z["c1_%d" % k]), ("L1", k)
            r1 = fields.get((p1, 0, p1 - p0 - 1), (None,))[0]
            lc, _, _, out = o.cell(cfg, planes[b], planes[p0], planes[p1], dsf, m0, c0, m1, c1, r1, ic, inv[b], False)
            assert out.cost_est * 100 // (120 + cfg.bframe_bias) == summ[0] == summ[3]
        else:
            lc, _, _, out = o.cell(cfg, planes[b], planes[p0], None, dsf, m0, c0, None, None, None, ic, inv[b], False)
            assert out.cost_est == summ[0] == summ[3] and out.intra_mbs == summ[2]
        assert np.array_equal(lc, z["lc_%d" % k]), ("lowres_costs", k)
        assert out.cost_est_aq == summ[1]


def check_lookahead_outputs(outs, z, nb, check_qp=True):
    assert [o.frame for o in outs] == [int(v) for v in z["idx"]], "coded order differs"
    assert [o.type for o in outs] == [int(v) for v in z["type"]], "slice types differ"
    if check_qp and "qp_offset" in z and hasattr(outs[0], "qp_offset"):
        # MB-tree / AQ output read by rate control: FP32, same operation order as the reference build -> bit-exact
        for k, o in enumerate(outs):
            assert np.array_equal(o.qp_offset, z["qp_offset"][k]), ("f_qp_offset", k, o.frame, o.type,
                                                                   float(np.abs(o.qp_offset - z["qp_offset"][k]).max()))
    if check_qp and "qp_crc" in z and hasattr(outs[0], "qp_offset"):
        # full-size fixtures hold a CRC-32 of every frame's f_qp_offset instead of the map
        import zlib
        for k, o in enumerate(outs):
            assert zlib.crc32(np.ascontiguousarray(o.qp_offset, np.float32).tobytes()) == int(z["qp_crc"][k]), ("f_qp_offset crc", k, o.frame, o.type)
    if "planned_type" in z and hasattr(outs[0], "planned"):
        # VBV outputs (slicetype.c:1224-1286, :1916-1934): plans of the non-B frames, row sums of the cell each frame is coded with
        for k, o in enumerate(outs):
            if o.type not in (4, 5):
                want = []
                for t, s in zip(z["planned_type"][k], z["planned_satd"][k]):
                    if t == 0:
                        break
                    want.append((int(t), int(s)))
                assert o.planned == want, ("i_planned_type/satd", k, o.frame)
            # x264_rc_analyse_slice of the reference on the leaving frame: the cell, its cost and the row sums it leaves behind
            mbh = len(o.row_satds)
            assert o.own_cell == tuple(int(v) for v in z["rc_cells"][k]), ("cell", k, o.frame)
            assert o.rc_satd == z["rc"][k][0], ("rc satd", k, o.frame)
            assert np.array_equal(o.row_satds, z["rc"][k][1:1 + mbh]), ("i_row_satd", k, o.frame)
            if o.type not in (1, 2):
                assert np.array_equal(o.row_satds_intra, z["rc"][k][1 + mbh:]), ("i_row_satds[0][0]", k, o.frame)
    for k, o in enumerate(outs):
        ce = np.array([[o.cost_est[i][j] for j in range(nb)] for i in range(nb)])
        ca = np.array([[o.cost_est_aq[i][j] for j in range(nb)] for i in range(nb)])
        assert np.array_equal(ce, z["cost"][k]), ("i_cost_est", k, o.frame)
        m = z["cost"][k] >= 0
        assert np.array_equal(ca[m], z["cost_aq"][k][m]), ("i_cost_est_aq", k, o.frame)


@pytest.mark.parametrize("name", list(LOOKAHEAD_CASES) + list(LOOKAHEAD_CASES_R2))
def test_host_lookahead_vs_golden(name):
    preset, opts, over, depth, W, H, ckw, nf = {**LOOKAHEAD_CASES, **LOOKAHEAD_CASES_R2}[name]
    z = np.load(os.path.join(GOLD, "lookahead_%s.npz" % name))
    frames = make_clip(W, H, nf, bit_depth=depth, **ckw)
    over = dict(over)
    chroma = make_chroma(W, H, nf, seed=ckw.get("seed", 1), bit_depth=depth) if over.pop("_chroma", 0) else None
    cfg = lib.la_config(W, H, preset, bit_depth=depth, **over)
    be = OracleBackend(cfg)
    la = lib.Lookahead(cfg, backend=be.struct)
    try:
        outs = la.run(frames, qp_offsets=True, vbv=bool(cfg["vbv"]), chroma=chroma)
    finally:
        la.close()
    check_lookahead_outputs(outs, z, cfg["bframes"] + 2, check_qp=bool(cfg["aq_mode"]))


@pytest.mark.parametrize("depth", [8, 10])
def test_me_search_full_vs_golden(depth):
    """The oracle's main-encode motion search against results recorded from x264_me_search_ref (make_golden.py gen_me_full)."""
    from tests.common import ME_METHODS, ME_SIZES, oracle_me_search
    z = np.load(os.path.join(GOLD, "me_full_d%d.npz" % depth))
    o = Oracle(depth)
    planes = [np.ascontiguousarray(z["planes"][p]) for p in range(4)]
    integral = np.ascontiguousarray(z["integral"])
    cost_mv = np.ascontiguousarray(z["cost_mv"])
    frame = z["fenc_frame"]
    for me in ME_METHODS:
        for call in z["calls_%s" % me]:
            i_pixel, mb_x, mb_y, xoff, yoff, subme = (int(v) for v in call[:6])
            bw, bh = ME_SIZES[i_pixel]
            fenc = np.zeros((16, 16), o.dtype)
            sy, sx = 16 * mb_y + yoff, 16 * mb_x + xoff
            fenc[:bh, :bw] = frame[sy:sy + bh, sx:sx + bw]
            got = oracle_me_search(o, me, planes, integral if ME_METHODS[me] >= 3 else None, cost_mv, z["geom"], fenc, call)
            want = call[18:22]
            n = 4 if subme >= 2 else 3   # cost_mv is only defined when refine_subpel ran
            assert np.array_equal(got[:n], want[:n]), (me, call.tolist(), got.tolist())
