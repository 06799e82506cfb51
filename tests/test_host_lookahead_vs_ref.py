"""End-to-end lookahead parity on CPU: the product's host logic (lookahead_host.cpp) driven by the oracle
backend, against the REAL reference lookahead (oracle/_ref) -- slice types, coded order and every evaluated
i_cost_est / i_cost_est_aq cell must be identical."""
import numpy as np
import pytest

from oracle import refharness
from tests.common import clip, nearest_ref_cells  # noqa: F401
from tests.oracle_backend import OracleBackend
from x264_amd import lib
from x264_amd.synth import make_clip

pytestmark = pytest.mark.skipif(not refharness.available(8), reason="oracle/_ref not built (no /root/reference)")

CASES = [
    # (preset, ref opts, cfg overrides, depth, clip kwargs, n_frames)
    ("medium", "", {}, 8, dict(seed=1, scene_cuts=(25,), fade=(40, 8, 0.6, 10)), 60),
    ("slow", "me=dia", dict(me="dia"), 8, dict(seed=2, scene_cuts=(13, 14, 31)), 64),
    ("slower", "me=umh,merange=32", dict(me="umh", me_range=32), 8, dict(seed=3, pan=(9, 5), scene_cuts=(40,)), 72),
    ("medium", "bframes=8,rc-lookahead=60", dict(bframes=8, rc_lookahead=60), 8, dict(seed=4, fade=(10, 12, 1.5, -20)), 75),
    ("veryslow", "me=tesa", dict(me="tesa"), 10, dict(seed=5, scene_cuts=(33,)), 70),
    ("medium", "keyint=24,min-keyint=4", dict(keyint_max=24, keyint_min=4), 8, dict(seed=6, scene_cuts=(7, 50)), 60),
    ("veryfast", "", {}, 8, dict(seed=7, pan=(1, 1), noise=1), 40),
    ("medium", "b-adapt=0", dict(b_adapt=0), 8, dict(seed=8), 30),
    ("medium", "bframes=0", dict(bframes=0), 8, dict(seed=9, scene_cuts=(11,)), 30),
    ("medium", "b-pyramid=none,weightp=0", dict(b_pyramid=0, weightp=0), 8, dict(seed=10), 40),
    ("medium", "open-gop=1,keyint=30", dict(open_gop=1, keyint_max=30), 8, dict(seed=11), 70),
    # found by randomized configuration fuzzing against oracle/_ref:
    # weightp=0 + mbtree + psy = WEIGHTP_FAKE (encoder.c:1121-1122): the luma weight search still runs and its cost ratio
    # enters macroblock_tree_finish as weightdelta (slicetype.c:462-463, 1032-1034)
    ("fast", "bframes=1,b-adapt=1,b-pyramid=normal,keyint=24,min-keyint=0,rc-lookahead=10,weightp=0,open-gop=1",
     dict(bframes=1, b_adapt=1, b_pyramid=2, keyint_max=24, keyint_min=0, rc_lookahead=10, weightp=0, open_gop=1), 8,
     dict(seed=840, scene_cuts=(24, 36), pan=(3, 1), fade=(2, 10, 1.5, 5)), 50),
    ("fast", "bframes=4,b-adapt=2,b-pyramid=strict,keyint=12,min-keyint=0,rc-lookahead=40,weightp=0",
     dict(bframes=4, b_adapt=2, b_pyramid=1, keyint_max=12, keyint_min=0, rc_lookahead=40, weightp=0), 8,
     dict(seed=178, scene_cuts=(28,), pan=(3, 0), fade=(15, 10, 1.5, -4)), 48),
    # no B-frames: open GOPs, adaptive placement and weighted bi-prediction are switched off (encoder.c:1080-1086)
    ("faster", "bframes=0,open-gop=1,keyint=20", dict(bframes=0, open_gop=1, keyint_max=20), 8, dict(seed=12, scene_cuts=(9,)), 45),
    # no MB-tree (qcomp=1, encoder.c:1126-1127): slicetype_slice_cost skips the outermost ring of blocks (slicetype.c:823-833),
    # the delay shrinks to the B-frame count, and f_qp_offset is the plain AQ map
    ("veryfast", "bframes=2,keyint=24,scenecut=80,rc-lookahead=20,aq-strength=1.5,qcomp=1,subme=7,me=dia",
     dict(bframes=2, keyint_max=24, scenecut=80, rc_lookahead=20, aq_strength=1.5, qcompress=1.0, subme=7, me="dia"), 8,
     dict(seed=874, pan=(4, 3)), 47),
    ("medium", "mbtree=0,bframes=5,b-adapt=2,rc-lookahead=30", dict(mb_tree=0, bframes=5, b_adapt=2, rc_lookahead=30), 8,
     dict(seed=14, scene_cuts=(19,), fade=(30, 8, 0.6, 5)), 50),
    # constant QP (encoder.c:951-966): no AQ, no MB-tree, no costs ahead of time for rate control; the P-frame weight analysis of
    # slicetype_decide still fills the intra cell (slicetype.c:1937-1943, :365-370)
    ("slow", "bframes=1,b-adapt=0,keyint=8,scenecut=0,rc-lookahead=5,subme=1,qp=24",
     dict(bframes=1, b_adapt=0, keyint_max=8, scenecut=0, rc_lookahead=5, subme=1, rc_is_cqp=1), 8,
     dict(seed=526, scene_cuts=(21, 40), pan=(3, 3), fade=(9, 10, 0.6, 12)), 46),
    # auto-variance AQ (ratecontrol.c:354-393)
    ("medium", "aq-mode=2,aq-strength=1.5", dict(aq_mode=2, aq_strength=1.5), 8, dict(seed=15, scene_cuts=(22,)), 40),
    ("fast", "aq-mode=3,aq-strength=0.5,qcomp=0.4", dict(aq_mode=3, aq_strength=0.5, qcompress=0.4), 10,
     dict(seed=24, scene_cuts=(30,), pan=(2, 3), fade=(10, 10, 0.6, 15)), 44),
    # lookahead threads: two bands searched independently (slicetype.c:668, :917-918)
    ("medium", "threads=4,sync-lookahead=0,lookahead-threads=2", dict(threads=4, lookahead_threads=2), 8,
     dict(seed=16, pan=(23, 11), noise=30, texture=0.9, scene_cuts=(20,)), 40),
    # intra refresh: no key frames after the first, its own scenecut bias (slicetype.c:1405,1506,1681,1831; encoder.c:1087-1102)
    ("medium", "intra-refresh=1,keyint=12,bframes=5,b-pyramid=normal,ref=4", dict(intra_refresh=1, keyint_max=12, bframes=5, b_pyramid=2, frame_refs=4),
     8, dict(seed=9, scene_cuts=(13, 37), pan=(2, 1)), 60),
    ("medium", "intra-refresh=1,keyint=10,mbtree=0", dict(intra_refresh=1, keyint_max=10, mb_tree=0), 8, dict(seed=9, scene_cuts=(13, 37), pan=(2, 1)), 60),
    # lookahead-less MB-tree (rc-lookahead 0 is only kept with infinite keyint or intra refresh, encoder.c:1128-1133): the propagation
    # is extrapolated across calls by exchanging accumulators with the window's first frame (slicetype.c:1112-1124, :1173-1178)
    ("medium", "keyint=infinite,rc-lookahead=0", dict(keyint_max=1 << 30, rc_lookahead=0), 8, dict(seed=3, scene_cuts=(21,), pan=(3, 1)), 40),
    ("medium", "intra-refresh=1,rc-lookahead=0,keyint=30", dict(intra_refresh=1, rc_lookahead=0, keyint_max=30), 8,
     dict(seed=3, scene_cuts=(21,), pan=(3, 1)), 40),
    ("fast", "keyint=infinite,rc-lookahead=0,bframes=0", dict(keyint_max=1 << 30, rc_lookahead=0, bframes=0), 8,
     dict(seed=3, scene_cuts=(21,), pan=(3, 1)), 40),
    # more B-frames than the key-frame interval allows (encoder.c:1074)
    ("medium", "bframes=16,keyint=8,rc-lookahead=5", dict(bframes=16, keyint_max=8, rc_lookahead=5), 8, dict(seed=17), 40),
]


@pytest.mark.parametrize("preset,opts,over,depth,ckw,nf", CASES)
def test_lookahead_matches_reference(preset, opts, over, depth, ckw, nf):
    W, H = 176, 144
    frames = make_clip(W, H, nf, bit_depth=depth, **ckw)
    r = refharness.Ref(W, H, preset, opts=opts, bit_depth=depth)
    try:
        ref = r.lookahead_run(frames, with_qp_offsets=True)
        rc = r.cfg
        rc["weighted_bipred"] = int(bool(rc["weighted_bipred"]))
        cfg = lib.la_config(W, H, preset, bit_depth=depth, **over)
        # the derived configuration must be what the reference validated
        for k, rk in (("bframes", "bframes"), ("b_adapt", "b_adapt"), ("rc_lookahead", "rc_lookahead"), ("mv_range", "mv_range"),
                      ("me_range", "me_range"), ("la_me_method", "me_method"), ("la_subpel_refine", "subpel_refine"),
                      ("subme", "subme"), ("weightp", "weightp"), ("mb_tree", "mb_tree"), ("keyint_max", "keyint_max"),
                      ("keyint_min", "keyint_min"), ("b_pyramid", "b_pyramid"), ("mbcmp_satd", "mbcmp_satd"),
                      ("fpelcmp_satd", "fpelcmp_satd"), ("frame_refs", "refs"), ("open_gop", "open_gop"), ("aq_mode", "aq_mode"),
                      ("psy", "psy"), ("weighted_bipred", "weighted_bipred"), ("bframe_bias", "b_bias"),
                      ("lookahead_threads", "lookahead_threads")):
            assert cfg[k] == rc[rk], (k, cfg[k], rc[rk])
        be = OracleBackend(cfg)
        la = lib.Lookahead(cfg, backend=be.struct)
        assert la.delay == rc["delay"] - (cfg["threads"] - 1)  # frame threads add their own latency (encoder.c:1610)
        try:
            outs = la.run(frames, qp_offsets=True)
        finally:
            la.close()
    finally:
        r.close()
    assert len(outs) == nf
    got_idx = [o.frame for o in outs]
    got_type = [o.type for o in outs]
    assert got_idx == list(ref["idx"]), "coded order differs"
    assert got_type == list(ref["type"]), "slice types differ"
    nb = cfg["bframes"] + 2
    for k, o in enumerate(outs):
        ce = np.array([[o.cost_est[i][j] for j in range(nb)] for i in range(nb)])
        ca = np.array([[o.cost_est_aq[i][j] for j in range(nb)] for i in range(nb)])
        rce = ref["cost"][k][:nb, :nb]
        assert np.array_equal(ce, rce), ("i_cost_est", k, o.frame)
        m = rce >= 0
        assert np.array_equal(ca[m], ref["cost_aq"][k][:nb, :nb][m]), ("i_cost_est_aq", k, o.frame)
        assert np.array_equal(o.qp_offset, ref["qp_offset"][k]), ("f_qp_offset", k, o.frame, o.type)
    assert be.n_eval > (0 if cfg["rc_is_cqp"] else nf)  # constant QP: only the intra cells of weighted P candidates
    _check_classes(cfg, be)


def _check_classes(cfg, be):
    """what x264hip_lookahead_open tells its device context about this flow (x264hip_lookahead_classes -> x264hip_spec_classes): every cell
    the decisions asked the backend for lies inside it"""
    ok, m0, m1 = lib.lookahead_classes(cfg)
    outside = sorted(c for c in be.requested if not ok[c[0], c[1]])
    assert not outside, ("cells the lookahead ruled out were requested", outside, cfg["bframes"], cfg["b_pyramid"], cfg["b_adapt"])
    for d0, d1 in be.requested:
        assert (not d0 or m0 >> (d0 - 1) & 1) and (not d1 or m1 >> (d1 - 1) & 1), (d0, d1, m0, m1)


@pytest.mark.parametrize("b_adapt", [0, 1, 2])
@pytest.mark.parametrize("bframes,pyr", [(2, 2), (3, 1), (3, 2), (4, 2), (5, 1), (6, 2), (7, 2), (8, 1), (8, 2), (16, 2), (5, 0)])
def test_requested_cell_classes_stay_inside_the_statement(bframes, pyr, b_adapt):
    """The lookahead states ahead of time which (d0, d1) cells its decisions can ask for (with B-pyramid: the middle frame of a run of
    B-frames and pairs inside one half; slicetype.c:1062-1095, :1120-1160, :1922-1933) so that nothing else is speculated on the device.
    Every request of whole runs over the oracle backend must lie inside, whatever the run lengths the content produces."""
    W, H = 64, 48
    nf = 44
    seen = set()
    for seed, ckw in ((5, dict(pan=(2, 1))), (6, dict(scene_cuts=(9, 30), fade=(14, 8, 0.6, 6))), (7, dict(pan=(0, 0), noise=2))):
        cfg = lib.la_config(W, H, "medium", bframes=bframes, b_adapt=b_adapt, b_pyramid=pyr, rc_lookahead=20, keyint_max=60, frame_refs=6)
        frames = make_clip(W, H, nf, seed=seed, **ckw)
        be = OracleBackend(cfg)
        la = lib.Lookahead(cfg, backend=be.struct)
        try:
            outs = la.run(frames)
        finally:
            la.close()
        assert len(outs) == nf
        _check_classes(cfg, be)
        seen |= be.requested
    if pyr and bframes >= 4:
        ok, _, _ = lib.lookahead_classes(cfg)
        assert ok.sum() < (bframes + 1) * (bframes + 2) // 2 + 1  # the statement does rule classes out


@pytest.mark.parametrize("paced", [True, False])
def test_speculative_weight_pairs_predict_every_request(paced):
    """The host computes the (scale, offset) candidate of every weight test ahead of time and announces the pairs
    (x264hip_prefetch_weight_costs); the decisions must not change and every weighted cost request that
    x264_weights_analyse makes later must be one of the announced pairs (same frames, same weight)."""
    W, H, nf = 176, 144, 70
    frames = make_clip(W, H, nf, seed=12, fade=(8, 14, 1.6, -25), scene_cuts=(44,))
    r = refharness.Ref(W, H, "medium", opts="")
    try:
        ref = r.lookahead_run(frames)
    finally:
        r.close()
    cfg = lib.la_config(W, H, "medium")
    be = OracleBackend(cfg, speculative=True)
    la = lib.Lookahead(cfg, backend=be.struct, max_frames=nf + 4)
    try:
        outs = la.run(frames, paced=paced) if paced else _run_unpaced(la, frames)
    finally:
        la.close()
    assert [o.frame for o in outs] == list(ref["idx"]) and [o.type for o in outs] == list(ref["type"])
    for o, c in zip(outs, ref["cost"]):
        got = np.array([[o.cost_est[i][j] for j in range(5)] for i in range(5)])
        assert np.array_equal(got, c[:5, :5])
    assert be.weighted_requests > 0, "the clip was meant to exercise weightp"
    assert be.weighted_predicted == be.weighted_requests
    # ... and every weighted SEARCH a P request triggered had been announced with that very weight before the request came
    # (x264hip_prefetch_weighted_fields: the verdict of the weight analysis taken ahead of time for every queued pair)
    assert be.weighted_searches > 0
    assert be.weighted_searches_predicted == be.weighted_searches, (be.weighted_searches_predicted, be.weighted_searches)


def _run_unpaced(la, frames):
    for f in frames:
        la.put(f)
    outs = []
    while True:
        o = la.get(True)
        if o is None:
            break
        outs.append(o)
    return outs


def test_chunked_submission_of_a_long_queue():
    """All frames queued up front, more than reach + chunk of them: the host submits the speculative work in chunks
    (flush_prefetch: reach of the next decision + 64 frames, the next chunk while half a chunk is still ahead) --
    decisions and cost cells must equal the encoder-paced reference run."""
    W, H, nf = 64, 48, 230
    frames = make_clip(W, H, nf, seed=13, scene_cuts=(60, 61, 150), fade=(100, 12, 0.7, 9), pan=(2, 1))
    r = refharness.Ref(W, H, "slow", opts="me=dia")
    try:
        ref = r.lookahead_run(frames)
    finally:
        r.close()
    cfg = lib.la_config(W, H, "slow", me="dia")
    be = OracleBackend(cfg, speculative=True)
    calls = []
    orig = be._prefetch
    def spy(user, slots, numbers, n):
        calls.append([numbers[i] for i in range(n)])
        return orig(user, slots, numbers, n)
    spy_fn = lib.PREFETCH_FN(spy)  # keep the callback object alive for the lifetime of the lookahead
    be.struct.prefetch = spy_fn
    la = lib.Lookahead(cfg, backend=be.struct, max_frames=nf + 4)
    la.set_chunk(64)
    try:
        outs = _run_unpaced(la, frames)
    finally:
        la.close()
    assert [o.frame for o in outs] == list(ref["idx"]) and [o.type for o in outs] == list(ref["type"])
    for o, c in zip(outs, ref["cost"]):
        got = np.array([[o.cost_est[i][j] for j in range(5)] for i in range(5)])
        assert np.array_equal(got, c[:5, :5])
    # several submissions, each extending the previous one, none after everything has been submitted
    newest = [max(c) for c in calls]
    assert len(calls) >= 3 and newest == sorted(newest) and newest[-1] == nf - 1
    assert len(set(newest)) == len(newest), "a submission that added no frame"


PRESET_NAMES = ("ultrafast", "superfast", "veryfast", "faster", "fast", "medium", "slow", "slower", "veryslow", "placebo")
TUNE_NAMES = ("", "film", "animation", "grain", "stillimage", "psnr", "ssim", "fastdecode", "zerolatency", "touhou")
_CFG_KEYS = (("bframes", "bframes"), ("b_adapt", "b_adapt"), ("rc_lookahead", "rc_lookahead"), ("keyint_max", "keyint_max"),
             ("keyint_min", "keyint_min"), ("b_pyramid", "b_pyramid"), ("weightp", "weightp"), ("open_gop", "open_gop"),
             ("aq_mode", "aq_mode"), ("mb_tree", "mb_tree"), ("psy", "psy"), ("la_me_method", "me_method"),
             ("la_subpel_refine", "subpel_refine"), ("mbcmp_satd", "mbcmp_satd"), ("fpelcmp_satd", "fpelcmp_satd"),
             ("mv_range", "mv_range"), ("me_range", "me_range"), ("frame_refs", "refs"), ("scenecut", "scenecut"))


@pytest.mark.parametrize("preset", PRESET_NAMES)
def test_every_preset_and_tune(preset):
    """x264_param_default_preset / x264_param_apply_tune (common/base.c:496-700) + validate_parameters for every preset x tune:
    the derived configuration, the delay, the decisions, the cost cells and the quantiser offset maps of a short clip."""
    W, H, nf = 176, 144, 36
    frames = make_clip(W, H, nf, seed=5, scene_cuts=(17,), fade=(22, 8, 0.6, 8))
    for tune in TUNE_NAMES:
        r = refharness.Ref(W, H, preset, tune=tune)
        try:
            ref = r.lookahead_run(frames, with_qp_offsets=True)
            rc = r.cfg
        finally:
            r.close()
        cfg = lib.la_config(W, H, preset, tune=tune)
        for k, rk in _CFG_KEYS:
            assert cfg[k] == rc[rk], (preset, tune, k, cfg[k], rc[rk])
        assert bool(cfg["weighted_bipred"]) == bool(rc["weighted_bipred"])
        assert abs(cfg["aq_strength"] * 65536 - rc["aq_strength_q16"]) <= 1
        be = OracleBackend(cfg)
        la = lib.Lookahead(cfg, backend=be.struct, max_frames=nf + 4)
        try:
            assert la.delay == rc["delay"], (preset, tune)
            outs = la.run(frames, qp_offsets=True)
        finally:
            la.close()
        assert [o.frame for o in outs] == list(ref["idx"]), (preset, tune)
        assert [o.type for o in outs] == list(ref["type"]), (preset, tune)
        _check_classes(cfg, be)
        nb = cfg["bframes"] + 2
        for k, o in enumerate(outs):
            ce = np.array([[o.cost_est[i][j] for j in range(nb)] for i in range(nb)])
            assert np.array_equal(ce, ref["cost"][k][:nb, :nb]), (preset, tune, "i_cost_est", o.frame)
            assert np.array_equal(o.qp_offset, ref["qp_offset"][k]), (preset, tune, "f_qp_offset", o.frame)


@pytest.mark.parametrize("opts,over,forced", [
    ("", {}, {5: 1, 11: 1, 24: 7, 38: 3}),                       # IDRs, an unknown type (= AUTO), a forced P
    ("bframes=8,b-adapt=0,b-pyramid=strict,keyint=60", dict(bframes=8, b_adapt=0, b_pyramid=1, keyint_max=60), {30: 6, 34: 5, 35: 6}),
    ("open-gop=1,b-pyramid=normal,bframes=5", dict(open_gop=1, b_pyramid=2, bframes=5), {9: 6, 10: 4, 11: 4, 12: 4, 20: 2, 21: 5, 33: 1}),
    ("b-adapt=2,keyint=30", dict(b_adapt=2, keyint_max=30), {3: 5, 4: 5, 5: 5, 6: 5, 7: 5, 29: 5, 30: 5, 40: 2}),
])
def test_forced_picture_types(opts, over, forced):
    """x264_picture_t.i_type of the input pictures (x264.h:274-280): forced IDR / I / P / BREF / B / KEYFRAME requests, the
    corrections slicetype_decide applies to impossible ones (slicetype.c:1803-1885) and unknown values (frame.c:392-400)."""
    W, H, nf = 176, 144, 50
    frames = make_clip(W, H, nf, seed=31, scene_cuts=(17,), pan=(3, 1))
    ft = np.zeros(nf, np.int32)
    for k, v in forced.items():
        ft[k] = v
    r = refharness.Ref(W, H, "medium", opts=opts)
    try:
        ref = r.lookahead_run(frames, with_qp_offsets=True, forced_types=ft)
    finally:
        r.close()
    cfg = lib.la_config(W, H, "medium", **over)
    for paced in (True, False):
        be = OracleBackend(cfg, speculative=not paced)
        la = lib.Lookahead(cfg, backend=be.struct, max_frames=nf + 4)
        try:
            outs = la.run(frames, qp_offsets=True, paced=paced, forced_types=ft)
        finally:
            la.close()
        assert [o.frame for o in outs] == list(ref["idx"])
        assert [o.type for o in outs] == list(ref["type"])
        nb = cfg["bframes"] + 2
        for k, o in enumerate(outs):
            ce = np.array([[o.cost_est[i][j] for j in range(nb)] for i in range(nb)])
            assert np.array_equal(ce, ref["cost"][k][:nb, :nb]), ("i_cost_est", o.frame)
            assert np.array_equal(o.qp_offset, ref["qp_offset"][k]), ("f_qp_offset", o.frame)


@pytest.mark.parametrize("preset,opts,over", [
    ("medium", "bitrate=500,vbv-bufsize=300,vbv-maxrate=600", dict(bitrate=500, vbv_bufsize=300, vbv_maxrate=600)),
    ("medium", "vbv-bufsize=300,vbv-maxrate=600,mbtree=0,b-pyramid=none", dict(vbv_bufsize=300, vbv_maxrate=600, mb_tree=0, b_pyramid=0)),
    ("veryfast", "vbv-bufsize=100,vbv-maxrate=600,aq-mode=0", dict(vbv_bufsize=100, vbv_maxrate=600, aq_mode=0)),
    ("superfast", "bframes=5,b-adapt=1,b-pyramid=strict,keyint=60,scenecut=0,rc-lookahead=60,open-gop=1,aq-mode=3,vbv-bufsize=20,"
     "vbv-maxrate=200,bitrate=400", dict(bframes=5, b_adapt=1, b_pyramid=1, keyint_max=60, scenecut=0, rc_lookahead=60, open_gop=1,
                                          aq_mode=3, vbv_bufsize=20, vbv_maxrate=200, bitrate=400)),
    ("fast", "b-adapt=2,bframes=5,vbv-bufsize=2000,vbv-maxrate=1000,rc-lookahead=20,keyint=24",
     dict(b_adapt=2, bframes=5, vbv_bufsize=2000, vbv_maxrate=1000, rc_lookahead=20, keyint_max=24)),
    ("medium", "vbv-bufsize=300,bitrate=400,rc-lookahead=0", dict(vbv_bufsize=300, bitrate=400, rc_lookahead=0)),
])
@pytest.mark.parametrize("paced", [True, False])
def test_vbv_lookahead(preset, opts, over, paced):
    """VBV configurations (vbv_lookahead / vbv_frame_cost, slicetype.c:1186-1286; the extra evaluations of slicetype_decide,
    :1916-1934; the per-reference MB-tree finish, :1087-1088): decisions, cost cells, f_qp_offset, i_planned_type / i_planned_satd
    of every non-B frame, and the i_row_satds of the cell each frame is coded with plus its intra rows."""
    W, H, nf = 100, 70, 46
    frames = make_clip(W, H, nf, seed=101, scene_cuts=(32, 38), pan=(2, 0), fade=(8, 10, 0.6, 14))
    r = refharness.Ref(W, H, preset, opts=opts)
    try:
        first = r.lookahead_run(frames)
    finally:
        r.close()
    cells = nearest_ref_cells(first["idx"], first["type"])
    r = refharness.Ref(W, H, preset, opts=opts)
    try:
        ref = r.lookahead_run(frames, with_qp_offsets=True, with_vbv=True, rc_cells=np.array(cells, np.int32))
        rc = r.cfg
    finally:
        r.close()
    cfg = lib.la_config(W, H, preset, **over)
    assert cfg["vbv"] == int(rc["vbv"] > 0) and cfg["rc_lookahead"] == rc["rc_lookahead"] and cfg["mv_range"] == rc["mv_range"]
    be = OracleBackend(cfg, speculative=not paced)
    la = lib.Lookahead(cfg, backend=be.struct, max_frames=nf + 4)
    try:
        assert la.delay == rc["delay"]
        outs = la.run(frames, qp_offsets=True, vbv=True, paced=paced)
    finally:
        la.close()
    assert [o.frame for o in outs] == list(ref["idx"])
    assert [o.type for o in outs] == list(ref["type"])
    nb = cfg["bframes"] + 2
    for k, o in enumerate(outs):
        ce = np.array([[o.cost_est[i][j] for j in range(nb)] for i in range(nb)])
        assert np.array_equal(ce, ref["cost"][k][:nb, :nb]), ("i_cost_est", o.frame)
        if cfg["aq_mode"]:
            assert np.array_equal(o.qp_offset, ref["qp_offset"][k]), ("f_qp_offset", o.frame, o.type)
        if o.type not in (4, 5):
            want = []
            for t, s in zip(ref["planned_type"][k], ref["planned_satd"][k]):
                if t == 0:
                    break
                want.append((int(t), int(s)))
            assert o.planned == want, ("i_planned_*", o.frame)
        # the real x264_rc_analyse_slice on the leaving frame (slicetype.c:1976-2009): cell, cost, row sums (rewritten by the
        # MB-tree recalculation where that is on); without MB-tree they are the raw sums of the evaluations
        mbh = (H + 15) // 16
        assert o.own_cell == cells[k], ("cell", o.frame, o.type)
        assert o.rc_satd == ref["rc"][k][0], ("rc satd", o.frame, o.type)
        assert np.array_equal(o.row_satds, ref["rc"][k][1:1 + mbh]), ("i_row_satd", o.frame, o.type)
        if o.type not in (1, 2):
            assert np.array_equal(o.row_satds_intra, ref["rc"][k][1 + mbh:]), ("i_row_satds[0][0]", o.frame, o.type)
        if not cfg["mb_tree"]:
            d0, d1 = o.own_cell
            assert np.array_equal(o.row_satds, ref["row_satds"][k][d0][d1]), ("raw i_row_satds", o.frame)


def test_level_mv_range():
    """param.analyse.i_mv_range of the automatically chosen level (encoder.c:1243-1268, x264_validate_levels): frame size, DPB,
    VBV rate / buffer per profile, MB rate."""
    for (W, H), preset, depth, (opts, over) in [
            ((96, 80), "medium", 8, ("", {})), ((352, 288), "veryslow", 10, ("", {})), ((720, 576), "medium", 8, ("bitrate=300", dict(bitrate=300))),
            ((1280, 720), "ultrafast", 8, ("", {})), ((1920, 1080), "slow", 8, ("", {})), ((3840, 2160), "slower", 8, ("", {})),
            ((3840, 2160), "medium", 8, ("", {})), ((7680, 4320), "veryslow", 10, ("", {})), ((100, 2000), "medium", 8, ("", {})),
            ((352, 288), "medium", 8, ("bitrate=3000,vbv-bufsize=6000,vbv-maxrate=9000", dict(bitrate=3000, vbv_bufsize=6000, vbv_maxrate=9000))),
            ((176, 144), "medium", 8, ("keyint=1", dict(keyint_max=1))), ((640, 360), "placebo", 8, ("fps=60", dict(fps_num=60, fps_den=1)))]:
        if not refharness.available(depth):
            continue
        r = refharness.Ref(W, H, preset, opts=opts, bit_depth=depth)
        want = r.cfg["mv_range"]
        r.close()
        assert lib.la_config(W, H, preset, bit_depth=depth, **over)["mv_range"] == want, (W, H, preset, depth, opts)


@pytest.mark.parametrize("preset,opts,over", [
    ("medium", "", {}), ("veryslow", "", {}), ("superfast", "", {}), ("fast", "b-pyramid=strict,bframes=6,b-adapt=2", dict(b_pyramid=1, bframes=6, b_adapt=2)),
    ("medium", "open-gop=1,keyint=20,aq-mode=0", dict(open_gop=1, keyint_max=20, aq_mode=0)),
])
def test_rc_analyse_slice_outputs(preset, opts, over):
    """Without VBV: the frame complexity ABR / CRF rate control reads (x264_rc_analyse_slice run by the reference on every
    leaving I / P frame; B frames are only analysed with VBV, ratecontrol.c:2472-2474) and the cell each frame is coded with."""
    W, H, nf = 176, 144, 40
    frames = make_clip(W, H, nf, seed=41, scene_cuts=(19,), pan=(3, 2))
    r = refharness.Ref(W, H, preset, opts=opts)
    first = r.lookahead_run(frames)
    r.close()
    cells = nearest_ref_cells(first["idx"], first["type"])
    r = refharness.Ref(W, H, preset, opts=opts)
    ref = r.lookahead_run(frames, rc_cells=np.array(cells, np.int32))
    r.close()
    cfg = lib.la_config(W, H, preset, **over)
    be = OracleBackend(cfg)
    la = lib.Lookahead(cfg, backend=be.struct, max_frames=nf + 4)
    try:
        outs = la.run(frames, vbv=True)
    finally:
        la.close()
    assert [o.frame for o in outs] == list(ref["idx"])
    for k, o in enumerate(outs):
        assert o.own_cell == cells[k], (o.frame, o.type)
        assert o.rc_satd == (-1 if o.type in (4, 5) else ref["rc"][k][0]), (o.frame, o.type)
        assert o.planned == []


def _run_collect(la, frames, **kw):
    outs = la.run(frames, qp_offsets=True, **kw)
    return [(o.frame, o.type, o.bframes, o.keyframe) for o in outs], \
           [np.array(o.cost_est) for o in outs], [o.qp_offset.copy() for o in outs]


@pytest.mark.parametrize("preset,over", [("medium", {}), ("fast", dict(b_adapt=2, bframes=5, open_gop=1, keyint_max=30))])
def test_reset_starts_a_new_sequence(preset, over):
    """x264hip_lookahead_reset: after a complete sequence, and in the middle of one (frames still queued and undecided), the next
    sequence gives exactly what a fresh context gives -- frame numbering, key-frame distance, last_nonb and slot bookkeeping all
    start over (bench.py reuses one context per GOP segment this way)."""
    W, H = 176, 144
    a = make_clip(W, H, 37, seed=51, scene_cuts=(14,), pan=(4, 1))
    b = make_clip(W, H, 45, seed=52, scene_cuts=(30,), fade=(8, 8, 1.5, -10))
    cfg = lib.la_config(W, H, preset, **over)
    fresh = lib.Lookahead(cfg, backend=OracleBackend(cfg).struct, max_frames=64)
    try:
        want = _run_collect(fresh, b)
    finally:
        fresh.close()
    be = OracleBackend(cfg)
    la = lib.Lookahead(cfg, backend=be.struct, max_frames=64)
    try:
        _run_collect(la, a)                    # a whole sequence, drained
        la.reset()
        got = _run_collect(la, b, paced=False)
        for i in range(20):                    # a sequence abandoned half way: frames queued, some decided, none drained
            la.put(a[i])
            la.get()
        la.reset()
        got2 = _run_collect(la, b)
    finally:
        la.close()
    for g in (got, got2):
        assert g[0] == want[0]
        assert all(np.array_equal(x, y) for x, y in zip(g[1], want[1]))
        assert all(np.array_equal(x, y) for x, y in zip(g[2], want[2]))


def test_api_misuse_is_reported():
    W, H = 96, 80
    cfg = lib.la_config(W, H, "medium")
    be = OracleBackend(cfg)
    la = lib.Lookahead(cfg, backend=be.struct, max_frames=cfg["rc_lookahead"] + cfg["bframes"] + 8)
    try:
        assert la.get() is None and la.get(flush=True) is None            # nothing queued: not an error, just nothing
        assert la.L.x264hip_lookahead_delayed_frames(la.h) == 0
        fr = make_clip(W, H, 80, seed=1)
        with pytest.raises(lib.X264HipError):                             # more frames than slots without draining
            for i in range(80):
                la.put(fr[i])
        la.reset()                                                        # the context stays usable
        for i in range(10):
            la.put(fr[i])
        assert la.L.x264hip_lookahead_delayed_frames(la.h) == 10
        la.reset()
        outs = la.run(fr[:30])
        assert [o.frame for o in sorted(outs, key=lambda o: o.frame)] == list(range(30))
        assert la.L.x264hip_lookahead_delayed_frames(la.h) == 0
    finally:
        la.close()
    import ctypes as C
    L = lib.load()
    h = C.c_void_p()
    p = lib.make_la_params(cfg)
    p.b_adapt = 3
    assert L.x264hip_lookahead_open_backend(C.byref(h), C.byref(p), C.byref(be.struct)) == -2


@pytest.mark.parametrize("preset,opts,over,stamps,paced", [
    ("medium", "vfr-input=1", dict(vfr_input=1), "frames", True),
    ("medium", "vfr-input=1", dict(vfr_input=1), "steps", True),
    ("medium", "vfr-input=1,timebase=1/1000", dict(vfr_input=1, timebase_num=1, timebase_den=1000), "ms", True),
    ("fast", "vfr-input=1,timebase=1/1000,b-adapt=2,bframes=5,fps=30000/1001",
     dict(vfr_input=1, timebase_num=1, timebase_den=1000, b_adapt=2, bframes=5, fps_num=30000, fps_den=1001), "ms", False),
    ("medium", "vfr-input=1,timebase=1001/30000,vbv-bufsize=300,vbv-maxrate=600",
     dict(vfr_input=1, timebase_num=1001, timebase_den=30000, vbv_bufsize=300, vbv_maxrate=600), "steps", True),
])
def test_vfr_input_durations(preset, opts, over, stamps, paced):
    """b_vfr_input: frame durations from the time stamps (slicetype.c:1755-1771) weigh the MB-tree propagation (:1031,1063,
    1098-1101); one more frame of delay (encoder.c:1612).  Decisions, cost cells and f_qp_offset against the reference."""
    W, H, nf = 176, 144, 50
    rng = np.random.default_rng(1)
    pts = {"frames": np.arange(nf), "steps": np.cumsum(rng.choice([1, 1, 2, 3], size=nf)),
           "ms": np.cumsum(rng.choice([33, 34, 40, 66, 17], size=nf))}[stamps].astype(np.int64)
    frames = make_clip(W, H, nf, seed=7, scene_cuts=(23,), pan=(3, 1))
    r = refharness.Ref(W, H, preset, opts=opts)
    try:
        ref = r.lookahead_run(frames, with_qp_offsets=True, pts=pts)
        rc = r.cfg
    finally:
        r.close()
    cfg = lib.la_config(W, H, preset, **over)
    la = lib.Lookahead(cfg, backend=OracleBackend(cfg).struct, max_frames=nf + 6)
    try:
        assert la.delay == rc["delay"]
        outs = la.run(frames, qp_offsets=True, pts=pts, paced=paced)
    finally:
        la.close()
    assert [o.frame for o in outs] == list(ref["idx"])
    assert [o.type for o in outs] == list(ref["type"])
    nb = cfg["bframes"] + 2
    for k, o in enumerate(outs):
        assert np.array_equal(np.array(o.cost_est)[:nb, :nb], ref["cost"][k][:nb, :nb]), o.frame
        assert np.array_equal(o.qp_offset, ref["qp_offset"][k]), ("f_qp_offset", o.frame, o.type)


@pytest.mark.parametrize("preset,opts,over,depth,W,H", [
    ("medium", "", {}, 8, 176, 144),
    ("medium", "aq-mode=2", dict(aq_mode=2), 10, 176, 144),
    ("fast", "aq-mode=3,mbtree=0", dict(aq_mode=3, mb_tree=0), 8, 100, 70),
])
def test_chroma_planes_enter_adaptive_quant(preset, opts, over, depth, W, H):
    """x264hip_lookahead_put_picture: adaptive quantisation adds the AC energy of Cb and Cr to the luma energy of every macroblock
    (ac_energy_mb, ratecontrol.c:258-276), so i_inv_qscale_factor, the AQ-weighted costs, MB-tree and f_qp_offset all depend on the
    chroma planes.  Against the reference fed with the same 4:2:0 pictures (and the reference fed with grey chroma differs)."""
    from x264_amd.synth import make_chroma
    nf = 40
    frames = make_clip(W, H, nf, seed=9, bit_depth=depth, scene_cuts=(17,), pan=(3, 1))
    chroma = make_chroma(W, H, nf, seed=9, bit_depth=depth)
    r = refharness.Ref(W, H, preset, opts=opts, bit_depth=depth)
    ref = r.lookahead_run(frames, with_qp_offsets=True, chroma=chroma)
    r.close()
    r = refharness.Ref(W, H, preset, opts=opts, bit_depth=depth)
    grey = r.lookahead_run(frames, with_qp_offsets=True)
    r.close()
    assert not np.array_equal(ref["qp_offset"], grey["qp_offset"])
    cfg = lib.la_config(W, H, preset, bit_depth=depth, **over)
    la = lib.Lookahead(cfg, backend=OracleBackend(cfg).struct, max_frames=nf + 4)
    try:
        outs = la.run(frames, qp_offsets=True, chroma=chroma)
    finally:
        la.close()
    assert [o.frame for o in outs] == list(ref["idx"]) and [o.type for o in outs] == list(ref["type"])
    nb = cfg["bframes"] + 2
    for k, o in enumerate(outs):
        assert np.array_equal(np.array(o.cost_est)[:nb, :nb], ref["cost"][k][:nb, :nb])
        m = ref["cost"][k][:nb, :nb] >= 0
        assert np.array_equal(np.array(o.cost_est_aq)[:nb, :nb][m], ref["cost_aq"][k][:nb, :nb][m])
        assert np.array_equal(o.qp_offset, ref["qp_offset"][k])


def test_put_pictures_batch_equals_one_by_one():
    """x264hip_lookahead_put_pictures (all pictures of a clip in one call, with chroma, forced types and time stamps) gives what
    x264hip_lookahead_put_picture gives picture by picture."""
    from x264_amd.synth import make_chroma
    W, H, nf = 176, 144, 36
    frames = make_clip(W, H, nf, seed=61, scene_cuts=(15,), pan=(2, 2))
    cb, cr = make_chroma(W, H, nf, seed=61)
    types = np.zeros(nf, np.int32); types[10] = 1; types[20] = 5
    pts = np.cumsum(np.random.default_rng(2).choice([1, 2, 3], size=nf)).astype(np.int64)
    cfg = lib.la_config(W, H, "medium", vfr_input=1)
    res = []
    for batch in (False, True):
        la = lib.Lookahead(cfg, backend=OracleBackend(cfg).struct, max_frames=nf + 6)
        try:
            if batch:
                la.put_pictures([f.ctypes.data for f in frames], W, [c.ctypes.data for c in cb], [c.ctypes.data for c in cr], cb.shape[2], types, pts)
                outs = []
                while True:
                    o = la.get(True, True)
                    if o is None:
                        break
                    outs.append(o)
            else:
                outs = la.run(frames, qp_offsets=True, chroma=(cb, cr), forced_types=types, pts=pts, paced=False)
        finally:
            la.close()
        res.append(outs)
    assert len(res[0]) == len(res[1]) == nf
    for a, b in zip(*res):
        assert (a.frame, a.type) == (b.frame, b.type)
        assert np.array_equal(np.array(a.cost_est), np.array(b.cost_est)) and np.array_equal(a.qp_offset, b.qp_offset)


@pytest.mark.parametrize("csp,fmt,depth,opts,over,W,H", [
    ("i422", 2, 8, "", {}, 176, 144), ("i444", 3, 8, "", {}, 176, 144),
    ("i422", 2, 10, "aq-mode=3", dict(aq_mode=3), 176, 144), ("i444", 3, 8, "aq-mode=2,bitrate=3000", dict(aq_mode=2, bitrate=3000), 100, 70),
])
def test_chroma_formats(csp, fmt, depth, opts, over, W, H):
    """4:2:2 and 4:4:4 pictures: adaptive quantisation measures 8x16 resp. 16x16 chroma blocks with their own shifts
    (ac_energy_plane, ratecontrol.c:238-256) and the profile raises the level limits (set.c:881-883); the rest of the lookahead only
    reads luma.  Decisions and f_qp_offset against the reference opened with that colour space."""
    nf = 36
    frames = make_clip(W, H, nf, seed=9, bit_depth=depth, scene_cuts=(17,), pan=(3, 1))
    maxv = (1 << depth) - 1
    rng = np.random.default_rng(3)
    dt = np.uint8 if depth == 8 else np.uint16
    cw, ch = (W if fmt == 3 else (W + 1) // 2), H
    cb = rng.integers(0, maxv + 1, size=(nf, ch, cw)).astype(dt)
    cr = np.clip(rng.normal(maxv / 2, maxv / 6, size=(nf, ch, cw)), 0, maxv).astype(dt)
    r = refharness.Ref(W, H, "medium", opts="csp=%s" % csp + ("," + opts if opts else ""), bit_depth=depth)
    try:
        ref = r.lookahead_run(frames, with_qp_offsets=True, chroma=(cb, cr))
        rc = r.cfg
    finally:
        r.close()
    cfg = lib.la_config(W, H, "medium", bit_depth=depth, chroma_format=fmt, **over)
    assert cfg["mv_range"] == rc["mv_range"]
    la = lib.Lookahead(cfg, backend=OracleBackend(cfg).struct, max_frames=nf + 4)
    try:
        outs = la.run(frames, qp_offsets=True, chroma=(cb, cr))
    finally:
        la.close()
    assert [o.frame for o in outs] == list(ref["idx"]) and [o.type for o in outs] == list(ref["type"])
    for k, o in enumerate(outs):
        assert np.array_equal(o.qp_offset, ref["qp_offset"][k]), (o.frame, o.type)


@pytest.mark.parametrize("preset,opts,over,depth", [
    ("medium", "", {}, 8), ("fast", "aq-mode=3,b-adapt=2", dict(aq_mode=3, b_adapt=2), 10),
    ("medium", "aq-strength=0", dict(aq_strength=0.0), 8),       # AQ "on with strength 0" for MB-tree: only the caller's offsets remain
    ("superfast", "", {}, 8),
])
def test_picture_quant_offsets(preset, opts, over, depth):
    """x264_picture_t.prop.quant_offsets (x264hip_picture.quant_offsets): per-macroblock offsets added to the adaptive-quantisation
    offsets of the picture (ratecontrol.c:318-326,396-397) -- they move i_inv_qscale_factor, the AQ-weighted costs, MB-tree and
    f_qp_offset.  Against the reference given the same offsets (region-of-interest style: a box of negative offsets that moves)."""
    from x264_amd.synth import make_chroma
    W, H, nf = 176, 144, 36
    frames = make_clip(W, H, nf, seed=71, bit_depth=depth, scene_cuts=(16,), pan=(2, 1))
    chroma = make_chroma(W, H, nf, seed=71, bit_depth=depth)
    mb_w, mb_h = (W + 15) // 16, (H + 15) // 16
    offs = np.zeros((nf, mb_h, mb_w), np.float32)
    rng = np.random.default_rng(5)
    for i in range(nf):
        x0 = (i // 3) % (mb_w - 3)
        offs[i, 2:6, x0:x0 + 4] = -6.0 + rng.normal(0, 0.5, size=(4, 4)).astype(np.float32)
        offs[i, 0, :] = 2.5
    offs = offs.reshape(nf, -1)
    r = refharness.Ref(W, H, preset, opts=opts, bit_depth=depth)
    try:
        ref = r.lookahead_run(frames, with_qp_offsets=True, chroma=chroma, quant_offsets=offs)
    finally:
        r.close()
    cfg = lib.la_config(W, H, preset, bit_depth=depth, **over)
    la = lib.Lookahead(cfg, backend=OracleBackend(cfg).struct, max_frames=nf + 4)
    try:
        outs = la.run(frames, qp_offsets=True, chroma=chroma, quant_offsets=offs)
    finally:
        la.close()
    assert [o.frame for o in outs] == list(ref["idx"]) and [o.type for o in outs] == list(ref["type"])
    nb = cfg["bframes"] + 2
    for k, o in enumerate(outs):
        assert np.array_equal(np.array(o.cost_est)[:nb, :nb], ref["cost"][k][:nb, :nb])
        m = ref["cost"][k][:nb, :nb] >= 0
        assert np.array_equal(np.array(o.cost_est_aq)[:nb, :nb][m], ref["cost_aq"][k][:nb, :nb][m])
        assert np.array_equal(o.qp_offset, ref["qp_offset"][k]), (o.frame, o.type)
