"""Device fuzz: tests/tools/fuzz/gpu1.py <seed> <n> [big]   (needs a GPU and oracle/_ref; "big": 540p ... 1080p pictures).
Random picture sizes (odd widths, non-mod-16, tiny), bit depths, presets and lookahead options; the whole lookahead on the device
(lib.Lookahead, paced or batched) against the real reference build (refharness.Ref.lookahead_run): slice types, every cost cell that
was evaluated, f_qp_offset.  Exercises the search, cell, intra, AQ, weight and MB-tree kernels over geometries the fixed tests do
not list."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np

from oracle import refharness
from tests.test_golden import check_lookahead_outputs
from x264_amd import lib
from x264_amd.synth import make_clip

SIZES = [(96, 80), (100, 70), (48, 32), (176, 144), (64, 48), (128, 272), (200, 120), (352, 288), (416, 240), (330, 190), (640, 360), (34, 34)]
BIG = [(960, 540), (1280, 720), (1000, 562), (1920, 1080), (1366, 768), (720, 1280)]


def run(seed, n, big=False, verbose=True):
    """n random configurations; returns the number that differ from the reference build"""
    rng = np.random.default_rng(int(seed))
    bad = 0
    sizes = BIG if big else SIZES
    say = print if verbose else (lambda *a, **k: None)
    for t in range(int(n)):
        W, H = sizes[int(rng.integers(0, len(sizes)))]
        depth = int(rng.choice([8, 8, 10]))
        bframes = int(rng.choice([0, 1, 2, 3, 5, 8])); b_adapt = int(rng.integers(0, 3)); pyr = int(rng.integers(0, 3))
        keyint = int(rng.choice([8, 24, 60, 250])); sc = int(rng.choice([0, 40, 80])); la = int(rng.choice([5, 20, 40, 60]))
        wp = int(rng.integers(0, 3)); og = int(rng.integers(0, 2)); aqm = int(rng.integers(0, 4)); mbt = int(rng.integers(0, 3) > 0)
        me = str(rng.choice(["dia", "hex", "umh", "tesa"])); subme = int(rng.choice([0, 1, 2, 4, 7, 9]))
        preset = str(rng.choice(["medium", "fast", "faster", "veryfast", "slow", "slower"]))
        if sc == 0 and b_adapt == 0 and mbt:
            sc = 40  # the reference reads uninitialised intra costs there (known divergences)
        if int(rng.integers(0, 8)) == 0:
            # lookahead-less MB-tree: rc-lookahead 0 survives validation only with an infinite key interval (encoder.c:1128-1133); the
            # propagation then carries over from call to call (X264HIP_MBT_SWAP / RESET_QP)
            la, keyint = 0, 1 << 30
        nf = int(rng.integers(12, 40))
        opts = "bframes=%d,b-adapt=%d,b-pyramid=%s,keyint=%s,scenecut=%d,rc-lookahead=%d,weightp=%d,open-gop=%d,aq-mode=%d,mbtree=%d,me=%s,subme=%d" % (
            bframes, b_adapt, ["none", "strict", "normal"][pyr], "infinite" if keyint == 1 << 30 else str(keyint), sc, la, wp, og, aqm, mbt, me, subme)
        over = dict(bframes=bframes, b_adapt=b_adapt, b_pyramid=pyr, keyint_max=keyint, scenecut=sc, rc_lookahead=la, weightp=wp, open_gop=og, aq_mode=aqm,
                    mb_tree=mbt, me=me, subme=subme)
        ckw = dict(seed=int(rng.integers(0, 1000)), scene_cuts=tuple(sorted(int(x) for x in rng.integers(3, nf, size=int(rng.integers(0, 3))))),
                   pan=(int(rng.integers(0, 6)), int(rng.integers(0, 4))))
        if rng.integers(0, 2):
            ckw["fade"] = (int(rng.integers(2, max(3, nf - 12))), 10, float(rng.choice([0.6, 1.5])), int(rng.integers(-20, 20)))
        paced = bool(rng.integers(0, 2))
        desc = (preset, W, H, depth, opts, nf, ckw, "paced" if paced else "batched")
        say("TRY", *desc, flush=True)
        try:
            frames = make_clip(W, H, nf, bit_depth=depth, **ckw)
            r = refharness.Ref(W, H, preset, opts=opts, bit_depth=depth)
            try:
                ref = r.lookahead_run(frames, with_qp_offsets=True)
            finally:
                r.close()
            cfg = lib.la_config(W, H, preset, bit_depth=depth, **over)
            dev = lib.Lookahead(cfg, max_frames=0 if paced else nf + 4)
            try:
                outs = dev.run(frames, paced=paced, qp_offsets=True)
                # what x264hip_lookahead_open said about its flow (x264hip_spec_classes) must hold: no request for a class it ruled out
                fr, cr, ca, fa = dev.class_requests()
                bf = cfg["bframes"]  # (validation may have lowered it, e.g. under a short key interval)
                ruled_out = [(d0, d1) for d0 in range(bf + 2) for d1 in range(bf + 2) if cr[d0, d1] and not ca[d0, d1]]
                ruled_out += [("L%d" % l, d + 1) for l in range(2) for d in range(bf + 1) if fr[l, d] and not (int(fa[l]) >> d) & 1]
                assert not ruled_out, "requests for classes the lookahead ruled out: %s" % ruled_out
            finally:
                dev.close()
            nb = cfg["bframes"] + 2
            z = dict(idx=ref["idx"], type=ref["type"], cost=ref["cost"][:, :nb, :nb], cost_aq=ref["cost_aq"][:, :nb, :nb], qp_offset=ref["qp_offset"])
            check_lookahead_outputs(outs, z, nb)
            ok = True
        except Exception as e:  # noqa: BLE001
            print("EXC", repr(e)[:400], "in", *desc)
            ok = False
        say("OK" if ok else "BAD", flush=True)
        bad += not ok
    return bad


if __name__ == "__main__":
    print("bad", run(sys.argv[1], sys.argv[2], big=len(sys.argv) > 3 and sys.argv[3] == "big"))
