"""GPU end-to-end for the configurations whose device paths were written after the last GPU session of round 1: edge ring
not evaluated (no MB-tree: --no-mbtree, --qp, superfast / ultrafast, qcomp 1), lookahead bands (lookahead_threads > 1),
auto-variance AQ (aq-mode 2 / 3), constant QP, VBV lookahead.  Same golden fixtures and checks as tests/test_gpu_lookahead.py; the file
name sorts last on purpose, so that with `pytest -x` a failure here cannot hide the results of the verified files."""
import os

import numpy as np
import pytest

from tests.golden.make_golden import LOOKAHEAD_CASES_R2
from tests.test_golden import GOLD, check_lookahead_outputs
from x264_amd import lib
from x264_amd.synth import make_clip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(LOOKAHEAD_CASES_R2))
@pytest.mark.parametrize("paced", [True, False])
def test_lookahead_vs_golden_new_configs(name, paced):
    preset, opts, over, depth, W, H, ckw, nf = LOOKAHEAD_CASES_R2[name]
    z = np.load(os.path.join(GOLD, "lookahead_%s.npz" % name))
    frames = make_clip(W, H, nf, bit_depth=depth, **ckw)
    cfg = lib.la_config(W, H, preset, bit_depth=depth, **over)
    la = lib.Lookahead(cfg, max_frames=0 if paced else nf + 4)
    try:
        outs = la.run(frames, paced=paced, qp_offsets=True, vbv=bool(cfg["vbv"]))
    finally:
        la.close()
    check_lookahead_outputs(outs, z, cfg["bframes"] + 2, check_qp=bool(cfg["aq_mode"]))


def test_aq_modes_against_oracle():
    """x264hip_frame_put with aq-mode 2 / 3: the Q8 inverse qscale map bit for bit against the oracle (sequential FP32 sums,
    correctly rounded roots and divisions, ratecontrol.c:354-398 in the reference build's operation order)."""
    from oracle.oraclelib import Oracle
    for depth in (8, 10):
        o = Oracle(depth)
        W, H = 352, 288
        fr = make_clip(W, H, 2, seed=7, bit_depth=depth, noise=20, texture=0.8)
        for mode, strength in ((2, 1.0), (3, 0.7), (2, 2.5)):
            ctx = lib.Context(W, H, bit_depth=depth, aq_mode=mode, aq_strength=strength, max_frames=8)
            try:
                ctx.frame_put(0, fr[0])
                inv = ctx.inv_qscale(0)
                qp = ctx.qp_offsets(0)
            finally:
                ctx.close()
            want, want_qp = o.aq_frame(fr[0], (W + 15) // 16, (H + 15) // 16, mode, strength)[:2]
            assert np.array_equal(qp, want_qp), (depth, mode, strength, float(np.abs(qp - want_qp).max()))
            assert np.array_equal(inv, want), (depth, mode, strength, int(np.abs(inv.astype(int) - want.astype(int)).max()))
