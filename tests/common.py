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
                      bframe_bias=rcfg["b_bias"], cost_mv=cost_mv)
