"""The reference encoder's own control flow on the real library: oracle/_ref/libx264ref8hip.so (jpsdr/x264 with its accelerator seam
bound to libx264hip.so by x264_amd/csrc/slicetype_hip.c -- see tests/test_reference_seam.py) runs x264_encoder_encode twice, hook off and
hook on (--opencl), on the GPU: coded order, slice types, every i_cost_est / i_cost_est_aq cell, the CRCs of lowres_costs / lowres_mvs /
lowres_mv_costs / f_qp_offset / row sums / VBV plans of every coded frame, every frame's size and the CRC of the bitstream are identical.
The memo, first-trigger flags, x264_weights_analyse, scene cuts, slicetype_path, MB-tree, VBV and the main encode are the reference's."""
import os

import pytest

from oracle import refharness
from x264_amd.synth import make_clip
from tests.test_reference_seam import compare_runs

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not refharness.available(8, seam=True), reason="oracle/_ref/libx264ref8hip.so not built")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "x264_amd", "libx264hip.so")

CASES = [
    ("cif_medium", 352, 288, 64, "medium", "", dict(seed=5, scene_cuts=(21,), fade=(34, 10, 0.6, 8))),
    ("cif_slow_dia_b8", 352, 288, 64, "slow", "me=dia,bframes=8", dict(seed=9, scene_cuts=(40,), pan=(5, 3))),
    ("cif_trellis_vbv", 352, 288, 48, "slower", "vbv-maxrate=800,vbv-bufsize=600,bitrate=500,repeat-headers=0", dict(seed=2, scene_cuts=(17,), fade=(25, 8, 0.7, 5))),
    ("1080p_slow_dia", 1920, 1080, 64, "slow", "me=dia", dict(seed=7, scene_cuts=(29,), fade=(40, 10, 0.6, 8))),
    ("1080p_medium_b8", 1920, 1080, 60, "medium", "bframes=8", dict(seed=3, scene_cuts=(33,), pan=(6, 2))),
]


@pytest.mark.parametrize("name,w,h,nf,preset,opts,clip", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("speculate", [0, 1], ids=["on-demand", "prefetch"])
def test_reference_encoder_on_libx264hip_equals_c_path(name, w, h, nf, preset, opts, clip, speculate):
    if speculate and not name.startswith("cif"):
        pytest.skip("the speculative variant is covered at CIF")
    os.environ["X264HIP_SEAM_PREFETCH"] = str(speculate)
    try:
        frames = make_clip(w, h, nf, **clip)
        a, b = compare_runs(w, h, frames, preset, opts, LIB)
    finally:
        os.environ.pop("X264HIP_SEAM_PREFETCH", None)
    types = "".join("?IiPbB"[t] if 0 <= t < 6 else "?" for t in a["type"])
    print(name, types, "stream crc %08x" % a["stream_crc"], "whole encode: hook off %.2f s, on %.2f s" % (a["seconds"], b["seconds"]))
    assert "P" in types and ("b" in types or "B" in types)
