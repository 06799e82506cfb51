"""The reference's OWN lookahead control flow on the library's level-1 entry points, unpatched: oracle/_ref/libx264ref8hip.so is
jpsdr/x264 compiled with its accelerator seam on (HAVE_OPENCL) and x264_amd/csrc/slicetype_hip.c standing in for
encoder/slicetype-cl.c + common/opencl.c, so that slicetype_frame_cost's hook (encoder/slicetype.c:878-897) calls `x264hip_frame_put /
_frame_cost / getters` of whatever $X264HIP_LIB names while the memo, first-trigger flags, weights, scene cuts, slicetype_path, MB-tree,
VBV and the main encode stay the reference's code.

Here (no GPU) $X264HIP_LIB is tests/tools/x264hip_oracle_shim.c -- the same entry points over the CPU oracle -- which checks the BINDING
(what it writes into the reference's arrays and when) and pins the oracle at the level of a whole x264_encoder_encode run: coded order,
slice types, every i_cost_est / i_cost_est_aq cell, CRCs of lowres_costs / lowres_mvs / lowres_mv_costs / f_qp_offset of every coded
frame, every frame's size and the CRC of the bitstream must equal the plain C run's.  tests/test_gpu_reference_seam.py runs the same
comparison with the real libx264hip.so."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oraclelib, refharness
from x264_amd.synth import make_clip

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

pytestmark = pytest.mark.skipif(not refharness.available(8, seam=True), reason="oracle/_ref/libx264ref8hip.so not built")


def shim_path():
    oraclelib.build()
    out = os.path.join(HERE, "tools", "_build", "libx264hip_oracle_shim.so")
    src = os.path.join(HERE, "tools", "x264hip_oracle_shim.c")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    deps = [src, os.path.join(ROOT, "oracle", "liboracle.so"), os.path.join(ROOT, "include", "x264hip.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-shared", "-fPIC", "-fvisibility=hidden", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "oracle"), "-o", out, src, "-L" + os.path.join(ROOT, "oracle"), "-loracle",
                               "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return out


class _library:
    """$X264HIP_LIB names the library the seam opens -- for the runs inside the block only: whatever the variable held before is what the
    tests that follow (and the processes they start) see again"""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        self.before = os.environ.get("X264HIP_LIB")
        os.environ["X264HIP_LIB"] = self.path

    def __exit__(self, *exc):
        if self.before is None:
            os.environ.pop("X264HIP_LIB", None)
        else:
            os.environ["X264HIP_LIB"] = self.before


def compare_runs(width, height, frames, preset, opts, lib, chroma=None):
    """the same encode twice in the seam build: hook off / hook on"""
    sep = "," if opts else ""
    outs = []
    with _library(lib):
        for accel in (0, 1):
            r = refharness.Ref(width, height, preset, opts=opts + sep + "opencl=%d" % accel, seam=True)
            try:
                assert r.accel_state() == accel, "the encoder did not keep the accelerator hook (library or device missing?)"
                outs.append(r.encode_run(frames, chroma))
                assert r.accel_state() == accel
            finally:
                r.close()
    a, b = outs
    assert a["frame"].tolist() == b["frame"].tolist()
    assert a["type"].tolist() == b["type"].tolist()
    for k in range(len(a["frame"])):
        where = "coded frame %d (display %d, type %d)" % (k, a["frame"][k], a["type"][k])
        assert np.array_equal(a["cost"][k], b["cost"][k]), (where, "i_cost_est", np.argwhere(a["cost"][k] != b["cost"][k])[:4].tolist())
        assert np.array_equal(a["cost_aq"][k], b["cost_aq"][k]), (where, "i_cost_est_aq")
        assert a["map_crc"][k].tolist() == b["map_crc"][k].tolist(), (where, "lowres_costs / lowres_mvs / lowres_mv_costs / f_qp_offset CRCs", a["map_crc"][k], b["map_crc"][k])
        assert a["bytes"][k] == b["bytes"][k], (where, "coded size")
    assert a["stream_crc"] == b["stream_crc"]
    return a, b


CASES = [
    ("medium_cif", 352, 288, 64, "medium", "", dict(seed=5, scene_cuts=(21,), fade=(34, 10, 0.6, 8))),
    ("slow_dia_b8", 352, 288, 64, "slow", "me=dia,bframes=8", dict(seed=9, scene_cuts=(40,), pan=(5, 3))),
    # (rate control that feeds coded sizes back -- ABR, VBV -- counts the version SEI of frame 0, whose option list is nine bytes longer
    # with " opencl=1" (common/base.c:1446-1447): repeat-headers=0 takes that SEI out, encoder.c:3731)
    ("trellis_vbv", 352, 288, 48, "slower", "vbv-maxrate=800,vbv-bufsize=600,bitrate=500,repeat-headers=0", dict(seed=2, scene_cuts=(17,), fade=(25, 8, 0.7, 5))),
    ("fast_nombtree", 320, 240, 40, "veryfast", "no-mbtree=1,rc-lookahead=20", dict(seed=4, scene_cuts=(11,))),
]


@pytest.mark.parametrize("name,w,h,nf,preset,opts,clip", CASES, ids=[c[0] for c in CASES])
def test_reference_lookahead_on_level1_entries_equals_c_path(name, w, h, nf, preset, opts, clip):
    frames = make_clip(w, h, nf, **clip)
    a, b = compare_runs(w, h, frames, preset, opts, shim_path())
    types = "".join("?IiPbB"[t] if 0 <= t < 6 else "?" for t in a["type"])
    print(name, types, "stream crc %08x" % a["stream_crc"], "hook off %.2fs on %.2fs" % (a["seconds"], b["seconds"]))
    assert "P" in types and ("b" in types or "B" in types)


def test_seam_falls_back_without_the_library():
    """no library to open: x264_opencl_load_library returns NULL and the encoder keeps its C path (encoder.c:1748-1752)"""
    with _library("/nonexistent/libx264hip.so"):
        r = refharness.Ref(352, 288, "medium", opts="opencl=1", seam=True)
        try:
            assert r.accel_state() == 0
        finally:
            r.close()
