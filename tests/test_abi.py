"""The C-ABI library loads without a GPU and exports every entry point include/x264hip.h declares."""
import ctypes as C
import os
import re

import numpy as np

from x264_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "x264hip.h")).read()
    names = sorted(set(re.findall(r"\b(x264hip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    L = lib.load()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_struct_sizes_match_header():
    # x264hip_params: 14 ints + float + 5 ints + pointer; la_frame: 4 ints + two 18x18 matrices + 18 ints
    assert C.sizeof(lib.Params) == 20 * 4 + 8
    assert C.sizeof(lib.LaFrameOut) == (4 + 2 * 18 * 18 + 18) * 4
    assert C.sizeof(lib.Cost) == 20 and C.sizeof(lib.Weight) == 16


def test_invalid_arguments_are_rejected_without_a_device():
    L = lib.load()
    h = C.c_void_p()
    assert L.x264hip_open(C.byref(h), 0, None) == -2
    tab, centre = lib.cost_mv_table(128, 1)
    p = lib.Params(9, 352, 288, 3, 1, 1, 4, 16, 128, 7, 1, 0, 1, 1, 1.0, 0, 8, 0, 1, 1, tab.ctypes.data + 2 * centre)
    assert L.x264hip_open(C.byref(h), 0, C.byref(p)) == -2  # bit depth 9
    assert L.x264hip_strerror(-1).decode().startswith("no usable HIP device")


def test_open_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    try:
        lib.Context(352, 288)
    except lib.X264HipError as e:
        assert e.code == -1
    else:
        raise AssertionError("x264hip_open must not succeed without a device (no CPU fallback)")


def test_la_config_mv_range():
    assert lib.mv_range_for(176, 144) == 128 and lib.mv_range_for(1920, 1080) == 512 and lib.mv_range_for(3840, 2160) == 512


def test_struct_layouts_match_header(tmp_path):
    """Every field of every struct that crosses the C ABI: offset and total size as a C compiler lays out include/x264hip.h
    against the ctypes mirrors in x264_amd/lib.py (the header is the contract; a silent mismatch would shift every field after it)."""
    import subprocess
    pairs = {"x264hip_params": lib.Params, "x264hip_weight": lib.Weight, "x264hip_cost": lib.Cost, "x264hip_mbtree_op": lib.MbtreeOp,
             "x264hip_la_params": lib.LaParams, "x264hip_backend": lib.Backend, "x264hip_la_frame": lib.LaFrameOut,
             "x264hip_la_vbv": lib.LaVbv, "x264hip_me_request": lib.MeRequest, "x264hip_picture": lib.Picture}
    rename = {"lambda_": "lambda"}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "x264hip.h"', 'int main(void){']
    for cname, cls in pairs.items():
        lines.append('printf("%s.sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, rename.get(fname, fname)))
    lines.append('return 0;}')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    want = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in pairs.items():
        assert C.sizeof(cls) == int(want[cname + ".sizeof"]), cname
        for fname, _ in cls._fields_:
            assert getattr(cls, fname).offset == int(want["%s.%s" % (cname, fname)]), (cname, fname)


def test_new_entries_reject_null_arguments_without_a_device():
    """every entry added for the next rows validates its arguments before touching the device (X264HIP_EINVAL = -2)"""
    L = lib.load()
    for name, nargs in (("x264hip_me_search_batch", 11), ("x264hip_pixel_metric_batch", 9), ("x264hip_frame_dct_quant8x8", 11),
                        ("x264hip_integral_init", 7), ("x264hip_frame_filter", 11), ("x264hip_frame_put_batch_yuv", 8),
                        ("x264hip_lookahead_put_picture", 6), ("x264hip_lookahead_put_pictures", 9), ("x264hip_lookahead_get_frame_vbv", 8),
                        ("x264hip_lookahead_put_frame_pts", 6)):
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = None
        assert fn(*([None] * nargs)) == -2, name
