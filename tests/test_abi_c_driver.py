"""A plain C program (tests/tools/abi_driver.c) that includes include/x264hip.h, dlopen()s x264_amd/libx264hip.so and runs the call
sequence of the reference's slicetype_frame_cost hook (frame_put x3 -> first-trigger P evaluation -> B
evaluation -> getters), the way a maintainer's patch to encoder/slicetype.c:878-897 would.  This file writes the inputs and the
arrays the oracle expects to a temporary directory; the C program compares every returned array with them.
CPU part: the driver compiles against the header as C (not C++) and every entry point it resolves exists in the library."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle.oraclelib import Oracle
from tests.common import clip

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "tools", "abi_driver.c")
BIN = os.path.join(HERE, "tools", "_build", "abi_driver")
LIB = os.path.join(ROOT, "x264_amd", "libx264hip.so")


def build_driver():
    if not os.path.exists(BIN) or max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, "include", "x264hip.h"))) > os.path.getmtime(BIN):
        os.makedirs(os.path.dirname(BIN), exist_ok=True)
        subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", BIN, SRC, "-ldl"])
    return BIN


def test_driver_compiles_as_c_and_finds_its_symbols():
    build_driver()
    import ctypes
    L = ctypes.CDLL(LIB)
    src = open(SRC).read()
    import re
    names = set(re.findall(r"RESOLVE\( (x264hip_\w+) \)", src))
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), n


@pytest.mark.gpu
@pytest.mark.parametrize("depth,me_method,subpel_refine,subme,mbcmp,fpelcmp", [(8, 1, 4, 7, 1, 0), (10, 0, 2, 1, 0, 0)])
def test_c_driver_hook_sequence(depth, me_method, subpel_refine, subme, mbcmp, fpelcmp):
    build_driver()
    W, H, mv_range, bframes = 352, 288, 128, 3
    o = Oracle(depth)
    frames = clip("fastpan", W, H, 3, depth)
    cfg = o.make_cfg((W + 15) // 16, (H + 15) // 16, me_method=me_method, subpel_refine=subpel_refine, me_range=16, mv_range=mv_range, subme=subme,
                     mbcmp_satd=mbcmp, fpelcmp_satd=fpelcmp)
    pl = [o.lowres_init(cfg, f) for f in frames]
    aq = [o.aq_frame(f, cfg.mb_w, cfg.mb_h, 1, 1.0) for f in frames]
    intra = [o.intra_costs(cfg, p) for p in pl]
    with tempfile.TemporaryDirectory() as d:
        def put(name, arr):
            np.ascontiguousarray(arr).tofile(os.path.join(d, name))
        open(os.path.join(d, "params.txt"), "w").write("%d %d %d %d %d %d %d %d %d %d %d %d\n" % (
            W, H, depth, bframes, cfg.lambda_, me_method, subpel_refine, 16, mv_range, subme, mbcmp, fpelcmp))
        put("cost_mv.bin", o._cost_mv)
        put("frames.bin", frames)
        put("expect_stats0.bin", np.array([aq[0][2], aq[0][3]], np.uint64))
        put("expect_invq0.bin", aq[0][0])
        # P evaluation (0, 2, 2)
        m20, c20 = o.search_field(cfg, pl[2], pl[0])
        lc, rows, rows_i, oo = o.cell(cfg, pl[2], pl[0], None, 128, m20, c20, None, None, None, intra[2], aq[2][0], True)
        put("expect_p_sums.bin", np.array([oo.cost_est, oo.cost_est_aq, oo.intra_mbs, oo.intra_cost_est, oo.intra_cost_est_aq], np.int32))
        put("expect_p_mvs.bin", m20); put("expect_p_mvcosts.bin", c20); put("expect_p_lc.bin", lc); put("expect_p_rows.bin", rows)
        put("expect_p_intra.bin", intra[2])
        # B evaluation (0, 2, 1) with the list-1 reference's own L0 vectors
        m10, c10 = o.search_field(cfg, pl[1], pl[0]); m11, c11 = o.search_field(cfg, pl[1], pl[2])
        lcb, rowsb, _, ob = o.cell(cfg, pl[1], pl[0], pl[2], 128, m10, c10, m11, c11, m20, intra[1], aq[1][0], True)
        put("expect_b_sums.bin", np.array([ob.cost_est, ob.cost_est_aq], np.int32))
        put("expect_b_lc.bin", lcb); put("expect_b_rows.bin", rowsb)
        r = subprocess.run([BIN, LIB, d], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        print(r.stdout.decode(errors="replace"))
        assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
