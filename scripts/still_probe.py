"""Eight segments of a clip that does not move (noise only): where cell_b_kernel's skipped candidates are.  usage: X264HIP_LIB=<so> python scripts/still_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch
import bench
from x264_amd import lib, shard
W, H, F, S = 1920, 1080, 160, 8
cfg = lib.la_config(W, H, "slow", bit_depth=8, me="dia", threads=1)
for pan in ((0, 0), (5, 3)):
    dev = [bench.make_clip_device(torch, W, H, F, 100 + i, 8, scene_cuts=(F // 3,), pan=pan) for i in range(S)]
    wl = bench.Workload(torch, lib, shard, cfg, 0, 0, S, F, dev, False)
    try:
        dt, o = wl.timed(6, 2)
        la = wl.las[0]
        lib.search_profile(la.L, la.ctx_handle(), 3)
        la.reset(); la.run(device_ptrs=wl.seg_ptrs[0], stride=W, paced=False)
        kt = lib.kernel_profile(la.L, la.ctx_handle())
        lib.search_profile(la.L, la.ctx_handle(), 0)
    finally:
        wl.close()
    print("pan %s: %.1f frames/s; kernel ms of one segment alone (lowres, aq, intra, cell_p, cell_b, cell_reduce): %s" % (pan, S * F * 6 / dt, [round(k[0], 3) for k in kt]), flush=True)
