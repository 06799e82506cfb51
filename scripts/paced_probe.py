"""One encoder-paced 1080p stream (BASELINE configs[1]) on the library X264HIP_LIB names: frames/s and the search launches' own time.
usage: python scripts/paced_probe.py [passes] [WxH]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch
from x264_amd import lib
import bench

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x"))
F = 160
cfg = lib.la_config(W, H, "slow", bit_depth=8, me="dia", threads=1)
clip = bench.make_clip_device(torch, W, H, F, 100, 8, scene_cuts=(F // 3, F // 3 + 47), fade=(2 * F // 3, 10, 0.6, 12), still=(2 * F // 3 - 2, 16))
ptrs = [clip[i].data_ptr() for i in range(F)]
la = lib.Lookahead(cfg, device=0, max_frames=F + 4)
la.reset(); la.run_frames(ptrs, stride=W, paced=True)
torch.cuda.synchronize()
lib.search_profile(la.L, la.ctx_handle(), 1)
t0 = time.perf_counter()
for _ in range(passes):
    la.reset()
    outs = la.run_frames(ptrs, stride=W, paced=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
lat = lib.search_profile_latency(la.L, la.ctx_handle())
ms, nl, ns = lib.search_profile(la.L, la.ctx_handle(), 0)
print("%s: paced %.1f frames/s | small launches: %d, %.1f searches each, %.1f us each | big launches %d, %.3f ms" % (
    os.path.basename(os.environ.get("X264HIP_LIB", "default")), passes * F / dt, lat[1], lat[2] / max(lat[1], 1), 1e3 * lat[0] / max(lat[1], 1), nl, ms))
la.close()
