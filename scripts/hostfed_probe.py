"""Eight (or argv[1]) host-fed segments of the headline workload, pictures in pinned host memory, through x264hip_lookahead_put_frames:
frames/s and the PCIe rate.  usage: python scripts/hostfed_probe.py [segments] [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 5
os.environ.setdefault("GPU_MAX_HW_QUEUES", str(2 * S + 4))
import torch
import bench
from x264_amd import lib, shard
W, H, F = 1920, 1080, 160
cfg = lib.la_config(W, H, "slow", bit_depth=8, me="dia", threads=1)
dev = [bench.make_clip_device(torch, W, H, F, 100 + i, 8, scene_cuts=(F // 3, F // 3 + 47), fade=(2 * F // 3, 10, 0.6, 12), still=(2 * F // 3 - 2, 16)) for i in range(S)]
host = [d.cpu().pin_memory() for d in dev]
for tag, clips in (("device-resident", dev), ("host-fed", host)):
    wl = bench.Workload(torch, lib, shard, cfg, 0, 0, S, F, clips, False)
    try:
        dt, o = wl.timed(passes, 1)
    finally:
        wl.close()
    fps = S * F * passes / dt
    print("%s: %d segments x %d passes: %.1f frames/s, %.1f ms per step, %.1f GB/s of pictures" % (tag, S, passes, fps, dt / passes * 1e3, fps * W * H / 1e9), flush=True)
