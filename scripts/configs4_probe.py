"""BASELINE configs[4] as the bench runs it (two contexts, 72 frames of 7680x4320 10-bit, veryslow + tesa), alone: batched passes, then a paced
one.  usage: python scripts/configs4_probe.py [segments=2] [passes=2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch
import bench
from x264_amd import lib, shard
cfg = lib.la_config(7680, 4320, "veryslow", bit_depth=10, me="tesa")
F = 72
dev = [bench.make_clip_device(torch, 7680, 4320, F, 300 + i, 10, scene_cuts=(47,)) for i in range(S)]
w = bench.Workload(torch, lib, shard, cfg, 0, 0, S, F, dev, False)
try:
    for mode in (False, True, False):
        t0 = time.perf_counter()
        try:
            dt, o = w.timed(passes, 1, paced=mode)
            print("paced" if mode else "batched", "%.1f frames/s" % (S * F * passes / dt), flush=True)
        except Exception as e:
            print("paced" if mode else "batched", "FAILED after %.1f s: %r" % (time.perf_counter() - t0, e), flush=True)
            break
finally:
    w.close()
