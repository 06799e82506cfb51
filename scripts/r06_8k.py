"""BASELINE configs[4] on one GPU as bench.py runs it (two 72-frame segments of 7680x4320 10-bit, veryslow + tesa), alone: batched passes, then paced."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "10")
import torch
import bench
from x264_amd import lib, shard
cfg = lib.la_config(7680, 4320, "veryslow", bit_depth=10, me="tesa")
S, F = 2, 72
dev = [bench.make_clip_device(torch, 7680, 4320, F, 300 + i, 10, scene_cuts=(47,)) for i in range(S)]
wl = bench.Workload(torch, lib, shard, cfg, 0, 0, S, F, dev, False)
try:
    for tag, paced in (("batched", False), ("paced", True), ("batched", False)):
        t0 = time.perf_counter()
        try:
            dt, o = wl.timed(2, 1, paced=paced)
            print(tag, "ok %.1f frames/s" % (S * F * 2 / dt), flush=True)
        except Exception as e:
            print(tag, "FAILED after %.1f s:" % (time.perf_counter() - t0), e, flush=True)
            break
finally:
    try:
        wl.close()
    except Exception as e:
        print("close:", e)
