"""The bench's encoder-paced single stream (device-resident, pinned, pageable pictures) and eight paced streams, again and again: do the
latency form's in-kernel waits ever time out?  usage: python scripts/paced_stress.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch
import bench
from x264_amd import lib, shard
W, H, F = 1920, 1080, 160
cfg = lib.la_config(W, H, "slow", bit_depth=8, me="dia", threads=1)
nb = cfg["bframes"] + 2
dev = [bench.make_clip_device(torch, W, H, F, 100 + i, 8, scene_cuts=(F // 3, F // 3 + 47), fade=(2 * F // 3, 10, 0.6, 12), still=(2 * F // 3 - 2, 16)) for i in range(8)]
pinned = dev[0].cpu().pin_memory()
pageable = dev[0].cpu()
w = bench.Workload(torch, lib, shard, cfg, 0, 0, 1, F, [dev[0]], False)
dt, o = w.timed(1, 1)
w.close()
sig = bench.outputs_signature(o[0], nb)
bad = 0
for r in range(rounds):
    for name, clips, S in (("device", [dev[0]], 1), ("pinned", [pinned], 1), ("pageable", [pageable], 1), ("8 streams", dev, 8)):
        w = bench.Workload(torch, lib, shard, cfg, 0, 0, S, F, clips, True)
        t0 = time.perf_counter()
        try:
            dt, o = w.timed(2, 1, paced=True)
            ok = bench.outputs_signature(o[0], nb) == sig
            bad += not ok
            print("round %d %-9s %.1f frames/s %s" % (r, name, S * F * 2 / dt, "equal" if ok else "DIFFERENT"), flush=True)
        except Exception as e:
            bad += 1
            print("round %d %-9s FAILED after %.2f s: %r" % (r, name, time.perf_counter() - t0, e), flush=True)
        finally:
            try:
                w.close()
            except Exception as e:
                print("close:", repr(e))
print("failures:", bad)
