"""The bench's host_fed sequence again and again (device-resident contexts stay open beside the host-fed ones, S then S/2 segments in
flight), to catch what happens once in a while.  usage: python scripts/hostfed_stress.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
S = 8
os.environ.setdefault("GPU_MAX_HW_QUEUES", str(2 * S + 4))
import torch
import bench
from x264_amd import lib, shard
W, H, F = 1920, 1080, 160
cfg = lib.la_config(W, H, "slow", bit_depth=8, me="dia", threads=1)
dev = [bench.make_clip_device(torch, W, H, F, 100 + i, 8, scene_cuts=(F // 3, F // 3 + 47), fade=(2 * F // 3, 10, 0.6, 12), still=(2 * F // 3 - 2, 16)) for i in range(S)]
host = [d.cpu().pin_memory() for d in dev]
wl = bench.Workload(torch, lib, shard, cfg, 0, 0, S, F, dev, False)
dt, outs = wl.timed(3, 1)
nb = cfg["bframes"] + 2
sig = [bench.outputs_signature(o, nb) for o in outs]
print("device-resident %.1f frames/s" % (S * F * 3 / dt), flush=True)
bad = 0
for r in range(rounds):
    for Sh in (S, S // 2):
        w = bench.Workload(torch, lib, shard, cfg, 0, 0, Sh, F, host[:Sh], False)
        try:
            t0 = time.perf_counter()
            dth, oh = w.timed(4, 1)
            ok = all(bench.outputs_signature(oh[i], nb) == sig[i] for i in range(Sh))
            print("round %d, %d in flight: %.1f frames/s, %s" % (r, Sh, Sh * F * 4 / dth, "equal" if ok else "DIFFERENT"), flush=True)
            bad += not ok
        except Exception as e:
            bad += 1
            print("round %d, %d in flight: FAILED after %.2f s: %r" % (r, Sh, time.perf_counter() - t0, e), flush=True)
        finally:
            try:
                w.close()
            except Exception as e:
                print("close:", repr(e))
wl.close()
print("failures:", bad)
