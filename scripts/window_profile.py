#!/usr/bin/env python
"""One stream alone (BASELINE configs[3] by default): where a pass goes.  Prints wall time per pass, the lookahead's own host
accounting (x264hip_lookahead_stats) and the device counters, for the plain batched pass, the hooked (window shard, world = 1) pass
and the encoder-paced pass.  Run under rocprofv3 --kernel-trace --stats / --hip-runtime-trace for the device and API side.
usage: python scripts/window_profile.py [--config 3|1|2] [--frames N] [--modes plain,shard,paced] [--passes K]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--modes", default="plain,shard,paced")
    ap.add_argument("--passes", type=int, default=2)
    a = ap.parse_args()
    import torch
    from x264_amd import lib, shard
    import bench
    if a.config == 3:
        W, H, F = 3840, 2160, a.frames or 250
        cfg = lib.la_config(W, H, "medium", bit_depth=8, bframes=8, rc_lookahead=60, keyint_max=250)
    elif a.config == 2:
        W, H, F = 3840, 2160, a.frames or 64
        cfg = lib.la_config(W, H, "slower", bit_depth=8, me="umh", me_range=32)
    else:
        W, H, F = 1920, 1080, a.frames or 160
        cfg = lib.la_config(W, H, "slow", bit_depth=8, me="dia")
    clip = bench.make_clip_device(torch, W, H, F, 4242, 8, scene_cuts=(F // 3,))
    ptrs = [clip[i].data_ptr() for i in range(F)]
    for mode in a.modes.split(","):
        for k in range(a.passes + 1):
            if mode == "shard":
                torch.cuda.synchronize()
                outs, dt, st = shard.run_window_shard(torch, lib, None, 0, 1, 0, cfg, clip, True)
                host = None
            else:
                la = lib.Lookahead(cfg, device=0, max_frames=F + 4 if mode == "plain" else 0)
                try:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    outs = la.run(device_ptrs=ptrs, stride=W, paced=(mode == "paced"))
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    hs = la.stats()
                    host = {"frame_cost_calls": int(hs[0]), "evaluations": int(hs[1]), "ms_frame_cost": round(hs[4] / 1e6, 2), "ms_weights": round(hs[5] / 1e6, 2),
                            "ms_prefetch_mbtree": round(hs[6] / 1e6, 2), "ms_api_total": round(hs[7] / 1e6, 2)}
                finally:
                    la.close()
            if k:
                print(json.dumps({"mode": mode, "pass": k, "frames": F, "seconds": round(dt, 4), "fps": round(F / dt, 1), "host": host,
                                  "types": "".join("?IiPbB"[o.type] for o in outs)[:80]}), flush=True)


if __name__ == "__main__":
    main()
