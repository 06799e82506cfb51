"""Quick GPU probe: device, a sanity check that parity checks bite, and first search timings."""
import sys, time
sys.path.insert(0, '.')  # run from the repo root: python tests/tools/gpu_probe.py
import numpy as np
from x264_amd import lib
from x264_amd.synth import make_clip
from oracle.oraclelib import Oracle

W, H, NF = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 12
frames = make_clip(W, H, NF, seed=5, pan=(17, -9), noise=9, texture=0.35)
o = Oracle(8)
cfg = o.make_cfg(120, 68, me_method=0, subpel_refine=4, me_range=16, mv_range=512, subme=8, mbcmp_satd=1)
ctx = lib.Context(W, H, bframes=3, me_method=0, subpel_refine=4, me_range=16, mv_range=512, subme=8, max_frames=NF + 2, cost_mv=o._cost_mv)
print("device:", ctx.device_name())
t0 = time.time()
for i in range(NF):
    ctx.frame_put(i, frames[i])
ctx.synchronize()
print("put %d frames: %.1f ms" % (NF, (time.time() - t0) * 1e3))
# single on-demand P evaluation
t0 = time.time()
out = ctx.frame_cost(0, 1, 1, 1, 0, (1, 0), None, True, False)
t1 = time.time()
ms, ns, nb = ctx.last_search_ms()
print("single P eval: wall %.2f ms, search kernel %.3f ms (%d searches, %d blocks)" % ((t1 - t0) * 1e3, ms, ns, nb), out.cost_est)
pl0, pl1 = o.lowres_init(cfg, frames[0]), o.lowres_init(cfg, frames[1])
t0 = time.time(); m, c = o.search_field(cfg, pl1, pl0); t1 = time.time()
gm, gc = ctx.mvs(1, 0, 0)
print("oracle search %.1f ms; mvs equal %s costs equal %s; nonzero mvs %d; mv range %s" % ((t1 - t0) * 1e3, np.array_equal(gm, m), np.array_equal(gc, c), int((m != 0).any(1).sum()), (m.min(), m.max())))
m2 = m.copy(); m2[5, 0] += 1
print("sanity (must be False):", np.array_equal(gm, m2))
# batched prefetch of everything
t0 = time.time()
ctx.prefetch(list(range(NF)), list(range(NF)))
ctx.synchronize()
t1 = time.time()
ms, ns, nb = ctx.last_search_ms()
print("prefetch: wall %.2f ms, kernel %.3f ms for %d searches (%d blocks) -> %.3f us/block-search, %.1f searches/ms" % ((t1 - t0) * 1e3, ms, ns, nb, ms * 1e3 / nb, ns / ms))
