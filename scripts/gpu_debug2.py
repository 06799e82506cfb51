import sys
sys.path.insert(0, '.')
import numpy as np
from x264_amd import lib
from x264_amd.synth import make_clip
W, H = 352, 288
fr = make_clip(W, H, 6, seed=1)
ctx = lib.Context(W, H, mv_range=128, max_frames=8)
for i in range(6):
    ctx.frame_put(i, fr[i])
ctx.prefetch(list(range(6)), list(range(6)))
for s in (0, 1, 2):
    o = ctx.frame_cost(s, s, s, 0, 0, (0, 0), None, True, False)
    print("slot", s, "I-cell", o.intra_cost_est, o.intra_cost_est_aq, "counters", ctx.counters()[:6])
o = ctx.frame_cost(0, 1, 1, 1, 0, (1, 0), None, False, False)
print("P", o.cost_est, o.cost_est_aq, o.intra_mbs, ctx.counters()[:6])
o = ctx.frame_cost(1, 3, 2, 1, 1, (1, 1), None, True, False)
print("B novalid", o.cost_est, ctx.counters()[:6])
