"""Which fields (list, distance) and cost cells (d0, d1) the decision flow asks for, per frame, by the position the frame ends up at in its mini-GOP:\nthe host lookahead over the CPU oracle backend (test infrastructure) on a small synthetic clip.  Evidence for DESIGN.md section 3, "Speculation: what it\ncosts to speculate less" (profiles/r03_speculation_tradeoff.txt).   usage: python scripts/request_log.py"""
import sys, collections
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.oracle_backend import OracleBackend
from x264_amd import lib
from x264_amd.synth import make_clip
W, H, F = 352, 288, 120
frames = make_clip(W, H, F, seed=100, scene_cuts=(F // 3, F // 3 + 47), fade=(2 * F // 3, 10, 0.6, 12), pan=(5, 3))
cfg = lib.la_config(W, H, "slow", me="dia")
be = OracleBackend(cfg, speculative=True)
req_f = collections.defaultdict(set); req_c = collections.defaultdict(set)
slot_frame = {}
orig_put = be.put_array
cnt = [0]
def put(slot, img, cb=None, cr=None):
    slot_frame[slot] = cnt[0]; cnt[0] += 1
    return orig_put(slot, img, cb, cr)
be.put_array = put
orig_cost = be._cost
def cost(user, s0, s1, sb, d0, d1, do_search, w, wi, rv, out):
    f = slot_frame[sb]
    req_c[f].add((d0, d1))
    if d0 and do_search[0]: req_f[f].add((0, d0))
    if d1 and do_search[1]: req_f[f].add((1, d1))
    return orig_cost(user, s0, s1, sb, d0, d1, do_search, w, wi, rv, out)
be._cost = cost
be.struct.frame_cost = lib.FRAME_COST_FN(cost)
la = lib.Lookahead(cfg, backend=be.struct)
outs = la.run(frames)
la.close()
types = {o.frame: o.type for o in outs}
s = "".join("?IiPbB"[types[i]] for i in range(F))
print(s)
# position = distance from previous anchor (non-B)
pos = {}; last = 0
for i in range(F):
    if types[i] in (1, 2, 3): pos[i] = (i - last, "A"); last = i
    else: pos[i] = (i - last, "B")
agg_f = collections.defaultdict(collections.Counter); agg_c = collections.defaultdict(collections.Counter); n = collections.Counter()
for i in range(1, F):
    # length of the mini-GOP this frame is in
    j = i
    while types[j] not in (1, 2, 3): j += 1
    k = i - 1
    while types[k] not in (1, 2, 3): k -= 1
    key = (j - k, i - k)
    n[key] += 1
    for c in req_f[i]: agg_f[key][c] += 1
    for c in req_c[i]: agg_c[key][c] += 1
for key in sorted(n):
    print("miniGOP len %d pos %d: frames %d  fields %s | cells %s" % (key[0], key[1], n[key], dict(sorted(agg_f[key].items())), dict(sorted(agg_c[key].items()))))
print("avg fields requested per frame", sum(len(v) for v in req_f.values()) / F, "cells", sum(len(v) for v in req_c.values()) / F)
