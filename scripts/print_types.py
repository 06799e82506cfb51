import sys, torch
sys.path.insert(0, '.')
import bench
from x264_amd import lib
W,H,F=1920,1080,160
cfg = lib.la_config(W,H,"slow",me="dia")
dv = bench.make_clip_device(torch, W, H, F, 100, 8, scene_cuts=(F // 3, F // 3 + 47), fade=(2 * F // 3, 10, 0.6, 12))
la = lib.Lookahead(cfg, max_frames=F+4)
outs = la.run(device_ptrs=[dv[i].data_ptr() for i in range(F)], stride=W, paced=False)
t = {o.frame:o.type for o in outs}
print("".join("?IiPbB"[t[i]] for i in range(F)))
la.close()
