import os, sys
sys.path.insert(0, '.')
import numpy as np
from tests.golden.make_golden import LOOKAHEAD_CASES
from tests.test_golden import GOLD
from x264_amd import lib
from x264_amd.synth import make_clip
name = "medium_cif"
preset, opts, over, depth, W, H, ckw, nf = LOOKAHEAD_CASES[name]
frames = make_clip(W, H, nf, bit_depth=depth, **ckw)
cfg = lib.la_config(W, H, preset, bit_depth=depth, **over)
z = np.load(os.path.join(GOLD, "lookahead_%s.npz" % name))
la = lib.Lookahead(cfg)
outs = la.run(frames, paced=True, qp_offsets=True)
la.close()
for k, o in enumerate(outs[:16]):
    d = np.abs(o.qp_offset - z["qp_offset"][k])
    print(o.frame, o.type, "mismatch", int((d > 0).sum()), "max", d.max(), "big(>1e-4)", int((d > 1e-4).sum()))
