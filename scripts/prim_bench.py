"""usage (GPU box): python scripts/prim_bench.py [keys...]   -- the `primitives` block of bench.py alone (SAD / SATD / hpel / copy GB/s ...),
one JSON line; environment switches of the library (X264HIP_COPY, X264HIP_CMP_ROWS) select kernel forms for A/B runs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from x264_amd import lib  # noqa: E402

cfg = lib.la_config(3840, 2160, "slow", bit_depth=8, me="dia")
out = bench.primitives_bench(torch, lib, cfg)
keys = sys.argv[1:]
print(json.dumps({k: v for k, v in out.items() if not keys or any(s in k for s in keys)}))
