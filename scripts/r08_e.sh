#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_lookahead.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3 | tee $out/tests.txt
bash scripts/r07_ab_libs.sh $1 2 libx264hip_prev.so libx264hip.so
# a clip that stands still half of the time: where the skipped candidates are
python - <<'PY' 2>&1 | grep -v Warning | tee -a $out/ab.txt
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["GPU_MAX_HW_QUEUES"] = "20"
import torch, bench
from x264_amd import lib, shard
W, H, F, S = 1920, 1080, 160, 8
for name in ("libx264hip_prev.so", "libx264hip.so"):
    pass
PY
