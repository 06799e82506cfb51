#!/bin/bash
# First GPU session of the next round (run on the GPU box from the repo root through gpurun, ~12 GPU-minutes):
#   1. the GPU tests of the device paths written after the last GPU session (unvisited ring, lookahead bands, AQ 2/3, VBV end to end)
#   2. the rest of the GPU suite (regression check of the main path after the same edits)
#   3. the default bench line, and the same configuration with x264's automatic lookahead bands (--threads 24 -> 4 bands at 1080p)
# Everything is written under gpurun_out/next/ ; nothing here uses rocprofv3 (scripts/profile_round.sh does).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/next
timeout 900 python -m pytest tests/zz_gpu_new_configs_impl.py -q -m gpu > gpurun_out/next/new_configs.log 2>&1; echo "new configs rc=$?" | tee -a gpurun_out/next/summary.txt
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_zz_gpu_new_configs.py > gpurun_out/next/suite.log 2>&1; echo "suite rc=$?" | tee -a gpurun_out/next/summary.txt
timeout 400 python bench.py > gpurun_out/next/bench_default.log 2>&1; grep -h '^{' gpurun_out/next/bench_default.log | tail -1 | cut -c1-160 | tee -a gpurun_out/next/summary.txt
timeout 400 python bench.py --threads 24 --no-primitives > gpurun_out/next/bench_bands.log 2>&1; grep -h '^{' gpurun_out/next/bench_bands.log | tail -1 | cut -c1-160 | tee -a gpurun_out/next/summary.txt
tail -3 gpurun_out/next/new_configs.log gpurun_out/next/suite.log
