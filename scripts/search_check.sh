#!/bin/bash
# usage (through gpurun): bash scripts/search_check.sh <tag> [variant-lib-tags...]
# Parity tests of the search and the lookahead, then a short bench run (eight segments in flight, then one) with the library in the
# tree and with every variant library x264_amd/libx264hip_<variant>.so (python -m x264_amd.build --variant <tag> DEF=..).
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out/$tag
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lookahead.py -q -m gpu -x ) > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/$tag/summary.txt
tail -5 gpurun_out/$tag/pytest.log
S=("X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip.so")
for v in "$@"; do S+=("X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip_$v.so"); done
bash scripts/sweep_env.sh $tag "${S[@]}"
BENCH_ARGS="--inflight 1" bash scripts/sweep_env.sh $tag "${S[@]}"
