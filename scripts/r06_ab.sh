#!/bin/bash
# usage (through gpurun): bash scripts/r06_ab.sh <out tag> <variant tag>...: parity tests on the default build, then short bench runs of the variants
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; shift
mkdir -p $out
if [ -z "$NO_PARITY" ]; then
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lookahead.py tests/test_gpu_configs.py -q -m gpu -x ) > $out/parity.log 2>&1; echo "parity rc=$?" | tee -a $out/summary.txt
tail -3 $out/parity.log
fi
bash scripts/ab_variants.sh "$@" 2>&1 | tee -a $out/summary.txt
for B in "--inflight 1" "--inflight 1 --paced"; do
  X264HIP_SEARCH=rows X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip_prof.so timeout 300 python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check $B > $out/prof.log 2>&1
  echo "== prof rows $B" | tee -a $out/summary.txt; grep -h "ME_PROFILE" $out/prof.log | tail -4 | tee -a $out/summary.txt
done
X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip_prof.so timeout 300 python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --inflight 1 --paced > $out/prof.log 2>&1
echo "== prof default dispatch --inflight 1 --paced" | tee -a $out/summary.txt; grep -h "ME_PROFILE" $out/prof.log | tail -4 | tee -a $out/summary.txt
