#!/bin/bash
# round 5, GPU call: the queued MB-tree lists on one workgroup each with LDS accumulators (X264HIP_MBT=lds) against the default, the
# multi-workgroup form with global atomics
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05mbt; mkdir -p $out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lookahead.py tests/test_gpu_fuzz.py -q -m gpu -x ) > $out/parity.log 2>&1; echo "parity rc=$?"; tail -3 $out/parity.log
short="--no-cpu-baseline --no-primitives --no-extra --no-check"
for rep in 1 2; do
for E in "X264HIP_MBT=lds" "A=0"; do
  for B in "--inflight 8" "--inflight 1"; do
    env $E timeout 300 python bench.py $short $B > $out/t.log 2>&1
    echo "$E $B: $(grep -h '^{' $out/t.log | tail -1 | cut -c1-90)"
  done
done
done
