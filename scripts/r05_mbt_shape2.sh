#!/bin/bash
# round 5, GPU call: MB-tree workgroup size, eight contexts and one; 40 timed steps per run, two runs each, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05mbt; mkdir -p $out
short="--no-cpu-baseline --no-primitives --no-extra --no-check --steps 40 --warmup 4"
for rep in 1 2; do
for E in "A=0" "X264HIP_MBT_THREADS=512" "X264HIP_MBT_THREADS=384" "X264HIP_MBT_THREADS=256"; do
  for B in "--inflight 8" "--inflight 1"; do
  env $E timeout 300 python bench.py $short $B > $out/t.log 2>&1
  echo "$E $B: $(grep -h '^{' $out/t.log | tail -1 | cut -c36-60)"
  done
done
done
