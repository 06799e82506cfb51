#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/pmc_passes.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
# One rocprofv3 --pmc pass per argument over a short bench run (64 frames, one step); prints the sums for me_rows_kernel.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
tag=$1; shift
i=0
for C in "$@"; do
  i=$((i+1))
  timeout 250 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/${tag}_$i -o p -- python bench.py --no-cpu-baseline --no-primitives --steps 1 --warmup 0 --frames 64 > gpurun_out/${tag}_$i.log 2>&1
  python scripts/pmc_kernel.py gpurun_out/${tag}_$i ${KERNEL:-me_rows}
done
