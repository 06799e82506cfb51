#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/trace -- python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --steps 3 --warmup 1 --paced --inflight 1 > $out/trace.log 2>&1
python scripts/trace_kernels.py $out/trace 40 20 2>&1 | tee $out/timeline.txt
rm -rf $out/trace
