#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s14
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d gpurun_out/${tag} -o run -- python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --paced --inflight 1 --warmup 1 --steps 2 > gpurun_out/${tag}/bench.log 2>&1; echo "rc=$?" | tee gpurun_out/$tag/summary.txt
grep -h '^{' gpurun_out/$tag/bench.log | cut -c1-300 | tee -a gpurun_out/$tag/summary.txt
B="python bench.py --no-cpu-baseline --no-primitives --no-extra --warmup 2 --steps 4"
for A in "" "--inflight 1" "--inflight 12"; do
timeout 400 $B $A 2>&1 | grep '^{' | cut -c1-120 | tee -a gpurun_out/$tag/summary.txt
done
