#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s15
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --hip-runtime-trace --stats --output-format csv -d gpurun_out/${tag} -o run -- python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --paced --warmup 1 --steps 2 > gpurun_out/${tag}/bench.log 2>&1; echo "rc=$?" | tee gpurun_out/$tag/summary.txt
grep -h '^{' gpurun_out/$tag/bench.log | cut -c1-300 | tee -a gpurun_out/$tag/summary.txt
head -14 gpurun_out/$tag/run_hip_api_stats.csv | cut -c1-150 | tee -a gpurun_out/$tag/summary.txt
