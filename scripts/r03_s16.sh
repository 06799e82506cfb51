#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s16
mkdir -p gpurun_out/$tag
: > gpurun_out/$tag/summary.txt
B="python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --warmup 1 --steps 2"
for A in "--paced --inflight 1" "--paced" "--inflight 1" ""; do
    echo "== $A" | tee -a gpurun_out/$tag/summary.txt
    X264HIP_LIB=x264_amd/libx264hip_prof.so timeout 400 $B $A > gpurun_out/$tag/ab.log 2>&1
    grep -h "ME_PROFILE" gpurun_out/$tag/ab.log | sort | uniq -c | sort -rn | head -4 | cut -c1-330 | tee -a gpurun_out/$tag/summary.txt
    grep -h '^{' gpurun_out/$tag/ab.log | cut -c1-100 | tee -a gpurun_out/$tag/summary.txt
done
