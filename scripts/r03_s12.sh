#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s12
mkdir -p gpurun_out/$tag
nproc | tee gpurun_out/$tag/summary.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
X264HIP_MBT_GROUPS=16 X264HIP_MBT_WGS=2 timeout 600 rocprofv3 --hip-runtime-trace --stats --output-format csv -d gpurun_out/${tag} -o run -- python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check > gpurun_out/${tag}/bench.log 2>&1; echo "stats rc=$?" | tee -a gpurun_out/$tag/summary.txt
ls -la gpurun_out/$tag | tee -a gpurun_out/$tag/summary.txt
head -40 gpurun_out/$tag/run_hip_api_stats.csv | tee -a gpurun_out/$tag/summary.txt
rm -f gpurun_out/$tag/run_hip_api_trace.csv.big
