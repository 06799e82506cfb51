#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s19
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag} -o run -- python scripts/prim_bench.py > gpurun_out/${tag}/bench.log 2>&1; echo "rc=$?" | tee gpurun_out/$tag/summary.txt
tail -1 gpurun_out/$tag/bench.log | tee -a gpurun_out/$tag/summary.txt
grep -i "hpel\|pixel_cmp\|copy16\|frame_dct\|me_full" gpurun_out/$tag/run_kernel_stats.csv | cut -c1-60,120-260 | tee -a gpurun_out/$tag/summary.txt
rm -f gpurun_out/$tag/run_kernel_trace.csv
