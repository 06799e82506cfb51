#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s20
mkdir -p gpurun_out/$tag
python scripts/dbg_hpel.py 2>&1 | grep -v "^  \|^x \|P2" | head -12 | tee gpurun_out/$tag/summary.txt
( timeout 600 python -m pytest tests -q -m gpu -k "hpel or frame_filter or table_fillers or me_full or full_search" ) > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/$tag/summary.txt
tail -5 gpurun_out/$tag/pytest.log | cut -c1-300 | tee -a gpurun_out/$tag/summary.txt
timeout 300 python scripts/prim_bench.py hpel 2>&1 | tail -1 | tee -a gpurun_out/$tag/summary.txt
X264HIP_HPEL_TILED=1 timeout 300 python scripts/prim_bench.py hpel 2>&1 | tail -1 | tee -a gpurun_out/$tag/summary.txt
