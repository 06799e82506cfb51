#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s18
mkdir -p gpurun_out/$tag
( timeout 600 python -m pytest tests -q -m gpu -x -k "me_full or me_search or full_search or primitives" ) > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/$tag/summary.txt
tail -3 gpurun_out/$tag/pytest.log | tee -a gpurun_out/$tag/summary.txt
timeout 300 python scripts/prim_bench.py me_full 2>&1 | tail -1 | tee -a gpurun_out/$tag/summary.txt
