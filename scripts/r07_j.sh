#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
for ord in ref req ref; do
echo "== X264HIP_SEARCH_ORDER=$ord" | tee -a $out/stress.txt
X264HIP_SEARCH_ORDER=$ord timeout 600 python scripts/hostfed_stress.py 8 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $out/stress.txt
done
