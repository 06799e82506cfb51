#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
export X264HIP_H2D_TRACE=1
for s in 4 6 8 10 12; do
python scripts/hostfed_probe.py $s 5 2>&1 | grep -v Warning | grep -v amdgpu.ids | tee -a $out/hostfed.txt
done
