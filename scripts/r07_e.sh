#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_host_fed.py -x -q -m gpu 2>&1 | tail -5 | tee $out/tests.txt
for s in 1 4 8 12; do
python scripts/hostfed_probe.py $s 5 2>&1 | grep -v Warning | grep -v amdgpu.ids | tee -a $out/hostfed.txt
done
X264HIP_H2D_INGEST=group python scripts/hostfed_probe.py 8 5 2>&1 | grep -v Warning | grep -v amdgpu.ids | tee -a $out/hostfed.txt
