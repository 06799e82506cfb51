#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for prio in 0 1 0 1; do
echo "== X264HIP_PRIO=$prio" | tee -a $out/prio.txt
X264HIP_PRIO=$prio python scripts/hostfed_probe.py 8 8 2>&1 | grep -v Warning | grep -v amdgpu.ids | tee -a $out/prio.txt
done
echo "== X264HIP_PRIO=1, 4 and 12 segments" | tee -a $out/prio.txt
X264HIP_PRIO=1 python scripts/hostfed_probe.py 4 8 2>&1 | grep -v Warning | grep -v amdgpu.ids | tee -a $out/prio.txt
X264HIP_PRIO=1 python scripts/hostfed_probe.py 12 6 2>&1 | grep -v Warning | grep -v amdgpu.ids | tee -a $out/prio.txt
