#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
python scripts/hostfed_probe.py 8 5 2>&1 | grep -v Warning | tee $out/hostfed.txt
python scripts/hostfed_probe.py 1 5 2>&1 | grep -v Warning | tee -a $out/hostfed.txt
python scripts/hostfed_probe.py 16 4 2>&1 | grep -v Warning | tee -a $out/hostfed.txt
rocprofv3 --kernel-trace --memory-copy-trace -d $out/trace -- python scripts/hostfed_probe.py 8 4 > $out/trace.log 2>&1
python scripts/trace_copies.py $out/trace 2>&1 | tee -a $out/hostfed.txt
rm -rf $out/trace
