#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $out/trace -- python scripts/hostfed_probe.py 8 4 > $out/trace.log 2>&1
python scripts/trace_copies.py $out/trace 2>&1 | tee $out/timeline.txt
rm -rf $out/trace
