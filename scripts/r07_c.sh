#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
python scripts/hostfed_probe.py 8 5 2>&1 | grep -v Warning | tee $out/hostfed.txt
python scripts/hostfed_probe.py 16 4 2>&1 | grep -v Warning | tee -a $out/hostfed.txt
python scripts/hostfed_probe.py 4 5 2>&1 | grep -v Warning | tee -a $out/hostfed.txt
timeout 900 python -m pytest tests/test_gpu_host_fed.py tests/test_gpu_c_shard.py -x -q -m gpu 2>&1 | tail -15 | tee $out/tests.txt
rocprofv3 --kernel-trace --memory-copy-trace -d $out/trace -- python scripts/hostfed_probe.py 8 4 > $out/trace.log 2>&1
python scripts/trace_copies.py $out/trace 2>&1 > $out/timeline.txt
rm -rf $out/trace
head -12 $out/timeline.txt
