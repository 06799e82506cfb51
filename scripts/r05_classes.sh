#!/bin/bash
# round 5, GPU call: the host lookahead states the classes its flow can ask for (x264hip_spec_classes) -- parity, then configs[3] on
# one GPU with and without the statement, and the request / speculation counts per class
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05cls; mkdir -p $out
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lookahead.py tests/test_gpu_fuzz.py -q -m gpu -x ) > $out/parity.log 2>&1; echo "parity rc=$?"; tail -4 $out/parity.log
short="--no-cpu-baseline --no-primitives --no-extra --no-check"
for E in "A=0" "X264HIP_NO_STATIC_CLASSES=1" "A=1" "X264HIP_NO_STATIC_CLASSES=1"; do
  env $E timeout 300 python bench.py $short --shard window > $out/w.log 2>&1
  echo "$E: $(grep -h '^{' $out/w.log | tail -1 | cut -c1-260)"
done
X264HIP_TRACE_CLASSES=1 timeout 300 python bench.py $short --shard window > $out/classes.log 2>&1; grep -h "L0 d1\|(1,0)" $out/classes.log | tail -4
for E in "A=0" "X264HIP_NO_STATIC_CLASSES=1"; do
  env $E timeout 300 python bench.py $short > $out/h.log 2>&1
  echo "$E headline: $(grep -h '^{' $out/h.log | tail -1 | cut -c1-160)"
done
