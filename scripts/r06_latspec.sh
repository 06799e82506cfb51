#!/bin/bash
# the latency form with / without the guessed below-left vector: parity of every kernel form, then the paced single stream (1080p and the 4K GOP)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lookahead.py -q -m gpu -x ) > $out/parity.log 2>&1; echo "parity rc=$?" | tee -a $out/summary.txt
tail -3 $out/parity.log
short="--no-cpu-baseline --no-primitives --no-extra --no-check --inflight 1 --paced --steps 6 --warmup 1"
for rep in 1 2; do
for lib in "" nospec; do
  L=$GRAFT_REPO_ROOT/x264_amd/libx264hip${lib:+_$lib}.so
  X264HIP_LIB=$L python bench.py $short 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$lib paced 1080p', j['value'])" | tee -a $out/summary.txt
done; done
X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip_prof.so timeout 300 python bench.py $short > $out/prof.log 2>&1
grep -h "ME_PROFILE" $out/prof.log | tail -4 | tee -a $out/summary.txt
