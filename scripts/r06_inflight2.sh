#!/bin/bash
# fade clip / round-5 clip at 8, 12, 16 segments in flight (--no-check runs)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$1
short="--no-cpu-baseline --no-primitives --no-extra --no-check --steps 20"
for e in "A=0" "X264HIP_BENCH_NO_STILL=1"; do
for n in 8 12 16; do
  env $e python bench.py $short --inflight $n 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e inflight $n', j['value'], j['ms_per_step'], j['lookahead_stats']['host_ms'])"
done; done | tee gpurun_out/$1/inflight.txt
