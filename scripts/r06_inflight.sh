#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
short="--no-cpu-baseline --no-primitives --no-extra --no-check"
for rep in 1 2; do
for e in "A=0" "X264HIP_LAT_WAVES=0"; do
for n in 8 12; do
  env $e python bench.py $short --inflight $n 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e inflight $n', j['value'], j['ms_per_step'])"
done; done; done
