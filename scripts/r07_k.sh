#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
for ord in req ref req ref req ref; do
v=$(X264HIP_SEARCH_ORDER=$ord python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --steps 20 --warmup 3 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['us_per_search'])")
echo "$ord $v" | tee -a $out/order.txt
done
for ord in req ref req ref; do
v=$(X264HIP_SEARCH_ORDER=$ord python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --steps 20 --warmup 3 --inflight 1 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['us_per_search'])")
echo "inflight1 $ord $v" | tee -a $out/order.txt
done
