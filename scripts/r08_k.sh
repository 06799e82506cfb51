#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
for i in 1 2; do
for e in A=0 X264HIP_LAT_ALWAYS=1; do
v=$(env $e python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --steps 6 --warmup 2 --paced 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'])")
echo "paced8 $e $v" | tee -a $out/ab.txt
v=$(env $e python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --steps 6 --warmup 2 --paced --inflight 4 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'])")
echo "paced4 $e $v" | tee -a $out/ab.txt
done
done
