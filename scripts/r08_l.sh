#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lookahead.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2 | tee $out/tests.txt
for i in 1 2 3; do
v=$(python bench.py --no-cpu-baseline --no-primitives --no-extra --steps 20 --warmup 3 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['us_per_search'], j['paced_fps'])")
echo "default $v" | tee -a $out/ab.txt
done
v=$(python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --steps 6 --warmup 2 --paced --inflight 4 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'])")
echo "paced4 $v" | tee -a $out/ab.txt
