#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_host_fed.py -x -q -m gpu 2>&1 | tail -4 | tee $out/tests.txt
for i in 1 2 3; do
v=$(python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --steps 20 --warmup 3 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['us_per_search'])")
echo "value $v" | tee -a $out/value.txt
done
python scripts/hostfed_probe.py 8 8 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $out/value.txt
python scripts/hostfed_probe.py 4 8 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $out/value.txt
python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --steps 10 --warmup 2 --paced 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('paced 8 streams', j['value'])" | tee -a $out/value.txt
