#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
short="--no-cpu-baseline --no-primitives --no-extra --no-check --inflight 1 --paced --steps 3 --warmup 1"
python bench.py $short 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('paced plain', j['value'], j['lookahead_stats']['host_ms'])" | tee $out/paced.txt
rocprofv3 --kernel-trace -d $out/trace -- python bench.py $short > $out/trace.log 2>&1
python scripts/paced_trace.py $out/trace 160 | tee -a $out/paced.txt
rm -rf $out/trace
