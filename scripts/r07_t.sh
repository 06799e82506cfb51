#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $out/tests.txt
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do
( time timeout 900 python bench.py --no-cpu-baseline --no-primitives ) > $out/bench_$i.log 2>&1
v=$(grep '^{' $out/bench_$i.log | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['us_per_search'], j['paced_fps'], j['single_stream']['configs1']['paced_fps'])")
echo "run $i: $(grep -c 'x264hip:' $out/bench_$i.log) timeouts; $v; $(grep '^real' $out/bench_$i.log)" | tee -a $out/summary.txt
grep "x264hip:" $out/bench_$i.log | head -4 | tee -a $out/summary.txt
done
