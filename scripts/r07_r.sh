#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
rocm-smi --showcomputepartition --showmemorypartition 2>&1 | grep -v "^=\|^$" | head -8 | tee $out/summary.txt
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
( time timeout 900 python bench.py --no-cpu-baseline --no-primitives ) > $out/bench_$i.log 2>&1
echo "run $i: $(grep -c 'x264hip:' $out/bench_$i.log) timeouts; $(grep '^real' $out/bench_$i.log)" | tee -a $out/summary.txt
grep "x264hip:" $out/bench_$i.log | head -4 | tee -a $out/summary.txt
if grep -q "x264hip:" $out/bench_$i.log; then break; fi
done
