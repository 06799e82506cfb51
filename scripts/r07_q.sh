#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
( time timeout 900 python bench.py --no-cpu-baseline --no-primitives ) > $out/bench.log 2>&1
grep "x264hip:" $out/bench.log | head -8 | tee -a $out/summary.txt
grep '^{' $out/bench.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('value',j['value'],'host_fed',json.dumps({k:v for k,v in j.get('host_fed',{}).items() if k in ('fps','error','failed_attempts','fps_by_segments_in_flight')})[:600])
print('configs4',json.dumps(j.get('configs4_8k_1gpu'))[:300])" | tee -a $out/summary.txt
grep real $out/bench.log | tee -a $out/summary.txt
