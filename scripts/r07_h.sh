#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for ord in req ref req ref; do
echo "== X264HIP_SEARCH_ORDER=$ord" | tee -a $out/order.txt
X264HIP_SEARCH_ORDER=$ord python bench.py --no-cpu-baseline --no-primitives --no-extra --steps 10 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('value',j['value'],'solo us',r.get('us_per_search'),'concurrent us',r['concurrent']['us_per_search'],'kernels',{k:v.get('ms') for k,v in j['roofline_kernels'].items()})" | tee -a $out/order.txt
done
bash scripts/r07_lat.sh $1/lat 2>&1 | grep -v "^absent" | tee $out/lat.txt
