#!/bin/bash
# usage (gpurun): bash scripts/r07_ab_libs.sh <tag> <rounds> <lib.so>...: builds of the library against each other, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out; rounds=$2; shift; shift
for i in $(seq 1 $rounds); do
for lib in "$@"; do
v=$(X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/$lib python bench.py --no-cpu-baseline --no-primitives --no-extra --steps 20 --warmup 3 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['us_per_search'], j['paced_fps'])")
echo "$lib $v" | tee -a $out/ab.txt
done
done
