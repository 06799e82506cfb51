#!/bin/bash
# usage (through gpurun): bash scripts/ab_variants.sh <variant tag>...   -- short bench runs of x264_amd/libx264hip_<tag>.so builds
# (python -m x264_amd.build --variant <tag> DEF=..) against the default build: 8 segments in flight twice, one segment batched and paced
python -c "import torch; p=torch.cuda.get_device_properties(0); print('device', p.name, p.multi_processor_count, 'CUs', p.total_memory >> 30, 'GiB')"
short="--no-cpu-baseline --no-primitives --no-extra --no-check"
r() { python bench.py $short $2 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', j['value'], j['roofline']['us_per_search'])"; }
for rep in 1 2; do
for v in "" "$@"; do
  if [ -n "$v" ]; then export X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip_$v.so; else unset X264HIP_LIB; fi
  r "lib=$v" "--inflight 8"
done; done
for v in "" "$@"; do
  if [ -n "$v" ]; then export X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip_$v.so; else unset X264HIP_LIB; fi
  r "lib=$v" "--inflight 1"
  r "lib=$v" "--inflight 1 --paced"
done
