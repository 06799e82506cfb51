#!/bin/bash
# round 5, GPU call: what MB-tree costs the default workload and whether more macroblocks in flight per thread (MBT_UNROLL 8, libx264hip_mbt8.so)
# or smaller workgroups change it; 40 timed steps per run, three runs each, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05mbt; mkdir -p $out
short="--no-cpu-baseline --no-primitives --no-extra --no-check --steps 40 --warmup 4"
for rep in 1 2 3; do
for C in ":A=0" "_mbt8:A=0" ":X264HIP_MBT_THREADS=512" "_mbt8:X264HIP_MBT_THREADS=512" ":X264HIP_MBT_SKIP=1" ":X264HIP_MBT=lds"; do
  V=${C%%:*}; E=${C#*:}
  env $E X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip$V.so timeout 300 python bench.py $short > $out/t.log 2>&1
  echo "lib$V $E: $(grep -h '^{' $out/t.log | tail -1 | cut -c36-60)"
done
done
