#!/bin/bash
# round 5: the default library against several variant builds, 40 timed steps per run, interleaved: bash scripts/r05_ab_libs.sh <reps> <tag> [<tag> ...]
cd "$GRAFT_REPO_ROOT" || exit 1
reps=$1; shift
out=gpurun_out/r05ab; mkdir -p $out
short="--no-cpu-baseline --no-primitives --no-extra --no-check --steps 40 --warmup 4"
for rep in $(seq $reps); do
for V in "" "$@"; do
  for B in "--inflight 8" "--inflight 1"; do
  X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip${V:+_$V}.so timeout 300 python bench.py $short $B > $out/t.log 2>&1
  echo "lib${V:+_$V} $B: $(grep -h '^{' $out/t.log | tail -1 | cut -c36-60)"
  done
done
done
