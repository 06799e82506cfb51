#!/bin/bash
# usage (through gpurun, from the repo root): bash scripts/gpu_session.sh <tag> <stage...>
# stages: parity (search + lookahead parity tests), suite (whole GPU suite), ab (short bench runs: default dispatch vs X264HIP_SEARCH=rows,
# 8 / 1 contexts, batched and paced), prof (cycle breakdown of the search kernel from the -DME_PROFILE build), bench (default bench line),
# stats (rocprofv3 kernel stats of the bench command), stats1 (the same with one segment in flight), pmc (counter passes of the search kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
short="--no-cpu-baseline --no-primitives --no-extra --no-check"
line() { python - "$1" "$2" <<'PY' | tee -a $out/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    r = j["roofline"]
    print("%-44s fps %9.1f  us/search %7.3f  launch ms %7.3f  searches/launch %6.0f" % (sys.argv[1], j["value"], r["us_per_search"], r["avg_launch_ms"], r["searches"] / max(r["launches"], 1)))
except Exception as e:
    print("%-44s FAILED %s" % (sys.argv[1], e))
PY
}
for stage in "$@"; do
case $stage in
parity)
  ( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lookahead.py -q -m gpu -x ) > $out/parity.log 2>&1; echo "parity rc=$?" | tee -a $out/summary.txt
  tail -4 $out/parity.log ;;
suite)
  ( time timeout 2400 python -m pytest tests -q -m gpu -x --durations=8 ) > $out/pytest.log 2>&1; echo "suite rc=$?" | tee -a $out/summary.txt
  tail -6 $out/pytest.log ;;
ab)
  i=0
  for E in "A=0" "X264HIP_SEARCH=rows"; do
    for B in "--inflight 8" "--inflight 1" "--inflight 8 --paced" "--inflight 1 --paced"; do
      i=$((i+1))
      env $E timeout 300 python bench.py $short $B > $out/ab_$i.log 2>&1
      line "$E $B" $out/ab_$i.log
    done
  done ;;
lat)
  i=0
  for E in "X264HIP_LAT_WAVES=2048" "X264HIP_LAT_WAVES=8192" "X264HIP_LAT_WAVES=16384"; do
    for B in "--inflight 8 --paced" "--inflight 1 --paced" "--inflight 8"; do
      i=$((i+1))
      env $E timeout 300 python bench.py $short $B > $out/lat_$i.log 2>&1
      line "$E $B" $out/lat_$i.log
    done
  done ;;
prof)
  for E in "A=0" "X264HIP_SEARCH=rows"; do
    for B in "--inflight 1" "--inflight 1 --paced"; do
      env $E X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip_prof.so timeout 300 python bench.py $short $B > $out/prof.log 2>&1
      echo "== $E $B" | tee -a $out/summary.txt; grep -h "ME_PROFILE" $out/prof.log | tail -4 | tee -a $out/summary.txt
    done
  done ;;
bench)
  ( time timeout 900 python bench.py ) > $out/bench.log 2>&1; echo "bench rc=$?" | tee -a $out/summary.txt
  grep -h '^{' $out/bench.log | tail -1 > $out/bench.json; cut -c1-600 $out/bench.json ;;
stats)
  cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
  mkdir -p ${out}_stats
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_stats -o run -- python bench.py $short > ${out}_stats/bench.log 2>&1; echo "stats rc=$?" | tee -a $out/summary.txt ;;
stats1)
  # the same summary with ONE segment in flight: every kernel's own duration (no launches of other contexts stretching it)
  cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
  mkdir -p ${out}_stats1
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_stats1 -o run -- python bench.py $short --inflight 1 > ${out}_stats1/bench.log 2>&1; echo "stats1 rc=$?" | tee -a $out/summary.txt ;;
window)
  cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
  timeout 600 python scripts/window_profile.py $WINDOW_ARGS > $out/window.log 2>&1; grep -h '^{' $out/window.log | cut -c1-420 | tee -a $out/summary.txt
  mkdir -p ${out}_wstats ${out}_wapi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_wstats -o run -- python scripts/window_profile.py $WINDOW_ARGS --modes plain --passes 1 > ${out}_wstats/log.txt 2>&1; echo "wstats rc=$?" | tee -a $out/summary.txt
  timeout 600 rocprofv3 --hip-runtime-trace --stats --output-format csv -d ${out}_wapi -o run -- python scripts/window_profile.py $WINDOW_ARGS --modes plain --passes 1 > ${out}_wapi/log.txt 2>&1; echo "wapi rc=$?" | tee -a $out/summary.txt
  find ${out}_wstats ${out}_wapi -name "*_trace.csv" -size +8M -delete ;;
pmc)
  bash scripts/pmc_search.sh ${tag} 2>&1 | tee -a $out/summary.txt ;;
esac
done
