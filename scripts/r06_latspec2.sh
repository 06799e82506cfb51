#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
( timeout 900 python -m pytest tests/test_gpu_host_fed.py -q -m gpu -x ) > $out/hostfed_test.log 2>&1; echo "host_fed test rc=$?" | tee -a $out/summary.txt; tail -3 $out/hostfed_test.log
for rep in 1 2; do
for lib in "" nospec fused rows4 rows4spec; do
  X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip${lib:+_$lib}.so python scripts/paced_probe.py 4 2>/dev/null | tail -1 | tee -a $out/summary.txt
done; done
for lib in prof profnospec; do
  X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip_$lib.so python scripts/paced_probe.py 2 > $out/$lib.log 2>&1
  echo "== $lib" | tee -a $out/summary.txt; grep -h "ME_PROFILE\|paced" $out/$lib.log | tail -5 | tee -a $out/summary.txt
done
( time timeout 1500 python bench.py ) > $out/bench.log 2>&1; echo "bench rc=$?" | tee -a $out/summary.txt
grep -h '^{' $out/bench.log | tail -1 > $out/bench.json
python - $out/bench.json <<'PY' | tee -a $out/summary.txt
import json, sys
j = json.load(open(sys.argv[1]))
print("value", j["value"], "paced", j.get("paced_fps"))
print("host_fed", json.dumps({k: v for k, v in j["host_fed"].items() if k not in ("what", "pcie_peak_what", "checked")}))
print("single", json.dumps(j["single_stream"]["configs1"])[:300])
PY
