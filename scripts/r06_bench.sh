#!/bin/bash
# usage (through gpurun): bash scripts/r06_bench.sh <out tag> [pytest targets...]: optional GPU tests, then the default bench line with its key figures printed
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; shift
mkdir -p $out
if [ $# -gt 0 ]; then
  ( time timeout 2400 python -m pytest "$@" -q -m gpu -x ) > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
  tail -4 $out/pytest.log
fi
( time timeout 1500 python bench.py $BENCH_ARGS ) > $out/bench.log 2>&1; echo "bench rc=$?" | tee -a $out/summary.txt
grep -h '^{' $out/bench.log | tail -1 > $out/bench.json
python - $out/bench.json <<'PY' | tee -a $out/summary.txt
import json, sys
j = json.load(open(sys.argv[1]))
r = j["roofline"]
print("value", j["value"], "paced", j.get("paced_fps"), "ms/step", j["ms_per_step"], "| search solo us", r.get("us_per_search"), "frac", r.get("frac"))
ls = j["lookahead_stats"]
print("weights analysed/kept", ls["weights_analysed"], ls["weights_kept"], "device", {k: v for k, v in ls["device"].items() if k != "note"})
print("weighted", ls.get("weighted_speculation"))
for k in ("host_fed", "single_stream", "configs2_4k", "configs4_8k_1gpu", "configs3_4k_1gpu"):
    v = j.get(k)
    if isinstance(v, dict):
        v = {a: b for a, b in v.items() if a not in ("workload", "what", "amdahl", "checked", "pcie_peak_what", "exchange")}
    print(k, json.dumps(v)[:420])
rk = j.get("roofline_kernels", {})
print("kernels ms:", {k: v.get("ms") for k, v in rk.items()})
PY
tail -3 $out/bench.log | grep real
