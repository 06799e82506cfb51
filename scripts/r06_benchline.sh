#!/bin/bash
# usage: bash scripts/r06_benchline.sh <tag> [bench args]: the default bench line, key figures only
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; shift; mkdir -p $out
( time timeout 1500 python bench.py "$@" ) > $out/bench.log 2>&1; echo "bench rc=$?" | tee -a $out/summary.txt
grep -h '^{' $out/bench.log | tail -1 > $out/bench.json
python - $out/bench.json <<'PY' | tee -a $out/summary.txt
import json, sys
j = json.load(open(sys.argv[1]))
r = j["roofline"]
print("value", j["value"], "paced", j.get("paced_fps"), "ms/step", j["ms_per_step"], "| search solo us", r.get("us_per_search"), "frac", r.get("frac"), "| round5 clip", (j.get("round5_clip") or {}).get("value"))
d = j["lookahead_stats"]["device"]
print("unclaimed", d["unclaimed_field_share"], "unused cells", d["unused_cell_share"], "host_ms", j["lookahead_stats"]["host_ms"])
for k in ("host_fed", "single_stream", "configs2_4k", "configs4_8k_1gpu", "configs3_4k_1gpu"):
    v = j.get(k)
    if isinstance(v, dict):
        v = {a: b for a, b in v.items() if a not in ("workload", "what", "amdahl", "checked", "pcie_peak_what", "exchange")}
        if k == "single_stream":
            v = {a: {x: y for x, y in b.items() if x in ("batched_fps", "paced_fps", "error")} for a, b in v.items()}
    print(k, json.dumps(v)[:400])
print("kernels ms:", {k: v.get("ms") for k, v in j.get("roofline_kernels", {}).items()})
PY
