"""Reduce the counter passes of scripts/pmc_search.sh to profiles/<tag>_search_pmc.json: counter sums of the search kernel over
its launches, the number of block searches they covered (from the bench line of the same run), per-block figures and the
derivation of what bounds the kernel.
usage: python scripts/summarize_search_pmc.py <tag> [kernel substring] [note]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "me_rows_kernel"
note = sys.argv[3] if len(sys.argv) > 3 else ""
out = {"command": "scripts/pmc_search.sh %s  (rocprofv3 --kernel-trace --pmc <group> -- python bench.py --no-cpu-baseline --no-primitives --no-extra "
                  "--no-check --steps 1 --warmup 0 --frames 64 --inflight 1; one run per counter group)" % tag,
       "kernel": sub, "note": note, "passes": [], "counters": {}}
blocks = searches = None
dur_ns = []
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_p*"))):
    acc = collections.defaultdict(float)
    launches = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                launches.add(r["Dispatch_Id"])
    kt = 0
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                kt += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    line = None
    try:
        line = json.loads([l for l in open(os.path.join(d, "bench.log")) if l.startswith("{")][-1])
    except Exception:
        pass
    if line and searches is None:
        searches = line["roofline"]["searches"]
        w = line["config"]["workload"].split()[0].split("x")
        mb = ((int(w[0]) + 15) // 16) * ((int(w[1]) + 15) // 16)
        blocks = searches * mb
        out["workload"] = line["config"]["workload"]
    out["passes"].append({"dir": os.path.basename(d), "launches": len(launches), "kernel_ns": kt, "counters": dict(acc)})
    if kt:
        dur_ns.append(kt)
    out["counters"].update(acc)
c = out["counters"]
out["searches"], out["block_searches"] = searches, blocks
if blocks:
    ns = sum(dur_ns) / max(len(dur_ns), 1)
    out["kernel_ns_avg_over_passes"] = ns
    pb = {k: v / blocks for k, v in c.items()}
    d = {"ns_per_block_chip": ns / blocks, "us_per_search": ns / 1e3 / searches}
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_IFETCH"):
        if k in pb:
            d[k.lower()[3:] + "_per_block"] = round(pb[k], 2)
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"):
            if k in c:
                d[k.lower()[3:] + "_share_of_wave_cycles"] = round(c[k] / wc, 4)
        if "SQ_BUSY_CYCLES" in c and "SQ_WAVES" in c:
            d["avg_wave_quad_cycles"] = round(wc / c["SQ_WAVES"], 1)
    if "SQ_INSTS_VALU" in c and ns:
        # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles; 1024 SIMDs; clock from GRBM_GUI_ACTIVE / wall when collected
        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs
        clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur_ns[0] if dur_ns else ns) if c.get("GRBM_GUI_ACTIVE") else 2.4
        d["effective_clock_GHz"] = round(clk, 3)
        d["valu_issue_utilisation"] = round(c["SQ_INSTS_VALU"] * 2 / (1024 * ns * clk), 4)
        if "SQ_INSTS_SALU" in c:
            # one scalar issue per CU-cycle at best (256 CUs)
            d["salu_issue_utilisation_1_per_cu_cycle"] = round(c["SQ_INSTS_SALU"] / (256 * ns * clk), 4)
    if "TCC_HIT_sum" in c:
        d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0), 1), 4)
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c:
        d["l1_miss_share"] = round(c["TCP_TCC_READ_REQ_sum"] / max(c["TCP_TOTAL_CACHE_ACCESSES_sum"], 1), 4)
    if "FETCH_SIZE" in c:
        d["fetch_bytes_per_search_raw"] = round(c["FETCH_SIZE"] * 1024 / searches)
    if "WRITE_SIZE" in c:
        d["write_bytes_per_search_raw"] = round(c["WRITE_SIZE"] * 1024 / searches)
    if "SQ_LDS_IDX_ACTIVE" in c:
        d["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c["SQ_LDS_IDX_ACTIVE"], 1), 4)
    out["derived"] = d
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
path = os.path.join(ROOT, "profiles", "%s_search_pmc.json" % tag)
json.dump(out, open(path, "w"), indent=1)
print(path)
print(json.dumps(out.get("derived"), indent=1))
