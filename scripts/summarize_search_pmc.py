"""Reduce the counter passes of scripts/pmc_search.sh to profiles/<tag>_search_pmc.json: counter sums of the search kernel over
its launches, the number of block searches they covered (from the bench line of the same run), per-block figures and the
derivation of what bounds the kernel.
usage: python scripts/summarize_search_pmc.py <tag> [kernel substring] [note] [issue]     ("issue": also refresh profiles/search_issue.json)"""
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
        # cycles a wave64 VALU instruction occupies its SIMD: 2 for the plain VOP2 kinds, 4 for the rest (experiments/gen_valu_rate.py);
        # the kernel's mix comes from scripts/valu_mix.py.  1024 SIMDs; clock from GRBM_GUI_ACTIVE / wall when collected
        # (GRBM_GUI_ACTIVE comes back summed over the 8 XCDs)
        clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur_ns[0] if dur_ns else ns) if c.get("GRBM_GUI_ACTIVE") else 2.4
        d["effective_clock_GHz"] = round(clk, 3)
        try:
            cpv = json.load(open(os.path.join(ROOT, "profiles", "search_valu_mix.json")))["cycles_per_valu"]
        except Exception:
            cpv = 4.0
        d["cycles_per_valu_instruction"] = cpv
        d["valu_issue_utilisation"] = round(c["SQ_INSTS_VALU"] * cpv / (1024 * ns * clk), 4)
        if "SQ_WAVE_CYCLES" in c:
            # resident waves averaged over the launch, of 4096 slots at 4 waves/SIMD (SQ_WAVE_CYCLES counts quad-cycles)
            d["avg_resident_waves"] = round(c["SQ_WAVE_CYCLES"] * 4 / (ns * clk), 1)
            d["valu_issue_utilisation_while_resident"] = round(d["valu_issue_utilisation"] / max(d["avg_resident_waves"] / 4096, 1e-9), 4)
        if "SQ_INSTS_SALU" in c:
            # one scalar issue per CU-cycle at best (256 CUs)
            d["salu_issue_utilisation_1_per_cu_cycle"] = round(c["SQ_INSTS_SALU"] / (256 * ns * clk), 4)
    if "TCC_HIT_sum" in c:
        d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0), 1), 4)
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c:
        d["l1_miss_share"] = round(c["TCP_TCC_READ_REQ_sum"] / max(c["TCP_TOTAL_CACHE_ACCESSES_sum"], 1), 4)
        d["l1_line_accesses_per_block"] = round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / blocks, 2)
        # one tag lookup (64-byte line) per cycle and CU
        d["l1_pipe_utilisation"] = round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256 * ns * d.get("effective_clock_GHz", 2.4)), 4)
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
if len(sys.argv) > 4 and sys.argv[4] == "issue" and out.get("derived"):
    # the figures bench.py's issue roofline is computed from
    d = out["derived"]
    json.dump({"source": "profiles/%s_search_pmc.json (SQ_INSTS_VALU pass of scripts/pmc_search.sh: wave-level VALU instructions of %s / block searches "
                         "of the same run) and profiles/search_valu_mix.json (scripts/valu_mix.py)" % (tag, sub),
               "valu_per_block": d.get("insts_valu_per_block"), "salu_per_block": d.get("insts_salu_per_block"), "vmem_rd_per_block": d.get("insts_vmem_rd_per_block"),
               "lds_per_block": d.get("insts_lds_per_block"), "cycles_per_valu": d.get("cycles_per_valu_instruction"),
               "l1_line_accesses_per_block": d.get("l1_line_accesses_per_block"), "wait_any_share_of_wave_cycles": d.get("wait_any_share_of_wave_cycles"),
               "valu_issue_utilisation_while_resident": d.get("valu_issue_utilisation_while_resident"), "effective_clock_GHz": d.get("effective_clock_GHz")},
              open(os.path.join(ROOT, "profiles", "search_issue.json"), "w"), indent=1)
