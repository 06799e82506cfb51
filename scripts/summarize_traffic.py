"""Per-kernel HBM traffic from the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_search.sh (passes 6 and 7 of <tag>): bytes per
launch and, for the ingest kernels, per frame against the picture size.  Writes profiles/<tag>_traffic.json and profiles/pmc_traffic.json
(the per-search figure bench.py scales into roofline.traffic).
usage: python scripts/summarize_traffic.py <tag> [frames_per_launch=160]     (pmc_search.sh: PMC_FRAMES, default 160)"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 160


def per_kernel(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


fetch, write = per_kernel(tag + "_p6", "FETCH_SIZE"), per_kernel(tag + "_p7", "WRITE_SIZE")
line = json.loads([l for l in open(os.path.join(ROOT, "gpurun_out", tag + "_p6", "bench.log")) if l.startswith("{")][-1])
searches = line["roofline"]["searches"]
out = {"command": "scripts/pmc_search.sh %s, passes 6 (FETCH_SIZE) and 7 (WRITE_SIZE); counter values are KiB" % tag, "workload": line["config"]["workload"],
       "frames_per_ingest_launch": frames, "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0, 0.0]), write.get(k, [0, 0.0])
    out["kernels"][k] = {"launches": f[0] or w[0], "fetch_bytes_per_launch": f[1] * 1024 / max(f[0], 1), "write_bytes_per_launch": w[1] * 1024 / max(w[0], 1)}
# calibration on a kernel with known traffic (the guide: FETCH_SIZE under-reports wide coalesced reads; WRITE_SIZE is exact here):
# the lowres kernel reads the 1920x1080 luma of every frame once and writes 4 padded planes of 604 x 1024; lowres_tiles_kernel (the
# default since round 4) also writes their strip copies, twice that again (the two-kernel form left those to strips_kernel)
fused = "lowres_tiles_kernel" in out["kernels"]
lw = out["kernels"].get("lowres_tiles_kernel") or out["kernels"].get("lowres_kernel")
if lw:
    known_r, known_w = 1920 * 1080 * frames, (12 if fused else 4) * 604 * 1024 * frames
    cal_r, cal_w = known_r / lw["fetch_bytes_per_launch"], known_w / lw["write_bytes_per_launch"]
    out["calibration"] = {"kernel": "lowres_tiles_kernel" if fused else "lowres_kernel", "known_read_bytes": known_r, "known_write_bytes": known_w, "read_factor": cal_r, "write_factor": cal_w}
    st = out["kernels"].get("strips_kernel")
    if st:
        st["write_over_strip_bytes"] = st["write_bytes_per_launch"] / (8 * 608 * 1024 * frames)
    pic = 1920 * 1080
    for k in ("aq_kernel", "intra_kernel", "lowres_kernel", "lowres_tiles_kernel"):
        if k in out["kernels"]:
            v = out["kernels"][k]
            v["calibrated_fetch_bytes_per_frame"] = v["fetch_bytes_per_launch"] * cal_r / frames
            v["fetch_over_picture_bytes"] = v["calibrated_fetch_bytes_per_frame"] / pic
    me = out["kernels"].get("me_rows_kernel")
    if me:
        n = searches / max(me["launches"], 1)
        per_search = (me["fetch_bytes_per_launch"] * cal_r + me["write_bytes_per_launch"] * cal_w) / n
        out["me_rows_kernel"] = {"searches_per_launch": n, "calibrated_hbm_bytes_per_search": per_search,
                                 "fetch_bytes_per_search_calibrated": me["fetch_bytes_per_launch"] * cal_r / n,
                                 "write_bytes_per_search_calibrated": me["write_bytes_per_launch"] * cal_w / n,
                                 "algorithmic_bytes_per_search": 5 * 960 * 544 + 8 * 120 * 68,
                                 "result_bytes_per_search": 12 * 120 * 68,
                                 "write_amplification": me["write_bytes_per_launch"] * cal_w / n / ( 12 * 120 * 68 )}
        json.dump({"source": "profiles/%s_traffic.json" % tag, "workload": "1920x1080 slow+dia", "me_rows_kernel_hbm_bytes_per_search": per_search},
                  open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_traffic.json" % tag), "w"), indent=1)
for k in ("aq_kernel", "intra_kernel", "lowres_kernel", "lowres_tiles_kernel"):
    if k in out["kernels"]:
        print(k, {a: round(b, 3) for a, b in out["kernels"][k].items()})
print(json.dumps(out.get("me_rows_kernel"), indent=1))
