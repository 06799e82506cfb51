"""Copies the rocprofv3 summaries worth judging from gpurun_out/ (scratch) into profiles/ (tracked):
kernel stats of the bench command, and the FETCH_SIZE / WRITE_SIZE passes reduced to bytes per launch."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "prof2", "run_kernel_stats.csv"), os.path.join(dst, "%s_bench_kernel_stats.csv" % tag))
with open(os.path.join(src, "prof2", "bench.log")) as f:
    line = [l for l in f if l.startswith("{")][-1]
open(os.path.join(dst, "%s_bench_under_rocprof.json" % tag), "w").write(line)


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0, 0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
        agg[k][2] += int(r["Grid_Size"])
    return agg


fetch = per_kernel(os.path.join(src, "pmc_fetch", "run_counter_collection.csv"))
write = per_kernel(os.path.join(src, "pmc_write", "run_counter_collection.csv"))
out = {"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --no-cpu-baseline --no-primitives --steps 1 --warmup 0 --frames 64 --inflight 1 (one segment: PMC values of concurrently running kernels would mix)",
       "units": "counter values are KiB (bytes = value * 1024), one pass per counter", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0, 0.0, 0]), write.get(k, [0, 0.0, 0])
    out["kernels"][k] = {"launches": f[0] or w[0], "fetch_bytes_per_launch": f[1] * 1024 / max(f[0], 1), "write_bytes_per_launch": w[1] * 1024 / max(w[0], 1)}
# calibration on a kernel with known traffic: lowres_kernel reads the 1920x1080 luma once and writes 4 padded planes
lw = out["kernels"].get("lowres_kernel")
PMC_FRAMES = 64  # bench.py --frames 64 in the PMC passes: the batched ingest runs lowres_kernel once for all of them
known_read, known_write = 1920 * 1080 * PMC_FRAMES, 4 * 608 * 1024 * PMC_FRAMES
cal_r = known_read / lw["fetch_bytes_per_launch"]
cal_w = known_write / lw["write_bytes_per_launch"]
out["calibration"] = {"kernel": "lowres_kernel (one launch, %d frames)" % PMC_FRAMES, "known_read_bytes": known_read, "known_write_bytes": known_write,
                      "read_factor": cal_r, "write_factor": cal_w}
me = out["kernels"]["me_rows_kernel"]
mb_h = 68
n_search = fetch["me_rows_kernel"][2] / fetch["me_rows_kernel"][0] / (64 * mb_h)
out["me_rows_kernel"] = {"searches_per_launch": n_search,
                         "raw_hbm_bytes_per_search": (me["fetch_bytes_per_launch"] + me["write_bytes_per_launch"]) / n_search,
                         "calibrated_hbm_bytes_per_search": (me["fetch_bytes_per_launch"] * cal_r + me["write_bytes_per_launch"] * cal_w) / n_search,
                         "algorithmic_bytes_per_search": 5 * 960 * 544 + 8 * 120 * 68}
json.dump(out, open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w"), indent=1)
json.dump({"source": "profiles/%s_pmc_summary.json" % tag, "workload": "1920x1080 slow+dia",
           "me_rows_kernel_hbm_bytes_per_search": out["me_rows_kernel"]["calibrated_hbm_bytes_per_search"]},
          open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out["me_rows_kernel"], indent=1), json.dumps(out["calibration"]))
