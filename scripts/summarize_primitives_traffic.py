"""HBM traffic of the streaming primitives from the two passes of scripts/pmc_primitives.sh: per kernel, FETCH_SIZE + WRITE_SIZE (KiB units,
summed over the 8 XCDs' L2s) per launch against the launch time of the same run.  Launches of one kernel differ in size (one 4K frame pair,
the 265 MB mosaic, the three partition sizes ...): they are listed by grid size.  Writes profiles/<tag>_primitives_traffic.json.
usage: python scripts/summarize_primitives_traffic.py <tag>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]


def load(counter):
    d = os.path.join(ROOT, "gpurun_out", "%s_prim_%s" % (tag, counter))
    val = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r.get("Grid_Size", 0) or 0))
                val[k].append(float(r["Counter_Value"]) * 1024)
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return val, dur


fetch, dur_f = load("FETCH_SIZE")
write, dur_w = load("WRITE_SIZE")
keep = ("pixel_cmp_batch_kernel", "hpel_stream", "hpel_filter_kernel", "frame_dct_quant", "copy16_kernel", "lowres_tiles_kernel", "lowres_kernel", "strips_kernel", "me_full")
out = {"command": "scripts/pmc_primitives.sh %s (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python scripts/prim_bench.py)" % tag,
       "note": "bytes = counter x 1024, uncorrected.  Calibration inside this very run: copy16_kernel moves 512 MiB each way -- WRITE_SIZE is exact, FETCH_SIZE reports half "
               "(16 B / lane streaming reads: the guide's gfx950 correction, x2).  Narrower reads are tallied differently (lowres_kernel: x1.2-1.3, DESIGN.md section 5), so for the other "
               "kernels fetch_over_algorithmic is a lower bound of the re-read factor and 2x it the upper bound; write_over_algorithmic is exact (the excess = the padding columns of the planes).  "
               "times are from the counter runs themselves (slower than an unprofiled launch)", "kernels": []}
# algorithmic bytes (read, written) of the launches scripts/prim_bench.py makes, by (kernel substring, total threads): SURVEY 8(d) forms, as bench.py prices them
W, H = 3840, 2160
Wm, Hm = 4 * W, 4 * H
ALG = {("copy16_kernel", 1048576): (512 << 20, 512 << 20),
       ("hpel_stream_kernel", 552960): (W * H, 3 * W * H), ("hpel_stream_kernel", 2142720): (4 * W * H, 12 * W * H),
       ("hpel_stream16_kernel", 552960): (2 * W * H, 6 * W * H),
       ("frame_dct_quant4x4_kernel", 552960): (2 * W * H, 2 * W * H + W * H // 16),
       ("pixel_cmp_batch_kernel<unsigned char, 16, 16", 2073600): (2 * Wm * Hm, 4 * Wm * Hm // 256), ("pixel_cmp_batch_kernel<unsigned char, 16, 16", 261120): (2 * 3840 * 2176, 4 * 3840 * 2176 // 256),
       ("pixel_cmp_batch_kernel<unsigned char, 8, 8", 2073600): (2 * Wm * Hm, 4 * Wm * Hm // 64), ("pixel_cmp_batch_kernel<unsigned char, 8, 8", 261120): (2 * 3840 * 2176, 4 * 3840 * 2176 // 64),
       ("pixel_cmp_batch_kernel<unsigned char, 4, 4", 2073600): (2 * Wm * Hm, 4 * Wm * Hm // 16), ("pixel_cmp_batch_kernel<unsigned char, 4, 4", 261120): (2 * 3840 * 2176, 4 * 3840 * 2176 // 16)}
for k in sorted(set(fetch) | set(write)):
    if not any(s in k[0] for s in keep):
        continue
    f, w = fetch.get(k, []), write.get(k, [])
    d = dur_f.get(k) or dur_w.get(k) or []
    if not d:
        continue
    n = max(len(f), len(w), 1)
    fb, wb, ns = sum(f) / max(len(f), 1), sum(w) / max(len(w), 1), sorted(d)[len(d) // 2]
    out["kernels"].append({"kernel": k[0][:90], "grid_size": k[1], "launches": n, "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                           "median_launch_us": round(ns / 1e3, 2), "hbm_GBps_in_this_run": round((fb + wb) / ns, 1)})
    for (name, grid), (ar, aw) in ALG.items():
        if k[0].startswith(name) and k[1] == grid:
            out["kernels"][-1].update({"algorithmic_read_bytes": ar, "algorithmic_write_bytes": aw, "fetch_over_algorithmic": round(fb / ar, 3), "write_over_algorithmic": round(wb / aw, 3)})
path = os.path.join(ROOT, "profiles", "%s_primitives_traffic.json" % tag)
json.dump(out, open(path, "w"), indent=1)
print(path)
for r in out["kernels"]:
    print("%-70s grid %9d  fetch %7.1f MB write %7.1f MB  %8.1f us  %7.1f GB/s" % (r["kernel"][:70], r["grid_size"], r["fetch_bytes_per_launch"] / 1e6, r["write_bytes_per_launch"] / 1e6,
                                                                                  r["median_launch_us"], r["hbm_GBps_in_this_run"]))
