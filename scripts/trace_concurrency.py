"""How the kernels of the default bench run overlap on the device, from the rocprofv3 --kernel-trace CSV of that command (scripts/gpu_session.sh
keeps it under gpurun_out/<tag>_stats): share of the timed steps' wall time during which some kernel runs, a full-width kernel runs
(search, cells, ingest), and 0 / 1 / 2 / >= 3 search launches run at once.  Writes profiles/<tag>_trace_concurrency.json.
usage: python scripts/trace_concurrency.py <tag>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
files = glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_stats", "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]))
rows.sort()
ours = [r for r in rows if any(k in r[2] for k in ("me_rows", "cell_", "mbtree", "lowres", "strips", "intra", "aq_", "weight", "copyBuffer", "fillBuffer", "recalc"))]
# the timed steps: every context starts a step with one lowres launch (batch ingest of its segment); the default run is one warm-up step
# and three timed ones, so the timed region begins with the launch that opens the second quarter of them
lw = sorted(r[0] for r in ours if "lowres" in r[2])
t_last = max(r[1] for r in ours)
t0 = lw[len(lw) // 4] if len(lw) >= 4 else ours[0][0]
win = [r for r in ours if r[1] > t0]
wide = ("me_rows", "cell_b", "cell_p", "intra", "aq_kernel", "lowres", "strips", "weight")


def covered(intervals):
    ev = sorted(intervals)
    tot, cur_s, cur_e = 0, None, None
    for s, e in ev:
        s = max(s, t0)
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


span = t_last - t0
any_k = covered([(s, e) for s, e, n in win])
wide_k = covered([(s, e) for s, e, n in win if any(w in n for w in wide)])
# search launches in flight: sweep
ev = []
for s, e, n in win:
    if "me_rows" in n:
        ev.append((max(s, t0), 1)); ev.append((e, -1))
ev.sort()
lvl, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[min(lvl, 3)] += t - last
    last, lvl = t, lvl + d
hist[min(lvl, 3)] += t_last - last
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    per[n][0] += 1; per[n][1] += e - s
out = {"source": "gpurun_out/%s_stats (rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check), the timed steps (from the first ingest launch after the warm-up step to the last kernel)" % tag,
       "span_ms": round(span / 1e6, 2), "some_kernel_running": round(any_k / span, 4), "full_width_kernel_running": round(wide_k / span, 4),
       "search_launches_in_flight": {("%d" % k if k < 3 else ">=3"): round(v / span, 4) for k, v in sorted(hist.items())},
       "summed_kernel_time_ms": {k: [v[0], round(v[1] / 1e6, 2)] for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:16]}}
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_trace_concurrency.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
