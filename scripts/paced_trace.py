"""Timeline of a paced single stream from a rocprofv3 --kernel-trace database: busy union, idle gaps, per-kernel totals, and the chain
of one mini-GOP decision.  usage: python scripts/paced_trace.py <dir with *_results.db>"""
import collections
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel_dispatch" in t][0]
rows = list(c.execute("select name, start, end, queue_id from %s order by start" % kt))
rows = [r for r in rows if "at::native" not in r[0] and "rocclr" not in r[0]]
# the last pass = the last run of >= 150 lowres_tiles dispatches
lw = [i for i, r in enumerate(rows) if "lowres_tiles" in r[0]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 160
step = rows[lw[-n]:]
T0, T1 = step[0][1], max(r[2] for r in step)
print("last pass span %.2f ms, %d dispatches, %.1f us per frame" % ((T1 - T0) / 1e6, len(step), (T1 - T0) / 1e3 / n))
ev = sorted((r[1], r[2]) for r in step)
busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
for s, e in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("device busy (union) %.2f ms = %.1f %%; %d gaps, total %.2f ms, median %.1f us, > 50 us: %d" % (
    busy / 1e6, 100.0 * busy / (T1 - T0), len(gaps), sum(gaps) / 1e6, sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0, sum(1 for g in gaps if g > 50e3)))
agg = collections.defaultdict(lambda: [0, 0, 0])
for r in step:
    k = r[0].split("(")[0].replace("void ", "")[:40]
    agg[k][0] += 1; agg[k][1] += r[2] - r[1]; agg[k][2] = max(agg[k][2], r[2] - r[1])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("   %-42s n=%5d  %8.3f ms  avg %7.1f us  max %7.1f us" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3, v[2] / 1e3))
# a window of 40 consecutive dispatches in the middle, with times relative to the first
mid = len(step) // 2
print("--- 60 consecutive dispatches (start us, duration us, queue, kernel)")
b = step[mid][1]
for r in step[mid:mid + 60]:
    print("%9.1f %8.1f q%-3d %s" % ((r[1] - b) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0].split("(")[0].replace("void ", "")[:60]))
