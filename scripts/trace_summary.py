"""Per-queue busy time and per-kernel totals of the LAST bench step in a rocprofv3 --kernel-trace database.
usage: python scripts/trace_summary.py <dir with *_results.db>"""
import collections
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, queue_id, grid_z from kernels order by start"))
lw = [i for i, r in enumerate(rows) if "lowres_kernel" in r[0]]
step = rows[lw[-1]:]
T0 = step[0][1]
T1 = max(r[2] for r in step)
print("last step span %.2f ms, %d dispatches" % ((T1 - T0) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r[3]].append(r)
for q, rs in sorted(byq.items()):
    print("queue %d: %d dispatches, busy %.2f ms" % (q, len(rs), sum(r[2] - r[1] for r in rs) / 1e6))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        k = r[0].split("(")[0].replace("void ", "")[:34]
        agg[k][0] += 1
        agg[k][1] += r[2] - r[1]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("   %-36s n=%4d  %8.3f ms" % (k, v[0], v[1] / 1e6))
