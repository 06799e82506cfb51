"""Kernels of a rocprofv3 --kernel-trace database as a timeline: per stream, runs of activity (launches less than `gap` us apart merged) over the
last `span` ms, and how much of that time some kernel / a search kernel was running.  usage: python scripts/trace_kernels.py <dir> [span_ms=40] [gap_us=20]"""
import collections, glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]
span = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
gap = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
c = sqlite3.connect(db)
kc = [r[1] for r in c.execute("pragma table_info(kernels)")]
kx = {n: i for i, n in enumerate(kc)}
rows = [r for r in c.execute("select * from kernels order by start") if "me_" in str(r[kx["name"]]) or "cell" in str(r[kx["name"]]) or "lowres" in str(r[kx["name"]]) or "mbtree" in str(r[kx["name"]]) or "intra" in str(r[kx["name"]]) or "aq_" in str(r[kx["name"]]) or "upload" in str(r[kx["name"]]) or "weight" in str(r[kx["name"]]) or "copy16" in str(r[kx["name"]])]
T1 = max(r[kx["end"]] for r in rows) - 20e6  # (leave the tail of the run out)
T0 = T1 - span * 1e6
sel = [r for r in rows if r[kx["end"]] > T0 and r[kx["start"]] < T1]
def union(rs):
    busy = 0; cur = None
    for r in sorted(rs, key=lambda r: r[kx["start"]]):
        s, e = max(r[kx["start"]], T0), min(r[kx["end"]], T1)
        if cur is None or s > cur[1]:
            if cur: busy += cur[1] - cur[0]
            cur = [s, e]
        else:
            cur[1] = max(cur[1], e)
    if cur: busy += cur[1] - cur[0]
    return busy
print("last %.0f ms: some kernel running %.1f %%, a search kernel %.1f %%" % (span, 100 * union(sel) / (span * 1e6), 100 * union([r for r in sel if "me_" in str(r[kx["name"]])]) / (span * 1e6)))
by = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    n = str(r[kx["name"]]).replace("void ", "")[:24]
    by[n][0] += 1; by[n][1] += (r[kx["end"]] - r[kx["start"]]) / 1e3
for n, (k, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-26s %5d launches %9.1f us summed %8.1f us each" % (n, k, us, us / k))
# a stretch of 3 ms in the middle, launch by launch, main stream and the others
mid = T0 + span * 0.5e6
for r in sel:
    if mid <= r[kx["start"]] < mid + 3e6:
        print("   %8.1f .. %8.1f us  stream %-3s %s  grid %s" % ((r[kx["start"]] - mid) / 1e3, (r[kx["end"]] - mid) / 1e3, r[kx["stream_id"]], str(r[kx["name"]]).replace("void ", "")[:40], r[kx["grid_x"]]))
