"""Memory copies and kernels of a rocprofv3 --kernel-trace --memory-copy-trace database: the host-to-device transfers' sizes, rates,
the time the link is busy, and what the device does meanwhile.  usage: python scripts/trace_copies.py <dir>"""
import collections, glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
mt = [t for t in tabs if "memory_cop" in t.lower()]
print("tables:", [t for t in tabs if "memory" in t.lower() or t == "kernels"][:12])
if "memory_copies" in tabs:
    cols = [r[1] for r in c.execute("pragma table_info(memory_copies)")]
    print("memory_copies columns:", cols)
    rows = list(c.execute("select * from memory_copies order by start"))
    ix = {n: i for i, n in enumerate(cols)}
    big = [r for r in rows if r[ix["size"]] >= (1 << 20)]
    print("%d copies, %d of >= 1 MiB" % (len(rows), len(big)))
    if big:
        # the last second of the trace = the host-fed passes
        T1 = max(r[ix["end"]] for r in big)
        sel = [r for r in big if r[ix["start"]] > T1 - 0.6e9]
        by = collections.Counter(r[ix["size"]] >> 20 for r in sel)
        print("sizes (MiB: count) in the last 0.6 s:", sorted(by.items()))
        tot = sum(r[ix["size"]] for r in sel); busy = 0; cur = None
        for r in sorted(sel, key=lambda r: r[ix["start"]]):
            s, e = r[ix["start"]], r[ix["end"]]
            if cur is None or s > cur[1]:
                if cur: busy += cur[1] - cur[0]
                cur = [s, e]
            else:
                cur[1] = max(cur[1], e)
        if cur: busy += cur[1] - cur[0]
        span = max(r[ix["end"]] for r in sel) - min(r[ix["start"]] for r in sel)
        print("span %.1f ms, link busy %.1f ms (%.0f %%), %.1f GB moved: %.1f GB/s while busy, %.1f GB/s over the span" % (span / 1e6, busy / 1e6, 100.0 * busy / span, tot / 1e9, tot / busy, tot / span))
        d = sorted((r[ix["end"]] - r[ix["start"]]) / 1e3 for r in sel if (r[ix["size"]] >> 20) >= 30)
        if d:
            print("33 MiB transfers: median %.0f us, p10 %.0f, p90 %.0f (at 57 GB/s: 610 us)" % (d[len(d) // 2], d[len(d) // 10], d[9 * len(d) // 10]))

# ---- timeline of the last 160 ms: per stream, runs of activity (operations less than 100 us apart merged), with what ran in them
if "kernels" in tabs:
    kc = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print("kernels columns:", kc)
    kx = {n: i for i, n in enumerate(kc)}
    krows = list(c.execute("select * from kernels order by start"))
    if krows and "memory_copies" in tabs and big:
        T1 = max(r[ix["end"]] for r in big)
        T0 = T1 - 160e6
        skey = "stream_id" if "stream_id" in kx else "queue_id"
        ops = []
        for r in krows:
            if r[kx["end"]] > T0 and r[kx["start"]] < T1:
                ops.append((r[kx["start"]], r[kx["end"]], "k%s" % r[kx[skey]], str(r[kx["name"]])[:18]))
        for r in big:
            if r[ix["end"]] > T0 and r[ix["start"]] < T1:
                ops.append((r[ix["start"]], r[ix["end"]], "copy", "H2D"))
        # device busy = union of kernel intervals
        ks = sorted(o for o in ops if o[2] != "copy")
        busy = 0; cur = None
        for s, e, _, _ in ks:
            if cur is None or s > cur[1]:
                if cur: busy += cur[1] - cur[0]
                cur = [s, e]
            else:
                cur[1] = max(cur[1], e)
        if cur: busy += cur[1] - cur[0]
        print("last 160 ms: some kernel running %.1f ms (%.0f %%)" % (busy / 1e6, busy / 1.6e6))
        # sum of search-kernel time and count
        by_name = collections.defaultdict(lambda: [0, 0.0])
        for s, e, q, n in ks:
            by_name[n][0] += 1; by_name[n][1] += (e - s) / 1e6
        for n, (cnt, ms) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:10]:
            print("  %-20s %5d launches %8.2f ms summed, %.3f ms each" % (n, cnt, ms, ms / cnt))
        streams = sorted(set(o[2] for o in ops))
        for st in streams:
            runs = []
            for s, e, q, n in sorted(o for o in ops if o[2] == st):
                if runs and s - runs[-1][1] < 100e3:
                    runs[-1][1] = max(runs[-1][1], e); runs[-1][2][n] += 1
                else:
                    runs.append([s, e, collections.Counter({n: 1})])
            if sum(r[1] - r[0] for r in runs) < 2e6:
                continue
            print("stream %s:" % st)
            for s, e, cn in runs:
                if e - s > 200e3:
                    print("   %7.2f .. %7.2f ms (%6.2f)  %s" % ((s - T0) / 1e6, (e - T0) / 1e6, (e - s) / 1e6, ", ".join("%s x%d" % kv for kv in cn.most_common(4))))
