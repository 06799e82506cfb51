import sqlite3, glob, sys, collections
acc = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    c = sqlite3.connect(f)
    for name, cname, val, st, en in c.execute("select kernel_name, counter_name, value, start, end from counters_collection"):
        if "pattern" in name:
            acc[name][cname] = acc[name].get(cname, 0) + val
            acc[name]["ns"] = en - st
for k in sorted(acc, key=lambda n: int(n.split("<")[1].split(">")[0])):
    d = acc[k]
    n = d.get("SQ_INSTS_VMEM_RD", 1)
    print(k.split("(")[0], {c: round(v / n, 2) for c, v in d.items() if c not in ("SQ_INSTS_VMEM_RD", "ns")}, "instrs", int(n), "us", d["ns"] / 1e3)
