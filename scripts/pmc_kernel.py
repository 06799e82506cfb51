"""Sum the counters rocprofv3 --pmc collected for one kernel.
usage: python scripts/pmc_kernel.py <dir with *_results.db or *counter_collection.csv> [kernel substring]"""
import collections
import csv
import glob
import sqlite3
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "me_rows"
acc = collections.defaultdict(float)
launches = set()
dur = 0
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            launches.add((f, r["Dispatch_Id"]))
for f in glob.glob(d + "/**/*_results.db", recursive=True):
    c = sqlite3.connect(f)
    seen = {}
    for did, name, cname, val, st, en in c.execute("select dispatch_id, kernel_name, counter_name, value, start, end from counters_collection"):
        if sub in name:
            acc[cname] += val
            launches.add((f, did))
            seen[did] = en - st
    dur += sum(seen.values())
print("launches", len(launches), "total_ns", dur)
for k in sorted(acc):
    print("%-28s %18.0f" % (k, acc[k]))
