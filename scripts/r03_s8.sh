#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s8
mkdir -p gpurun_out/$tag
X264HIP_TRACE_MISS=1 timeout 300 python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --inflight 1 --steps 1 --warmup 2 > gpurun_out/$tag/miss.log 2>&1
python - <<'PY' | tee gpurun_out/r03s8/summary.txt
import re, collections
c = collections.Counter(); h = collections.Counter()
for l in open("gpurun_out/r03s8/miss.log"):
    m = re.match(r"miss b=(\d+) d0=(\d+) d1=(\d+) valid=(\d+) tags have (\d+)/(\d+)/(\d+) want (\d+)/(\d+)/(\d+) ref1_ok=(\d+) wi=(\d+) search=(\d+),(\d+) w=(\d+)", l)
    if m:
        b, d0, d1, valid, h0, h1, hr, w0, w1, wr, r1, wi, s0, s1, w = map(int, m.groups())
        why = "not speculated" if not valid else ("tag0" if h0 != w0 else "tag1" if h1 != w1 else "ref1 tag/variant" )
        c[(d0, d1, why, r1, "search" if (s0 or s1) else "")] += 1
    m = re.match(r"hit b=(\d+) d0=(\d+) d1=(\d+) ref1_ok=(\d+)", l)
    if m:
        h[(int(m.group(2)), int(m.group(3)), int(m.group(4)))] += 1
print("misses:")
for k, v in c.most_common(30): print(" ", k, v)
print("B hits by (d0,d1,ref1_ok):", sorted(h.items()))
PY
tail -2 gpurun_out/$tag/miss.log | cut -c1-300
