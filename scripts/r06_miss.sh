cd "$GRAFT_REPO_ROOT"
X264HIP_TRACE_MISS=1 python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --inflight 1 --steps 1 --warmup 0 2> gpurun_out/miss.txt > /dev/null
grep -c "^miss" gpurun_out/miss.txt
python - <<'PY'
import re, collections
c = collections.Counter()
for l in open("gpurun_out/miss.txt"):
    m = re.match(r"miss b=(\d+) d0=(\d+) d1=(\d+) valid=(\d) tags have (\d+)/(\d+)/(\d+) want (\d+)/(\d+)/(\d+) ref1_ok=(\d) wi=(\d) search=(\d),(\d) w=(\d)", l)
    if not m: continue
    b, d0, d1, valid, h0, h1, hr, w0, w1, wr, r1ok, wi, s0, s1, w = map(int, m.groups())
    why = []
    if not valid: why.append("novalid")
    if h0 != w0: why.append("tag0")
    if h1 != w1: why.append("tag1")
    if hr != wr: why.append("tagr")
    c[(d0, d1, "+".join(why), "w" if w else "")] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1]): print(k, v)
PY
