#!/bin/bash
# What do the search kernel's loads wait for?  Latency and stall counters of the L1 (TCP), the texture addresser (TA) and the L2's memory side
# (TCC_EA), one rocprofv3 --pmc pass each over one segment alone (as scripts/pmc_search.sh).  usage (gpurun): bash scripts/r07_lat.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; out=gpurun_out/$tag; mkdir -p $out
rocprofv3 -L > $out/avail.txt 2>&1
grep -o "\b\(TCP\|TA\|TCC\|TD\)_[A-Za-z0-9_]*" $out/avail.txt | sort -u > $out/names.txt
wc -l $out/names.txt
have() { grep -qx "$1" $out/names.txt; }
PASSES=(
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr"
 "TCC_EA_RDREQ_sum TCC_EA_RD_UNCACHED_32B_sum TCC_EA_RDREQ_32B_sum TCC_EA_RDREQ_LEVEL_sum"
 "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA_RDREQ_DRAM_sum TCC_EA_RDREQ_IO_CREDIT_STALL_sum"
)
i=0
for C in "${PASSES[@]}"; do
  i=$((i+1))
  use=""
  for c in $C; do if have $c; then use="$use $c"; else echo "absent: $c"; fi; done
  [ -z "$use" ] && continue
  mkdir -p $out/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $use --output-format csv -d $out/p$i -o run -- \
     python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --steps 1 --warmup 0 --frames 160 --inflight 1 > $out/p$i/bench.log 2>&1
  echo "pass $i rc=$? ($use)"
  python - $out/p$i <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter csv"); sys.exit()
tot = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(f[0])):
    if "me_rows_kernel" in r.get("Kernel_Name", ""):
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n += 1
print({k: v for k, v in tot.items()}, "rows", n)
PY
  rm -rf $out/p$i/*/  2>/dev/null
done
