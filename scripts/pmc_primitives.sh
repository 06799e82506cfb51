#!/bin/bash
# usage (through gpurun): bash scripts/pmc_primitives.sh <tag>
# HBM traffic of the streaming primitives (SAD / SATD batch, hpel_filter, frame DCT+quant, the copy kernel): FETCH_SIZE and WRITE_SIZE in
# separate rocprofv3 --pmc passes of scripts/prim_bench.py (never combined with a trace domain other than --kernel-trace); reduce with
# `python scripts/summarize_primitives_traffic.py <tag>` (writes profiles/<tag>_primitives_traffic.json).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1
for C in FETCH_SIZE WRITE_SIZE; do
  mkdir -p gpurun_out/${tag}_prim_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/${tag}_prim_$C -o run -- python scripts/prim_bench.py > gpurun_out/${tag}_prim_$C/prim.log 2>&1
  echo "pass $C rc=$?"
done
