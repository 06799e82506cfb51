#!/bin/bash
# usage (through gpurun): bash scripts/r06_pmc_ab.sh <variant tag>...  -- instruction counters (passes 1, 2) of the search kernel for the default build and the variants
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "" "$@"; do
  if [ -n "$v" ]; then export X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip_$v.so; else unset X264HIP_LIB; fi
  X264HIP_BENCH_NO_STILL=1 X264HIP_SEARCH=rows ONLY_PASSES="1 2" bash scripts/pmc_search.sh r06pmc_${v:-main} > /dev/null 2>&1
  python scripts/summarize_search_pmc.py r06pmc_${v:-main} > gpurun_out/r06pmc_${v:-main}.txt 2>&1
  echo "== ${v:-main}"; tail -25 gpurun_out/r06pmc_${v:-main}.txt
done
