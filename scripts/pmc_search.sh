#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/pmc_search.sh <tag> [extra bench.py args]
# The counter passes that say what bounds the search kernel (me_rows_kernel): one rocprofv3 --pmc run of a short bench
# command per counter group (SQ has 8 slots per pass, TCC 4; FETCH_SIZE / WRITE_SIZE need a pass each -- MI355X guide).
# --pmc is never combined with any trace domain other than --kernel-trace.  One segment at a time (--inflight 1) and no
# verification pass (--no-check): counters of concurrently running kernels would mix.
# Results: gpurun_out/<tag>_p<N>/ ; reduce with `python scripts/summarize_search_pmc.py <tag>` (writes profiles/<tag>_search_pmc.json).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU"
 "SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_READ_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for C in "${PASSES[@]}"; do
  i=$((i+1))
  if [ -n "$ONLY_PASSES" ] && [[ " $ONLY_PASSES " != *" $i "* ]]; then continue; fi   # ONLY_PASSES="6 7": just the traffic passes
  mkdir -p gpurun_out/${tag}_p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/${tag}_p$i -o run -- \
     python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --steps 1 --warmup 0 --frames ${PMC_FRAMES:-160} --inflight 1 "$@" > gpurun_out/${tag}_p$i/bench.log 2>&1
  echo "pass $i rc=$? ($C)"
done
