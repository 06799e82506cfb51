#!/bin/bash
# Run on the GPU box from the repo root (gpurun): kernel stats of the default bench command and the two HBM
# traffic passes (FETCH_SIZE / WRITE_SIZE in separate runs, as the MI355X guide prescribes).  Outputs land in
# gpurun_out/; `python scripts/summarize_profiles.py <tag>` then copies the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof2 gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2 -o run -- python bench.py --no-cpu-baseline --no-primitives > gpurun_out/prof2/bench.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o run -- python bench.py --no-cpu-baseline --no-primitives --steps 1 --warmup 0 --frames 64 --inflight 1 > gpurun_out/pmc_fetch/bench.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o run -- python bench.py --no-cpu-baseline --no-primitives --steps 1 --warmup 0 --frames 64 --inflight 1 > gpurun_out/pmc_write/bench.log 2>&1
grep -h '^{' gpurun_out/prof2/bench.log | tail -1 | cut -c1-200
ls gpurun_out/prof2 gpurun_out/pmc_fetch gpurun_out/pmc_write
