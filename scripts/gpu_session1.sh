#!/bin/bash
# GPU session 1 of round 2: full GPU suite (no xfail any more, 8K tests included), the new bench line, counter passes and kernel stats
# of the search kernel as it stood at the start of the round ("before").
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s1
( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 ) > gpurun_out/s1/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/s1/summary.txt
( time timeout 600 python bench.py ) > gpurun_out/s1/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/s1/summary.txt
grep -h '^{' gpurun_out/s1/bench.log | tail -1 > gpurun_out/s1/bench.json
bash scripts/pmc_search.sh r02before 2>&1 | tee -a gpurun_out/s1/summary.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s1_stats
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s1_stats -o run -- python bench.py --no-cpu-baseline --no-primitives --no-extra > gpurun_out/s1_stats/bench.log 2>&1; echo "stats rc=$?" | tee -a gpurun_out/s1/summary.txt
tail -5 gpurun_out/s1/pytest.log
cut -c1-600 gpurun_out/s1/bench.json
