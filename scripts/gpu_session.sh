#!/bin/bash
# usage (through gpurun, from the repo root): bash scripts/gpu_session.sh <tag> [pmc]
# GPU suite, default bench line, rocprofv3 kernel stats of the bench command and (with "pmc") the counter passes of the search kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1
mkdir -p gpurun_out/$tag
( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 ) > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/$tag/summary.txt
( time timeout 600 python bench.py ) > gpurun_out/$tag/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/$tag/summary.txt
grep -h '^{' gpurun_out/$tag/bench.log | tail -1 > gpurun_out/$tag/bench.json
if [ "$2" = "pmc" ]; then bash scripts/pmc_search.sh ${tag} 2>&1 | tee -a gpurun_out/$tag/summary.txt; fi
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/${tag}_stats
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o run -- python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check > gpurun_out/${tag}_stats/bench.log 2>&1; echo "stats rc=$?" | tee -a gpurun_out/$tag/summary.txt
tail -12 gpurun_out/$tag/pytest.log
cut -c1-400 gpurun_out/$tag/bench.json
