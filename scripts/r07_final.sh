#!/bin/bash
# round-6 closing run: the new host-fed test, then the profile stages of scripts/gpu_session.sh (bench line, rocprofv3 kernel stats with
# eight segments in flight and with one, counter passes of the search kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_host_fed.py tests/test_gpu_c_shard.py -x -q -m gpu 2>&1 | tail -5 | tee $out/tests.txt
bash scripts/gpu_session.sh $1 bench stats stats1 pmc
