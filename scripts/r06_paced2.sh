#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
python scripts/h2d_rates.py 2>&1 | grep -v Warning | tee $out/h2d_rates.txt
( timeout 1500 python -m pytest tests/test_gpu_host_fed.py tests/test_gpu_lookahead.py tests/test_gpu_parity.py -q -m gpu -x ) > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt; tail -3 $out/tests.log
for rep in 1 2; do
for e in "A=0" "X264HIP_LA_PUT_FLUSH=0" "X264HIP_MBT_GUARD=coarse" "X264HIP_LA_PUT_FLUSH=0 X264HIP_MBT_GUARD=coarse"; do
  env $e python scripts/paced_probe.py 4 2>/dev/null | tail -1 | sed "s/^/$e /" | tee -a $out/summary.txt
done; done
env python scripts/paced_probe.py 3 3840x2160 2>/dev/null | tail -1 | sed "s/^/4K /" | tee -a $out/summary.txt
env X264HIP_LA_PUT_FLUSH=0 X264HIP_MBT_GUARD=coarse python scripts/paced_probe.py 3 3840x2160 2>/dev/null | tail -1 | sed "s/^/4K old /" | tee -a $out/summary.txt
