#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
for e in "A=0" "X264HIP_MBT_GUARD=coarse"; do
  echo "== $e" | tee -a $out/8k.txt
  env $e timeout 600 python scripts/r06_8k.py 2>&1 | grep -v Warning | tail -5 | tee -a $out/8k.txt
done
