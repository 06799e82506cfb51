#!/bin/bash
# two ranks on the one GPU of a gpurun box (gloo carries the collectives): exercises bench.py's N > 1 paths -- GOP segments and the
# window-shard strong-scaling measurement -- end to end.  Not a performance measurement (both ranks share one GPU).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/shard
X264HIP_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 2 --warmup 1 --no-primitives --no-cpu-baseline > gpurun_out/shard/bench2.log 2>&1; echo "bench gpus=2 (gloo rig) rc=$?"
grep -h '^{' gpurun_out/shard/bench2.log | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j.get('window_shard'))"
timeout 600 python bench.py --shard window --no-primitives --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/shard/bench1w.log 2>&1; echo "bench --shard window N=1 rc=$?"
grep -h '^{' gpurun_out/shard/bench1w.log | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['scaling'], j.get('window_shard'))"
tail -5 gpurun_out/shard/bench2.log | cut -c1-300
