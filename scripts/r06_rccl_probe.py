"""Can this image's RCCL put two ranks on ONE device?  (the 8-GPU node is not ours to use: a two-rank rehearsal of the C window shard over
the RCCL transport would need it).  Launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/r06_rccl_probe.py"""
import datetime
import os
import sys

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=60))
    t = torch.full((4,), rank + 1.0, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", rank, "all_reduce on one device:", t.tolist(), flush=True)
    if rank == 0:
        a = torch.arange(6, device="cuda", dtype=torch.float32)
        dist.send(a[:5], 1)
    else:
        b = torch.zeros(5, device="cuda")
        dist.recv(b, 0)
        print("rank 1 received", b.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("rank", rank, "FAILED:", repr(e)[:400], flush=True)
    sys.exit(1)
