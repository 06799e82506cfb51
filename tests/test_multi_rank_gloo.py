"""N>1 path on CPU: two processes (gloo), each runs the lookahead of its own GOP segment (oracle backend
standing in for the device), summaries are all-gathered; rank 0 checks the gathered sequence against
standalone runs of the segments."""
import os
import socket
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from tests.oracle_backend import OracleBackend
    from x264_amd import lib, shard
    from x264_amd.synth import make_clip
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 176, 144
    clip = make_clip(W, H, n_total, seed=77, scene_cuts=(9, 30))
    lo, hi = shard.segment_bounds(n_total, rank, world)
    cfg = lib.la_config(W, H, "medium")
    la = lib.Lookahead(cfg, backend=OracleBackend(cfg).struct)
    outs = la.run(clip[lo:hi])
    la.close()
    summ = shard.summarize(outs, lo)
    allsum = shard.gather_summaries(summ, dist)
    dist.barrier()
    if rank == 0:
        q.put(allsum)
    dist.destroy_process_group()


def test_two_rank_segments_gather():
    sys.path.insert(0, ROOT)
    from tests.oracle_backend import OracleBackend
    from x264_amd import lib, shard
    from x264_amd.synth import make_clip
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    n_total, world = 48, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got.shape == (n_total, 4)
    assert sorted(got[:, 0].tolist()) == list(range(n_total))
    # standalone reference of each segment
    clip = make_clip(176, 144, n_total, seed=77, scene_cuts=(9, 30))
    cfg = lib.la_config(176, 144, "medium")
    exp = []
    for r in range(world):
        lo, hi = shard.segment_bounds(n_total, r, world)
        la = lib.Lookahead(cfg, backend=OracleBackend(cfg).struct)
        exp.append(shard.summarize(la.run(clip[lo:hi]), lo))
        la.close()
    assert np.array_equal(got, np.concatenate(exp))
    assert got[got[:, 0] == 24][0, 1] == 1  # a segment starts with an IDR
