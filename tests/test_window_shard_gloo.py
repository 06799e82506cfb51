"""SURVEY 8(e), the primary design: ONE stream, one lookahead window, the frames dealt round-robin to the ranks.  Two processes
(gloo, CPU): rank b % 2 runs the unweighted motion searches of frame b, the fields are gathered to rank 0, which decides.  The
gathered run must give the slice types and every cost cell of the SINGLE-STREAM run (a committed golden fixture generated from the
reference): no IDR is forced at any rank boundary, unlike the GOP-segment form of tests/test_multi_rank_gloo.py.  The oracle
backend stands in for the device (tests/oracle_backend.py); the GPU form of the same protocol is x264_amd.shard.HipAdapter."""
import os
import socket
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = "medium_cif"


def _case():
    from tests.golden.make_golden import LOOKAHEAD_CASES
    return LOOKAHEAD_CASES[CASE]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from tests.oracle_backend import OracleBackend, OracleShardAdapter
    from x264_amd import lib, shard
    from x264_amd.synth import make_clip
    dist.init_process_group("gloo", rank=rank, world_size=world)
    preset, opts, over, depth, W, H, ckw, nf = _case()
    clip = make_clip(W, H, nf, bit_depth=depth, **ckw)
    cfg = lib.la_config(W, H, preset, bit_depth=depth, **over)
    be = OracleBackend(cfg, speculative=True)
    ws = shard.WindowShard(OracleShardAdapter(be, clip, own_ingest=rank != 0), dist, rank, world)
    if rank == 0:
        be.on_prefetch = ws.on_prefetch
        la = lib.Lookahead(cfg, backend=be.struct)
        outs = la.run(clip, qp_offsets=True)
        la.close()
        ws.stop()
        q.put(dict(outs=[(o.frame, o.type, np.array(o.cost_est[:]).reshape(18, 18).copy(), np.array(o.cost_est_aq[:]).reshape(18, 18).copy(),
                          o.qp_offset.copy()) for o in outs],
                   stats=ws.stats, spec_used=be.spec_used, searched_here=be.searched_here))
    else:
        ws.serve()
        q.put(dict(rank=rank, stats=ws.stats))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_one_window():
    sys.path.insert(0, ROOT)
    from tests.test_golden import GOLD
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0 = next(g for g in got if "outs" in g)
    r1 = next(g for g in got if "outs" not in g)
    # the single-stream golden run of the reference
    z = np.load(os.path.join(GOLD, "lookahead_%s.npz" % CASE))
    preset, opts, over, depth, W, H, ckw, nf = _case()
    assert [o[0] for o in r0["outs"]] == z["idx"].tolist()
    assert [o[1] for o in r0["outs"]] == z["type"].tolist()
    nb = z["cost"].shape[1]
    for k, o in enumerate(r0["outs"]):
        assert np.array_equal(o[2][:nb, :nb], z["cost"][k]), k
        m = z["cost"][k] >= 0  # cells that were never evaluated keep whatever the reference's recycled frame held
        assert np.array_equal(o[3][:nb, :nb][m], z["cost_aq"][k][m]), k
        assert np.array_equal(o[4], z["qp_offset"][k]), k
    # the work really was shared: rank 1 searched its frames, rank 0 consumed the imported fields
    assert r1["stats"]["fields_searched"] > 0 and r0["stats"]["fields_imported"] == r1["stats"]["fields_searched"]
    assert abs(r0["stats"]["fields_searched"] - r1["stats"]["fields_searched"]) <= 0.25 * r0["stats"]["fields_searched"] + 8
    assert r0["spec_used"] > 0 and r0["spec_used"] >= 4 * r0["searched_here"]
