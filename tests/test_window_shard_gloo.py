"""SURVEY 8(e), the primary design: ONE stream, one lookahead window, the frames dealt round-robin to the ranks.  Two processes
(gloo, CPU): rank b % 2 runs the unweighted motion searches of frame b AND its cost cells; the only fields exchanged are the list-0
fields B cells read from their list-1 reference; rank 0, which decides and runs MB-tree, receives cell SUMMARIES and fetches the
per-block maps MB-tree reads.  The run must give the slice types, every cost cell and the quantiser offsets of the SINGLE-STREAM run
(a committed golden fixture generated from the reference): no IDR is forced at any rank boundary, unlike the GOP-segment form of
tests/test_multi_rank_gloo.py.  The oracle backend stands in for the device (tests/oracle_backend.py); the GPU form of the same
protocol is x264_amd.shard.HipAdapter."""
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
    be.verify_imported = True  # every summary taken from the other rank is checked against an evaluation of the cell here
    ws = shard.WindowShard(OracleShardAdapter(be, clip, own_ingest=rank != 0, dist=dist, rank=rank, world=world), dist, rank, world)
    if rank == 0:
        be.on_prefetch = ws.on_prefetch
        be.on_mbtree = ws.before_mbtree
        la = lib.Lookahead(cfg, backend=be.struct)
        outs = la.run(clip, qp_offsets=True)
        la.close()
        ws.stop()
        q.put(dict(outs=[(o.frame, o.type, np.array(o.cost_est[:]).reshape(18, 18).copy(), np.array(o.cost_est_aq[:]).reshape(18, 18).copy(),
                          o.qp_offset.copy()) for o in outs],
                   stats=ws.stats, spec_used=be.spec_used, searched_here=be.searched_here, cells_from_owner=be.cells_from_owner, cells_here=be.cells_here,
                   remote_fields_searched_here=be.remote_fields_searched_here, maps_recomputed_here=be.maps_recomputed_here))
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
    # the work really was shared: rank 1 searched its frames and evaluated their cells; rank 0 answered those cells from the
    # summaries, searched (almost) nothing of rank 1's frames itself and fetched the maps MB-tree read instead of recomputing them
    s0, s1 = r0["stats"], r1["stats"]
    assert s1["fields_searched"] > 0 and s1["cells_evaluated"] > 0 and s0["cells_imported"] == s1["cells_evaluated"]
    assert abs(s0["fields_searched"] - s1["fields_searched"]) <= 0.25 * s0["fields_searched"] + 8
    assert r0["cells_from_owner"] > 0 and s0["maps_fetched"] > 0
    assert r0["maps_recomputed_here"] == 0
    assert r0["remote_fields_searched_here"] <= 0.1 * s1["fields_searched"], (r0["remote_fields_searched_here"], s1["fields_searched"])
    # only summaries, the list-0 fields of list-1 references and the fetched maps crossed ranks
    n_mb = ((W + 15) // 16) * ((H + 15) // 16)
    assert s0["bytes_summaries"] + s0["bytes_l0_exchange"] + s0["bytes_maps"] < 8 * n_mb * (s0["fields_searched"] + s1["fields_searched"])
    print("window shard stats", s0, {k: r0[k] for k in ("cells_from_owner", "cells_here", "remote_fields_searched_here", "spec_used", "searched_here")})
