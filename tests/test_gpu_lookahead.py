"""GPU end-to-end: the full lookahead (HIP evaluations + host decisions) against the golden fixtures
generated from the reference, against the real reference build where it travelled (oracle/_ref), and
through size-independent properties at the BASELINE sizes."""
import os

import numpy as np
import pytest

from oracle import refharness
from tests.golden.make_golden import FULL_SIZE_CASES, LOOKAHEAD_CASES
from tests.test_golden import GOLD, check_lookahead_outputs
from x264_amd import lib
from x264_amd.synth import make_clip

pytestmark = pytest.mark.gpu


def _types(outs):
    return [(o.frame, o.type) for o in outs]


def _mats(outs, nb):
    return [np.array([[o.cost_est[i][j] for j in range(nb)] for i in range(nb)]) for o in outs]


@pytest.mark.parametrize("name", list(LOOKAHEAD_CASES))
@pytest.mark.parametrize("paced", [True, False])
def test_lookahead_vs_golden(name, paced):
    preset, opts, over, depth, W, H, ckw, nf = LOOKAHEAD_CASES[name]
    z = np.load(os.path.join(GOLD, "lookahead_%s.npz" % name))
    frames = make_clip(W, H, nf, bit_depth=depth, **ckw)
    cfg = lib.la_config(W, H, preset, bit_depth=depth, **over)
    la = lib.Lookahead(cfg, max_frames=0 if paced else nf + 4)
    try:
        outs = la.run(frames, paced=paced, qp_offsets=True)
        st = la.stats()
    finally:
        la.close()
    check_lookahead_outputs(outs, z, cfg["bframes"] + 2)
    assert st[1] > nf  # real evaluations happened on the device


@pytest.mark.skipif(not refharness.available(8) or not refharness.available(10), reason="oracle/_ref did not travel")
@pytest.mark.parametrize("W,H,depth,preset,opts,over,nf", [
    (1920, 1080, 8, "slow", "me=dia", dict(me="dia"), 56),          # BASELINE configs[1]
    (3840, 2160, 8, "slower", "me=umh,merange=32", dict(me="umh", me_range=32), 24),  # BASELINE configs[2], shortened
    # BASELINE configs[4] (8K 10-bit veryslow + tesa: HEX range 24, fpelcmp = SATD, bframes 8, b-adapt 2) at a quarter of
    # the area and a dozen frames, so that the reference C path finishes in seconds
    (3840, 2160, 10, "veryslow", "me=tesa", dict(me="tesa"), 14),
])
def test_full_size_vs_reference(W, H, depth, preset, opts, over, nf):
    frames = make_clip(W, H, nf, seed=21, bit_depth=depth, scene_cuts=(nf // 2,), pan=(5, 3))
    r = refharness.Ref(W, H, preset, opts=opts, bit_depth=depth)
    try:
        ref = r.lookahead_run(frames, with_qp_offsets=True)
    finally:
        r.close()
    cfg = lib.la_config(W, H, preset, bit_depth=depth, **over)
    la = lib.Lookahead(cfg)
    try:
        outs = la.run(frames, qp_offsets=True)
    finally:
        la.close()
    nb = cfg["bframes"] + 2
    z = dict(idx=ref["idx"], type=ref["type"], cost=ref["cost"][:, :nb, :nb], cost_aq=ref["cost_aq"][:, :nb, :nb],
             qp_offset=ref["qp_offset"])
    check_lookahead_outputs(outs, z, nb)


@pytest.mark.parametrize("name", list(FULL_SIZE_CASES))
def test_baseline_configs_as_written(name):
    """BASELINE configs[3] (3840x2160, ONE 250-frame GOP, --rc-lookahead 60 --bframes 8) and configs[4] (7680x4320 10-bit, --preset
    veryslow --me tesa, 72 frames: the 60-frame window fills and slides, b-adapt 2 trellis over a full window) at their full size against
    fixtures the real reference produced in the build container (tests/golden/make_golden.py --full-size): coded order, slice types,
    every cost cell, and a CRC of every frame's f_qp_offset -- encoder-paced and with every frame queued before the first decision."""
    import zlib
    from x264_amd.synth import upscaled_clip
    preset, opts, over, depth, W, H, ckw, nf = FULL_SIZE_CASES[name]
    z = np.load(os.path.join(GOLD, "fullsize_%s.npz" % name))
    frames = upscaled_clip(W, H, nf, depth, **ckw)
    for k in (0, nf // 2, nf - 1):  # the clip is the one the reference saw
        assert zlib.crc32(frames[k].tobytes()) == int(z["frame_crc"][k]), "clip differs from the fixture's (numpy version?)"
    cfg = lib.la_config(W, H, preset, bit_depth=depth, **over)
    gold_cfg = {str(k): int(v) for k, v in zip(z["cfg_keys"], z["cfg"])}
    assert (cfg["bframes"], cfg["b_adapt"], cfg["rc_lookahead"]) == (gold_cfg["bframes"], gold_cfg["b_adapt"], gold_cfg["rc_lookahead"])
    for paced in (True, False):
        la = lib.Lookahead(cfg, max_frames=0 if paced else nf + 4)
        try:
            outs = la.run(frames, paced=paced, qp_offsets=True)
        finally:
            la.close()
        check_lookahead_outputs(outs, z, cfg["bframes"] + 2)


@pytest.mark.parametrize("W,H,preset,over,nf", [(1920, 1080, "slow", dict(me="dia"), 70), (3840, 2160, "medium", dict(bframes=8, rc_lookahead=60), 30)])
def test_batching_invariance_full_size(W, H, preset, over, nf):
    """Size-independent property at BASELINE sizes: encoder-paced (small batches, on-demand evaluations) and
    deep-prefetch (one big speculative batch) runs give identical decisions and cost cells."""
    frames = make_clip(W, H, nf, seed=33, scene_cuts=(nf // 3,), fade=(nf // 2, 6, 0.7, 8))
    cfg = lib.la_config(W, H, preset, **over)
    nb = cfg["bframes"] + 2
    res = []
    for paced in (True, False):
        la = lib.Lookahead(cfg, max_frames=0 if paced else nf + 4)
        try:
            outs = la.run(frames, paced=paced)
        finally:
            la.close()
        res.append(outs)
    assert _types(res[0]) == _types(res[1])
    for a, b in zip(_mats(res[0], nb), _mats(res[1], nb)):
        assert np.array_equal(a, b)
    types = [t for _, t in _types(res[0])]
    assert types[0] == 1 and set(types) <= {1, 2, 3, 4, 5}
    assert sorted(f for f, _ in _types(res[0])) == list(range(nf))


def test_batching_invariance_8k_10bit():
    """BASELINE configs[4] (7680x4320 10-bit, --preset veryslow --me tesa): encoder-paced and deep-prefetch runs of a dozen frames
    give identical decisions, cost cells and quantiser offsets."""
    from tests.test_gpu_parity import upscaled_clip
    W, H, nf = 7680, 4320, 12
    frames = upscaled_clip(W, H, nf, 10, seed=44, scene_cuts=(7,), pan=(5, 3))
    cfg = lib.la_config(W, H, "veryslow", bit_depth=10, me="tesa")
    assert (cfg["bframes"], cfg["b_adapt"], cfg["fpelcmp_satd"], cfg["me_range"]) == (8, 2, 1, 24)
    nb = cfg["bframes"] + 2
    res = []
    for paced in (True, False):
        la = lib.Lookahead(cfg, max_frames=nf + 12)
        try:
            res.append(la.run(frames, paced=paced, qp_offsets=True))
        finally:
            la.close()
    assert _types(res[0]) == _types(res[1])
    for a, b in zip(_mats(res[0], nb), _mats(res[1], nb)):
        assert np.array_equal(a, b)
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a.qp_offset, b.qp_offset)
    assert sorted(f for f, _ in _types(res[0])) == list(range(nf))


def test_batch_ingest_device_pointers():
    """x264hip_lookahead_put_frames (device-resident frames, batched ingest) gives the same decisions and maps as
    frame-by-frame host ingest."""
    import torch
    W, H, nf = 352, 288, 40
    frames = make_clip(W, H, nf, seed=3, scene_cuts=(17,))
    cfg = lib.la_config(W, H, "medium")
    la = lib.Lookahead(cfg)
    try:
        ref = la.run(frames, qp_offsets=True)
    finally:
        la.close()
    dev = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    la = lib.Lookahead(cfg, max_frames=nf + 4)
    try:
        outs = la.run(device_ptrs=[dev[i].data_ptr() for i in range(nf)], stride=W, paced=False, qp_offsets=True)
    finally:
        la.close()
    assert _types(outs) == _types(ref)
    nb = cfg["bframes"] + 2
    for a, b in zip(_mats(outs, nb), _mats(ref, nb)):
        assert np.array_equal(a, b)
    for a, b in zip(outs, ref):
        assert np.array_equal(a.qp_offset, b.qp_offset)


@pytest.mark.parametrize("paced", [False, True])
def test_run_frames_is_the_put_get_sequence(paced):
    """x264hip_lookahead_run_frames (a whole clip of device-resident frames in one call) against the put / get calls it stands for."""
    import torch
    W, H, nf = 352, 288, 40
    frames = make_clip(W, H, nf, seed=5, scene_cuts=(21,))
    cfg = lib.la_config(W, H, "medium")
    dev = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    ptrs = [dev[i].data_ptr() for i in range(nf)]
    res = []
    for one_call in (False, True):
        la = lib.Lookahead(cfg, max_frames=nf + 4)
        try:
            res.append(la.run_frames(ptrs, stride=W, paced=paced) if one_call else la.run(device_ptrs=ptrs, stride=W, paced=paced))
        finally:
            la.close()
    assert len(res[1]) == nf and _types(res[0]) == _types(res[1])
    nb = cfg["bframes"] + 2
    for a, b in zip(_mats(res[0], nb), _mats(res[1], nb)):
        assert np.array_equal(a, b)


def _shard_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from x264_amd import lib as L2, shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, nf = 704, 576, 70
    frames = make_clip(W, H, nf, seed=12, scene_cuts=(23,), fade=(40, 8, 0.7, 6), pan=(4, 2))
    cfg = L2.la_config(W, H, "medium", bframes=8, rc_lookahead=60)
    dev = torch.from_numpy(frames).cuda()
    if rank:
        dev.zero_()  # the pictures arrive chunk by chunk from rank 0 (broadcast_input)
    outs, dt, stats = shard.run_window_shard(torch, L2, dist, rank, world, 0, cfg, dev, exchange_on_device=False, qp_offsets=True, vbv=True,
                                             broadcast_input=True)
    if rank == 0:
        q.put(dict(sig=[(o.frame, o.type, [o.cost_est[i][j] for i in range(10) for j in range(10)], o.qp_offset.tobytes()) for o in outs],
                   rows=[(o.row_satds.tobytes(), o.row_satds_intra.tobytes()) for o in outs], stats=stats))
    else:
        q.put(dict(stats=stats))
    dist.barrier()
    dist.destroy_process_group()


def _kernel_form_worker(env, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(env)
    from x264_amd import lib as L2
    W, H, nf = 704, 576, 48
    frames = make_clip(W, H, nf, seed=21, scene_cuts=(19,), fade=(30, 6, 0.7, 5), pan=(7, 3))
    cfg = L2.la_config(W, H, "slow", me="dia")
    la = L2.Lookahead(cfg, max_frames=nf + 4)
    try:
        outs = la.run(frames, paced=False, qp_offsets=True)
    finally:
        la.close()
    q.put([(o.frame, o.type, [o.cost_est[i][j] for i in range(5) for j in range(5)], o.qp_offset.tobytes()) for o in outs])


def test_every_form_of_the_search_kernel_gives_the_same_lookahead():
    """The two forms of the whole-frame motion search -- me_rows_kernel (strips through the L1) and the latency form out of LDS
    (me_latency.h) -- behind the same lookahead: types, cost cells and f_qp_offset must not depend on which of them a launch was sent to.  The switches are read once per process, so every form runs in a process of its own."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    got = {}
    for name, env in (("default", {}), ("rows", {"X264HIP_SEARCH": "rows"}),
                      ("latency", {"X264HIP_LAT_WAVES": "1000000"}), ("split ingest", {"X264HIP_INGEST": "split"}),
                      ("mbtree lists in LDS", {"X264HIP_MBT": "lds"}), ("mbtree per level", {"X264HIP_MBT": "levels"}),
                      ("mbtree in small workgroups", {"X264HIP_MBT_THREADS": "256", "X264HIP_MBT_WGS": "3"})):
        q = ctx.Queue()
        p = ctx.Process(target=_kernel_form_worker, args=(env, q))
        p.start()
        got[name] = q.get(timeout=600)
        p.join(timeout=120)
        assert p.exitcode == 0, name
    # split ingest: planes + strip copy from two kernels instead of lowres_tiles_kernel; the MB-tree forms: every queued list on one workgroup
    # with its accumulators in LDS / a launch per level instead of counter barriers (f_qp_offset is part of what is compared)
    for name in ("rows", "latency", "split ingest", "mbtree lists in LDS", "mbtree per level", "mbtree in small workgroups"):
        assert got[name] == got["default"], name


def _loopback_worker(port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    import torch
    import torch.distributed as dist
    from x264_amd import lib as L2, shard
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))  # RCCL, one rank: the collectives loop back
    W, H, nf = 704, 576, 70
    frames = make_clip(W, H, nf, seed=12, scene_cuts=(23,), fade=(40, 8, 0.7, 6), pan=(4, 2))
    cfg = L2.la_config(W, H, "medium", bframes=8, rc_lookahead=60)
    dev = torch.from_numpy(frames).cuda()
    outs, dt, stats = shard.run_window_shard(torch, L2, dist, 0, 1, 0, cfg, dev, exchange_on_device=True, qp_offsets=True, loopback=True)
    q.put(dict(sig=[(o.frame, o.type, [o.cost_est[i][j] for i in range(10) for j in range(10)], o.qp_offset.tobytes()) for o in outs], stats=stats))
    dist.destroy_process_group()


def test_window_shard_loopback_on_device():
    """The exchange code path of the window shard as the 8-GPU run will execute it -- Exchange(on_device=True): export kernels, RCCL
    collectives (broadcast, all_to_all_single, gather) issued under the context's own HIP stream as a torch ExternalStream, imports --
    on ONE GPU: a one-rank RCCL process group loops every collective back to the sender.  Results must equal the plain run, and every
    buffer that came back must equal what the export kernels wrote (the collectives really ran behind them on the stream)."""
    import socket
    import torch.multiprocessing as mp
    W, H, nf = 704, 576, 70
    frames = make_clip(W, H, nf, seed=12, scene_cuts=(23,), fade=(40, 8, 0.7, 6), pan=(4, 2))
    cfg = lib.la_config(W, H, "medium", bframes=8, rc_lookahead=60)
    la = lib.Lookahead(cfg, max_frames=nf + 4)
    try:
        ref = la.run(frames, paced=False, qp_offsets=True)
    finally:
        la.close()
    want = [(o.frame, o.type, [o.cost_est[i][j] for i in range(10) for j in range(10)], o.qp_offset.tobytes()) for o in ref]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_loopback_worker, args=(port, q))
    p.start()
    got = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert got["sig"] == want
    st = got["stats"]
    print("window shard loop-back:", st)
    assert st["chunks"] >= 1 and st["loopback_checks"] >= 2 and st["l0_fields_exchanged"] > 0 and st["cells_imported"] > 0 and st["bytes_l0_sent"] > 0


def test_window_shard_two_ranks_on_one_gpu():
    """SURVEY 8(e) on the device: two processes share one lookahead window (here also one GPU, gloo carrying the exchange): rank 1
    searches the odd frames AND evaluates their cost cells; rank 0 receives cell summaries, fetches the per-block maps MB-tree reads,
    decides and runs MB-tree.  Decisions, cost cells and f_qp_offset must equal the plain single-context run (BASELINE configs[3]
    options: --bframes 8 --rc-lookahead 60), and rank 0's share of the device work has to be the small one."""
    import socket
    import torch.multiprocessing as mp
    W, H, nf = 704, 576, 70
    frames = make_clip(W, H, nf, seed=12, scene_cuts=(23,), fade=(40, 8, 0.7, 6), pan=(4, 2))
    cfg = lib.la_config(W, H, "medium", bframes=8, rc_lookahead=60)
    la = lib.Lookahead(cfg, max_frames=nf + 4)
    try:
        ref = la.run(frames, paced=False, qp_offsets=True, vbv=True)
    finally:
        la.close()
    want = [(o.frame, o.type, [o.cost_est[i][j] for i in range(10) for j in range(10)], o.qp_offset.tobytes()) for o in ref]
    want_rows = [(o.row_satds.tobytes(), o.row_satds_intra.tobytes()) for o in ref]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0 = next(g for g in got if "sig" in g)
    r1 = next(g for g in got if "sig" not in g)
    assert r0["sig"] == want
    # what VBV rate control reads per frame -- the row sums of the cell the frame is coded with AND its intra row sums -- for the frames
    # another rank owns as well (an imported cell summary must not touch the intra rows rank 0 computed itself)
    assert r0["rows"] == want_rows
    s0, s1 = r0["stats"], r1["stats"]
    print("window shard on one GPU:", s0)
    assert s1["fields_searched"] > 100 and s1["cells_evaluated"] > 100 and s0["cells_imported"] == s1["cells_evaluated"]
    # (the row sums asked for above come through the getter that also serves the per-block map: a cell whose map was never fetched is
    # evaluated locally for it -- a handful at most)
    assert s0["maps_fetched"] > 0 and s0["remote_maps_recomputed_here"] <= 2
    # what rank 0 searched of the other rank's frames (cells evaluated on demand after all) stays a small part of that rank's searches
    assert s0["remote_fields_searched_here"] <= 0.2 * s1["fields_searched"], (s0["remote_fields_searched_here"], s1["fields_searched"])
    # rank 0's own share of the two big kernels: about half of the fields and cells (two ranks), not all of them
    assert s0["searches_here"] <= 0.65 * (s0["fields_searched"] + s1["fields_searched"]) + s0["remote_fields_searched_here"] + 40
    n_mb = ((W + 15) // 16) * ((H + 15) // 16)
    assert s0["bytes_summaries"] + s0["bytes_l0_exchange"] + s0["bytes_maps"] < 8 * n_mb * (s0["fields_searched"] + s1["fields_searched"])
