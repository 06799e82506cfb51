"""The window shard through the library's C entry points (x264_amd/csrc/shard_host.cpp: x264hip_shard_open / _put_frames / _serve /
_status / _close over an x264hip_shard_transport): one lookahead window over several ranks, driven the way a C host would drive it.
* one rank, RCCL transport in loop-back mode: every exchange step (picture broadcast, peer-to-peer fields, summaries gather, status
  all-reduce) runs on ONE GPU over librccl opened by the library itself; results equal the plain run, every buffer that came back
  equals what was sent;
* two ranks sharing the GPU over a host-staged transport (gloo): decisions, cost cells and f_qp_offset equal the single stream's;
* a rank that fails: the other ranks do not hang, the failure reaches every rank (X264HIP_EPEER on the healthy one);
* tests/tools/shard_driver.c: the same calls from plain C with dlopen (what INTEGRATION.md section 7 shows)."""
import os
import socket
import subprocess

import numpy as np
import pytest

from x264_amd import lib
from x264_amd.synth import make_clip

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
W, H, NF = 704, 576, 70
CLIP = dict(seed=12, scene_cuts=(23,), fade=(40, 8, 0.7, 6), pan=(4, 2))
OVER = dict(bframes=8, rc_lookahead=60)  # BASELINE configs[3] options


def _sig(outs):
    return [(o.frame, o.type, [o.cost_est[i][j] for i in range(10) for j in range(10)], o.qp_offset.tobytes()) for o in outs]


def _plain():
    frames = make_clip(W, H, NF, **CLIP)
    cfg = lib.la_config(W, H, "medium", **OVER)
    la = lib.Lookahead(cfg, max_frames=NF + 4)
    try:
        return _sig(la.run(frames, paced=False, qp_offsets=True))
    finally:
        la.close()


def _loopback_worker(q):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    from x264_amd import lib as L2, shard
    torch.cuda.set_device(0)
    L = L2.load()
    cfg = L2.la_config(W, H, "medium", **OVER)
    dev = torch.from_numpy(make_clip(W, H, NF, **CLIP)).cuda()
    t = shard.rccl_transport(L, shard.rccl_unique_id(L), 0, 1, 0, loopback=True)
    outs, dt, st, rc = shard.run_c_window_shard(torch, L2, 0, 1, 0, cfg, dev, t, qp_offsets=True)
    q.put(dict(sig=_sig(outs), stats=st))


def test_c_shard_loopback_over_rccl():
    import torch.multiprocessing as mp
    want = _plain()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_loopback_worker, args=(q,))
    p.start()
    got = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert got["sig"] == want
    st = got["stats"]
    print("C window shard, RCCL loop-back:", st)
    assert st["chunks"] >= 1 and st["loopback_checks"] >= 2 and st["l0_fields_exchanged"] > 0 and st["cells_imported"] > 0


def _two_rank_worker(rank, port, q, env):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    os.environ.update(env)
    import datetime
    import torch
    import torch.distributed as dist
    from x264_amd import lib as L2, shard
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=120))
    cfg = L2.la_config(W, H, "medium", **OVER)
    cfg["_frames"] = NF
    dev = torch.from_numpy(make_clip(W, H, NF, **CLIP)).cuda() if rank == 0 else None
    t = shard.HostStagedTransport(dist, rank, 2)
    try:
        passes = int(os.environ.get("X264HIP_TEST_PASSES", "1"))
        outs, dt, st, rc = shard.run_c_window_shard(torch, L2, rank, 2, 0, cfg, dev, t, qp_offsets=True, passes=passes)
        sig = None if outs is None else _sig(outs) if passes == 1 else [_sig(o) for o in outs]
        q.put(dict(rank=rank, sig=sig, stats=st, rc=rc, calls=dict(t.calls)))
    except L2.X264HipError as e:
        q.put(dict(rank=rank, error=e.code, calls=t.calls))
    dist.barrier()
    dist.destroy_process_group()


def _run_two(env):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, port, q, env)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    return {g["rank"]: g for g in got}


def test_c_shard_two_ranks_on_one_gpu():
    want = _plain()
    got = _run_two({})
    assert got[0]["sig"] == want
    assert got[1]["rc"] == 0
    s0, s1 = got[0]["stats"], got[1]["stats"]
    print("C window shard, two ranks:", s0, s1)
    assert s1["fields_searched"] > 0 and s1["cells_evaluated"] > 0 and s0["cells_imported"] > 0 and s0["maps_fetched"] > 0
    assert s1["bytes_input_broadcast"] == NF * W * H  # every picture reached rank 1 exactly once
    # some of the maps MB-tree read were the SPARE half of a cell evaluated both ways on its owner rank (the caller had asked for the variant
    # without the list-1 reference's vectors): the 7-word FETCH record carried the spare flag across ranks
    assert s0["maps_fetched_spare"] > 0, s0


def test_c_shard_pieces_and_reset():
    """Every exchange step in several pieces (64 KiB per exchange buffer: one picture, two fields per pair, three maps per piece), and the
    same clip a second time after x264hip_shard_reset: both passes equal the single stream's."""
    want = _plain()
    got = _run_two({"X264HIP_SHARD_PIECE_BYTES": "65536", "X264HIP_TEST_PASSES": "2"})
    assert got[1]["rc"] == 0
    assert len(got[0]["sig"]) == 2 and got[0]["sig"][0] == want and got[0]["sig"][1] == want
    s0, s1 = got[0]["stats"], got[1]["stats"]
    print("C window shard, pieces + reset:", s0, s1, got[0]["calls"])
    assert s1["bytes_input_broadcast"] == 2 * NF * W * H
    assert got[0]["calls"]["broadcast"] >= 2 * NF  # a picture per piece (+ the commands)


def _both_protocols_worker(rank, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    import datetime
    import torch
    import torch.distributed as dist
    from x264_amd import lib as L2, shard
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=120))
    cfg = L2.la_config(W, H, "medium", **OVER)
    cfg["_frames"] = NF
    clip = torch.from_numpy(make_clip(W, H, NF, **CLIP)).cuda()
    # the C shard (x264hip_shard_*), its command blocks decoded on rank 0 as they pass through the transport
    t = shard.HostStagedTransport(dist, rank, 2)
    t.command_log = []
    outs_c, _, st_c, rc = shard.run_c_window_shard(torch, L2, rank, 2, 0, cfg, clip if rank == 0 else None, t, qp_offsets=True)
    dist.barrier()
    # the Python orchestration (shard.WindowShard over the same context entry points), its commands as _send builds them
    log_py = []
    dev = clip.clone()
    if rank:
        dev.zero_()
    outs_p, _, st_p = shard.run_window_shard(torch, L2, dist, rank, 2, 0, cfg, dev, exchange_on_device=False, qp_offsets=True, broadcast_input=True,
                                             command_log=log_py)
    q.put(dict(rank=rank, c=t.command_log, py=log_py, sig_c=None if outs_c is None else _sig(outs_c), sig_py=None if outs_p is None else _sig(outs_p),
               st_c=st_c, st_py=st_p, rc=rc))
    dist.barrier()
    dist.destroy_process_group()


def test_python_and_c_shard_send_the_same_commands():
    """x264_amd/shard.py keeps a Python orchestration of the window shard (what the CPU suite runs over the oracle backend) beside the C one
    in shard_host.cpp (the product).  One window through both, two ranks each: every command rank 0 sends -- CHUNK (slots, frame numbers,
    the class statement), FETCH (the maps MB-tree is about to read, spare halves flagged), STOP -- agrees word for word, and so do the
    counts of what each rank then did (fields searched, cells evaluated, list-0 fields exchanged, maps fetched) and the results."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_both_protocols_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {g["rank"]: g for g in (q.get(timeout=600) for _ in range(2))}
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    g0, g1 = got[0], got[1]
    assert g0["sig_c"] == g0["sig_py"] == _plain()
    c, py = g0["c"], g0["py"]
    kinds = lambda log: [w[0] for w in log]  # noqa: E731
    print("commands: C %d, Python %d; kinds %s" % (len(c), len(py), sorted(set(kinds(c)))))
    assert len(c) > 3 and 1 in kinds(c) and 2 in kinds(c) and kinds(c)[-1] == 0  # CHUNK, FETCH, ..., STOP
    assert len(c) == len(py)
    for k, (a, b) in enumerate(zip(c, py)):
        assert a == b, (k, a[:8], b[:8])
    assert g1["c"] == [] and g1["py"] == []  # only rank 0 commands
    for r in (0, 1):
        for name in ("chunks", "fields_searched", "cells_evaluated", "l0_fields_exchanged", "cells_imported", "maps_fetched", "fetch_commands",
                     "bytes_input_broadcast", "bytes_l0_received", "bytes_summaries", "bytes_maps"):
            assert got[r]["st_c"][name] == got[r]["st_py"][name], (r, name, got[r]["st_c"][name], got[r]["st_py"][name])


def test_c_shard_failing_rank_fails_everybody():
    """rank 1 fails at its first chunk: it keeps the collectives going, the status word carries the failure, rank 0's calls fail with
    X264HIP_EPEER (-7) -- nobody is left inside a collective (the test would time out)"""
    got = _run_two({"X264HIP_SHARD_TEST_FAIL": "1:0"})
    assert got[1].get("rc", got[1].get("error")) == -4      # its own error: X264HIP_EDEVICE
    assert got[0].get("error") in (-7, -4), got[0]           # X264HIP_EPEER through the status word (or the hook's own failure code)


def test_c_shard_from_plain_c():
    """tests/tools/shard_driver.c: dlopen + the C entry points only (HIP runtime calls through dlsym as well), world = 1 over RCCL in
    loop-back mode; the program prints one line per frame, compared with the plain run"""
    src, exe = os.path.join(HERE, "tools", "shard_driver.c"), os.path.join(HERE, "tools", "_build", "shard_driver")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", exe, src, "-ldl"])
    Wc, Hc, nf = 352, 288, 40
    frames = make_clip(Wc, Hc, nf, seed=3, scene_cuts=(17,))
    cfg = lib.la_config(Wc, Hc, "medium")
    la = lib.Lookahead(cfg, max_frames=nf + 4)
    try:
        ref = la.run(frames, paced=False)
    finally:
        la.close()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        frames.tofile(os.path.join(d, "frames.bin"))
        params = lib.make_la_params(cfg, None, nf + 4)
        import ctypes as C
        open(os.path.join(d, "params.bin"), "wb").write(bytes(C.string_at(C.addressof(params), C.sizeof(params))))
        # (the cost_mv pointer inside the blob is meaningless in another process: the driver points it at its own copy of the table)
        cm, centre = lib.cost_mv_table(cfg["mv_range"], cfg["lam"])
        np.ascontiguousarray(cm, np.uint16).tofile(os.path.join(d, "cost_mv.bin"))
        out = subprocess.check_output([exe, os.path.join(ROOT, "x264_amd", "libx264hip.so"), d, str(Wc), str(Hc), str(nf), str(C.sizeof(params))],
                                      timeout=300).decode()
    rows = [tuple(int(v) for v in ln.split()[1:]) for ln in out.splitlines() if ln.startswith("frame ")]
    assert rows == [(o.frame, o.type, o.cost_est[0][0]) for o in ref], out[-2000:]
    assert "loopback checks" in out
