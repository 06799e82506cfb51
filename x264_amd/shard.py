"""Sharding of the lookahead across the GPUs of one node, one process per GPU (torch.distributed: backend "nccl" = RCCL over xGMI,
gloo on CPU for the tests).  Two forms:

* WindowShard -- ONE stream, one lookahead window, the frames of the window dealt round-robin to the ranks (BASELINE
  configs[3]): rank b % world runs the motion searches of frame b, the finished fields (mv + mv cost per block) are gathered to
  rank 0, which takes the decisions, evaluates the cost cells and runs MB-tree exactly as a single-GPU run would.  Slice types and
  every cost cell are those of the single-stream run (no IDR is forced anywhere).
* GOP segments (segment_bounds / summarize / gather_summaries) -- independent closed-GOP segments per rank, the only exchange an
  all-gather of per-frame summaries: the zero-communication form, used for throughput scaling over independent streams."""
import numpy as np


def segment_bounds(n_frames, rank, world):
    """Contiguous, near-equal display-order segments; every segment starts a new (closed) GOP."""
    base, rem = divmod(n_frames, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


_BMASK = np.zeros((18, 18), bool)
for _i in range(1, 18):
    for _j in range(1, 18 - _i):
        _BMASK[_i, _j] = True


def _bcell_max(ce):
    return ce[_BMASK].max()


def summarize(outs, frame_offset=0):
    """[n,4] int32 rows (display frame number, slice type, cost of the chosen cell, bframes) from lookahead outputs."""
    rows = np.zeros((len(outs), 4), np.int32)
    for k, o in enumerate(outs):
        ce = np.frombuffer(o.cost_est, dtype=np.int32).reshape(18, 18)  # the ctypes array, viewed in place
        if o.type < 3:
            cost = ce[0, 0]
        elif o.type == 3:
            cost = max(int(ce[1:, 0].max()), 0)
        else:
            cost = max(int(_bcell_max(ce)), 0)
        rows[k] = (o.frame + frame_offset, o.type, cost, o.bframes)
    return rows


def gather_summaries(summary, dist, device=None):
    """all_gather of equally sized [n,4] summaries; returns the [world*n,4] array on every rank."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(summary, np.int32))
    if device is not None:
        t = t.to(device)
    world = dist.get_world_size()
    out = torch.zeros((world * t.shape[0], 4), dtype=torch.int32, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy()


# ---------------------------------------------------------------------------------------------------------------------------
# One window over several ranks
# ---------------------------------------------------------------------------------------------------------------------------
CMD_STOP, CMD_CHUNK = 0, 1


class WindowShard:
    """Protocol shared by all ranks.  Rank 0 runs the host lookahead; whenever it would submit speculative work for a chunk of
    frames (the host logic's prefetch call: every resident frame the next decisions can reach) it broadcasts the chunk, every rank
    searches the fields of the frames it owns, and a gather brings them to rank 0.  The other ranks sit in serve().

    adapter (duck-typed; HipAdapter below, an oracle-backed one in the tests):
        n_mb, bframes                                 geometry
        ingest(slot, frame_number)                    make the frame resident in `slot` (no-op if it already is)
        classes() -> (mask_l0, mask_l1)               rank 0: the (list, distance) classes worth speculating
        search(reqs)                                  reqs: list of (slot_b, slot_ref, list, dist_m1): unweighted searches, asynchronous
        export(keys) -> tensor [len(keys), n_mb, 2]   int32 {mv, cost} of the fields (slot, list, dist_m1), on the exchange device
        import_(keys, tensor)                         rank 0: take fields searched elsewhere
        finish(slots, numbers)                        rank 0: speculative cost cells over the fields now present
    """

    def __init__(self, adapter, dist, rank, world, device=None):
        self.a, self.dist, self.rank, self.world, self.device = adapter, dist, rank, world, device
        self.done = set()       # (frame_number, list, dist_m1) fields already planned in an earlier chunk (same on every rank)
        self.resident = {}      # slot -> frame number, as announced by rank 0
        self.stats = dict(chunks=0, fields_searched=0, fields_imported=0, bytes_gathered=0)

    # ---- the plan of a chunk: identical on every rank ----
    def plan(self, slots, numbers, masks):
        by_owner = [[] for _ in range(self.world)]
        bf = self.a.bframes
        for i, ni in enumerate(numbers):
            for j, nj in enumerate(numbers):
                d = nj - ni
                if d == 0 or abs(d) > bf + 1 or (d > 0 and not bf):
                    continue
                lst, dm1 = int(d > 0), abs(d) - 1
                if not (masks[lst] >> dm1) & 1 or (ni, lst, dm1) in self.done:
                    continue
                self.done.add((ni, lst, dm1))
                by_owner[ni % self.world].append((slots[i], slots[j], lst, dm1))
        return by_owner

    def _bcast(self, t):
        import torch
        t = t.to(self.device) if self.device is not None else t
        self.dist.broadcast(t, src=0)
        return t.cpu()

    def _run_chunk(self, slots, numbers, masks):
        import torch
        for s, n in zip(slots, numbers):
            if self.resident.get(s) != n:
                self.resident[s] = n  # (a frame number never comes back in another slot within a stream)
                self.a.ingest(s, n)
        by_owner = self.plan(slots, numbers, masks)
        mine = by_owner[self.rank]
        self.a.search(mine)
        self.stats["chunks"] += 1
        self.stats["fields_searched"] += len(mine)
        width = max(len(x) for x in by_owner)
        if self.world == 1 or width == 0:
            return
        # one collective of equally sized buffers: [width, n_mb, 2] int32 per rank, padded.  all_gather rather than gather: rank 0 is
        # the only consumer, but its ingress is the bottleneck either way and all_gather is a collective every
        # backend implements natively (RCCL ring over xGMI; gloo in the tests)
        buf = torch.zeros((width, self.a.n_mb, 2), dtype=torch.int32, device=self.device if self.device is not None else "cpu")
        if mine:
            buf[:len(mine)] = self.a.export([(r[0], r[2], r[3]) for r in mine])
        out = [torch.empty_like(buf) for _ in range(self.world)]
        self.dist.all_gather(out, buf)
        if self.rank == 0:
            for r in range(1, self.world):
                if by_owner[r]:
                    self.a.import_([(q[0], q[2], q[3]) for q in by_owner[r]], out[r][:len(by_owner[r])])
                    self.stats["fields_imported"] += len(by_owner[r])
            self.stats["bytes_gathered"] += (self.world - 1) * buf.numel() * 4

    # ---- rank 0 ----
    def on_prefetch(self, slots, numbers):
        """the host logic's speculative submission on rank 0 (x264hip_prefetch_hook / the backend's prefetch entry)"""
        import torch
        m0, m1 = self.a.classes()
        n = len(slots)
        cmd = torch.tensor([CMD_CHUNK, n, m0, m1] + list(slots) + list(numbers), dtype=torch.int64)
        if self.world > 1:
            self._bcast(torch.tensor([cmd.numel()], dtype=torch.int64))
            self._bcast(cmd)
        self._run_chunk(list(slots), list(numbers), (m0, m1))
        self.a.finish(list(slots), list(numbers))

    def stop(self):
        import torch
        if self.world > 1:
            self._bcast(torch.tensor([1], dtype=torch.int64))
            self._bcast(torch.tensor([CMD_STOP], dtype=torch.int64))

    # ---- ranks 1 .. world-1 ----
    def serve(self):
        import torch
        while True:
            ln = int(self._bcast(torch.zeros(1, dtype=torch.int64))[0])
            cmd = self._bcast(torch.zeros(ln, dtype=torch.int64)).tolist()
            if cmd[0] == CMD_STOP:
                return
            n = cmd[1]
            self._run_chunk(cmd[4:4 + n], cmd[4 + n:4 + 2 * n], (cmd[2], cmd[3]))


class HipAdapter:
    """WindowShard over a device context (x264hip_ctx) of this rank's GPU.  frames: callable frame_number -> (device pointer, stride)
    of the full-resolution luma on THIS device (every rank needs the reference frames of the searches it runs: in the benchmark
    the clip is resident on every GPU; a caller with one input copy would all-gather the lowres planes instead)."""

    def __init__(self, L, ctx_handle, cfg, frames, torch_device):
        import ctypes as C
        self.C, self.L, self.h, self.cfg, self.frames, self.dev = C, L, ctx_handle, cfg, frames, torch_device
        mb_w, mb_h = (cfg["width"] + 15) // 16, (cfg["height"] + 15) // 16
        self.n_mb, self.bframes = mb_w * mb_h, cfg["bframes"]
        self.own_ingest = False  # rank 0's frames are ingested by its host lookahead (put_frame)
        L.x264hip_frame_put.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.x264hip_export_field.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.x264hip_import_field.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]

    def _ck(self, rc, what):
        from . import lib
        lib._ck(rc, what)

    def ingest(self, slot, number):
        if not self.own_ingest:
            return
        ptr, stride = self.frames(number)
        self._ck(self.L.x264hip_frame_put(self.h, slot, self.C.c_void_p(ptr), stride, 1, None, None, 0, None), "frame_put")

    def classes(self):
        a, b = self.C.c_uint(), self.C.c_uint()
        self._ck(self.L.x264hip_field_classes(self.h, self.C.byref(a), self.C.byref(b)), "field_classes")
        return a.value, b.value

    def search(self, reqs):
        if not reqs:
            return
        n = len(reqs)
        arr = lambda k: (self.C.c_int * n)(*[r[k] for r in reqs])  # noqa: E731
        self._ck(self.L.x264hip_search_fields(self.h, n, arr(0), arr(1), arr(2), arr(3)), "search_fields")

    def export(self, keys):
        import torch
        out = torch.empty((len(keys), self.n_mb, 2), dtype=torch.int32, device=self.dev)
        for i, (slot, lst, dm1) in enumerate(keys):
            self._ck(self.L.x264hip_export_field(self.h, slot, lst, dm1, self.C.c_void_p(out[i].data_ptr())), "export_field")
        self._ck(self.L.x264hip_synchronize(self.h), "synchronize")  # the collective runs on torch's stream
        return out

    def import_(self, keys, t):
        import torch
        t = t.to(self.dev).contiguous()  # (a CPU tensor when the exchange ran over gloo)
        torch.cuda.synchronize()  # the gathered data was produced on torch's stream; the context reads it on its own
        for i, (slot, lst, dm1) in enumerate(keys):
            self._ck(self.L.x264hip_import_field(self.h, slot, lst, dm1, self.C.c_void_p(t[i].data_ptr())), "import_field")
        self._keep = t  # until the context's stream has consumed it

    def finish(self, slots, numbers):
        n = len(slots)
        self.L.x264hip_prefetch_ex.argtypes = [self.C.c_void_p, self.C.c_void_p, self.C.c_void_p, self.C.c_int, self.C.c_int]
        self._ck(self.L.x264hip_prefetch_ex(self.h, (self.C.c_int * n)(*slots), (self.C.c_int * n)(*numbers), n, 1), "prefetch_ex")


def run_window_shard(torch, lib, dist, rank, world, dev_index, cfg, dev_clip, exchange_on_device, qp_offsets=False):
    """One pass of ONE stream over `world` ranks: returns (outputs on rank 0 | None, seconds, WindowShard.stats).
    dev_clip: [F, H, W] tensor of the whole clip resident on this rank's GPU (the same content on every rank)."""
    import time
    F, W = dev_clip.shape[0], cfg["width"]
    ptrs = [dev_clip[i].data_ptr() for i in range(F)]
    L = lib.load()
    adapter = HipAdapter(L, None, cfg, lambda n: (ptrs[n], W), torch.device("cuda", dev_index))
    ws = WindowShard(adapter, dist, rank, world, device=torch.device("cuda", dev_index) if exchange_on_device else None)
    # every rank opens the same context geometry; only rank 0 drives its lookahead
    la = lib.Lookahead(cfg, device=dev_index, max_frames=F + 4, prefetch_hook=ws.on_prefetch if rank == 0 else None)
    adapter.h = la.ctx_handle()
    adapter.own_ingest = rank != 0
    try:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        outs = None
        if rank == 0:
            outs = la.run(device_ptrs=ptrs, stride=W, paced=False, qp_offsets=qp_offsets)
            ws.stop()
        else:
            ws.serve()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return outs, time.perf_counter() - t0, ws.stats
    finally:
        la.close()
