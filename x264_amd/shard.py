"""Sharding of the lookahead across the GPUs of one node, one process per GPU (torch.distributed: backend "nccl" = RCCL over xGMI,
gloo on CPU for the tests).  Two forms:

* WindowShard -- ONE stream, one lookahead window, the frames of the window dealt round-robin to the ranks (BASELINE
  configs[3]): rank b % world runs frame b's motion searches AND its cost cells; the only fields that cross ranks are the list-0
  fields B cells read from their list-1 reference (encoder/slicetype.c:629-642), and rank 0 -- which takes the decisions and runs
  MB-tree -- receives per-cell SUMMARIES (the sums of slicetype.c:946-991 + row sums) plus, on request, the per-block maps of the
  cells MB-tree propagation reads.  Slice types and every cost cell are those of the single-stream run (no IDR is forced anywhere).
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
    rec = getattr(outs, "records", None)
    if rec is not None and len(rec) == len(outs):
        # the outputs of one run_frames call as a record array: the same rows without a Python loop over the frames
        ce, ty = rec["cost_est"], rec["type"]
        cost = np.where(ty < 3, ce[:, 0, 0], np.where(ty == 3, np.maximum(ce[:, 1:, 0].max(axis=1), 0), np.maximum(ce[:, _BMASK].max(axis=1), 0)))
        rows[:, 0] = rec["frame"] + frame_offset; rows[:, 1] = ty; rows[:, 2] = cost; rows[:, 3] = rec["bframes"]
        return rows
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
CMD_STOP, CMD_CHUNK, CMD_FETCH = 0, 1, 2


class WindowShard:
    """Protocol shared by all ranks.  Rank 0 runs the host lookahead; whenever it would submit speculative work for a chunk of
    frames (the host logic's prefetch call: every resident frame the next decisions can reach) it broadcasts the chunk.  Every rank
    derives the same plan from it -- field (b, list, d) and cell (b, d0, d1) belong to rank b % world -- and then, per chunk:
      1. searches the fields of its frames;
      2. takes part in ONE exchange of the list-0 fields some other rank's B cells read from their list-1 reference;
      3. evaluates the cost cells of its frames;
      4. contributes their summaries to a gather on rank 0, which registers the other ranks' fields as searched elsewhere and takes
         the summaries in as speculative cells.
    The per-block maps stay with their owners; when MB-tree on rank 0 is about to read some (before_mbtree), rank 0 names them in a
    FETCH command and a gather brings exactly those.  The other ranks sit in serve().  With exchange on the device every step is
    enqueued on the contexts' own streams (collectives included): rank 0's host thread never waits for a chunk, it only waits -- per
    batch, inside x264hip_frame_cost -- for a result it actually needs.

    adapter (duck-typed; HipAdapter below, an oracle-backed one in the tests):
        n_mb, mb_h, bframes                              geometry
        ingest(slot, frame_number)                       make the frame resident in `slot` (no-op if it already is)
        classes() -> (mask_l0, mask_l1, cell_class)      rank 0: the (list, distance) and (d0, d1) classes worth speculating
        search(reqs)                                     reqs: list of (slot_b, slot_ref, list, dist_m1): unweighted searches, asynchronous
        export_fields(keys, out)                         {mv, cost} of the fields (slot, list, dist_m1) into out[:len] ([., n_mb, 2] int32)
        import_fields(keys, tensor, rows)                field k from tensor[rows[k]]
        spec_cells(cells)                                cells: list of (slot_b, slot_p0, slot_p1, d0, d1, flags); d0 == 0: intra sums; flags: 1 with the list-1
                                                         reference's vectors, 2 both ways in one pass (own place + spare half), 4 (export / import) the spare half
        export_cells(cells, out)                         summaries into out[:len] ([., 8 + 2 mb_h] int32)
        import_cells(cells, tensor)                      rank 0
        fields_remote(keys)                              rank 0: keys (slot, frame_number, list, dist_m1) searched on other ranks
        cells_missing(cells) -> [0 | 1 | 2]              rank 0: whose per-block map is not local (2: the map of the cell's spare half)
        export_map(cell, out) ; import_map(cell, tensor) the per-block map of one cell ([3, n_mb] int32: lowres_costs, list-0 vectors, list-1 vectors)
        exchange                                         Exchange (below): zeros / all_gather / gather
    """

    def __init__(self, adapter, dist, rank, world, device=None, cmd_group=None, loopback=False):
        self.a, self.dist, self.rank, self.world, self.device, self.cmd_group = adapter, dist, rank, world, device, cmd_group
        # loopback (world == 1 only): every exchange step runs anyway, over the one-rank process group, with this rank as its own peer --
        # the export kernels, the collectives on the context's stream and the imports execute once on a single GPU (run_window_shard)
        self.loop = bool(loopback) and world == 1
        self.done = set()        # (frame_number, list, dist_m1) fields planned in this or an earlier chunk (same on every rank)
        self.cells_done = set()  # (frame_number, d0, d1)
        self.sums_done = set()   # rank 0: frames whose intra sums have been queued
        self.l0_sent = set()     # (receiving rank, frame_number, dist_m1): list-0 fields already shipped (same on every rank)
        self.slot_of = {}        # frame number -> slot, as announced by rank 0
        self.number_in = {}      # slot -> frame number
        self.stats = dict(chunks=0, fields_searched=0, cells_evaluated=0, l0_fields_exchanged=0, cells_imported=0, maps_fetched=0,
                          bytes_l0_exchange=0, bytes_l0_sent=0, bytes_l0_received=0, bytes_summaries=0, bytes_maps=0, bytes_input_broadcast=0,
                          fetch_commands=0, loopback_checks=0)
        self.ingested = set()    # frame numbers whose picture is on this rank (broadcast per chunk)

    # ---- the plan of a chunk: identical on every rank ----
    def plan(self, slots, numbers, masks, cell_class):
        bf, ns = self.a.bframes, self.a.bframes + 2
        owner = lambda n: n % self.world  # noqa: E731
        fields = [[] for _ in range(self.world)]
        for i, ni in enumerate(numbers):
            for j, nj in enumerate(numbers):
                d = nj - ni
                if d == 0 or abs(d) > bf + 1 or (d > 0 and not bf):
                    continue
                lst, dm1 = int(d > 0), abs(d) - 1
                if not (masks[lst] >> dm1) & 1 or (ni, lst, dm1) in self.done:
                    continue
                self.done.add((ni, lst, dm1))
                fields[owner(ni)].append((slots[i], slots[j], lst, dm1, ni))
        cells = [[] for _ in range(self.world)]
        l0_wanted = [set() for _ in range(self.world)]  # per receiving rank: (frame_number, dist_m1) list-0 fields of other ranks' frames
        here = dict(zip(numbers, slots))  # a chunk names every resident frame the decisions can reach: only those are safe to refer to
        for i, ni in enumerate(numbers):
            for d0 in range(1, bf + 2):
                p0 = ni - d0
                if p0 not in here or (ni, 0, d0 - 1) not in self.done:
                    continue
                for d1 in range(0, bf + 2 - d0):
                    c = cell_class[d0 * ns + d1]
                    if not c or (ni, d0, d1) in self.cells_done:
                        continue
                    p1, with_l0 = ni, 0
                    if d1:
                        p1, with_l0 = ni + d1, int(c == 2)
                        if p1 not in here or (ni, 1, d1 - 1) not in self.done:
                            continue
                        if c == 3:  # both ways in one pass where the list-1 reference's field exists (flags 1 | 2), else without it
                            with_l0 = 3 if (p1, 0, d0 + d1 - 1) in self.done else 0
                        if with_l0 and (p1, 0, d0 + d1 - 1) not in self.done:
                            continue
                        if with_l0 and (owner(p1) != owner(ni) or self.loop):
                            l0_wanted[owner(ni)].add((p1, d0 + d1 - 1))
                    self.cells_done.add((ni, d0, d1))
                    cells[owner(ni)].append((slots[i], here[p0], here[p1], d0, d1, with_l0, ni))
        return fields, cells, l0_wanted

    def _bcast_cmd(self, t):
        """small int64 command tensors: over the side group (gloo) when there is one, so that no rank touches the GPU for them"""
        if self.cmd_group is not None:
            self.dist.broadcast(t, src=0, group=self.cmd_group)
            return t
        t = t.to(self.device) if self.device is not None else t
        self.dist.broadcast(t, src=0)
        return t.cpu()

    def _send(self, cmd):
        import torch
        if getattr(self, "command_log", None) is not None:
            self.command_log.append([int(v) for v in cmd])  # (tests: held word for word against the C shard's commands)
        if self.world > 1:
            t = torch.tensor(cmd, dtype=torch.int64)
            self._bcast_cmd(torch.tensor([t.numel()], dtype=torch.int64))
            self._bcast_cmd(t)

    def _run_chunk(self, slots, numbers, masks, cell_class):
        import torch
        a, X = self.a, self.a.exchange
        # ---- the pictures of the chunk's new frames: ONE broadcast per run of consecutive frame numbers, on the context's stream (the ingest
        # kernels of this rank are ordered behind it there; nothing waits on the host)
        if getattr(a, "bcast_clip", None) is not None and (self.world > 1 or self.loop):
            new = sorted(n for n in numbers if n not in self.ingested)
            self.ingested.update(new)
            lo = 0
            while lo < len(new):
                hi = lo
                while hi + 1 < len(new) and new[hi + 1] == new[hi] + 1:
                    hi += 1
                self.stats["bytes_input_broadcast"] += a.broadcast_frames(new[lo], new[hi] + 1)
                lo = hi + 1
        for s, n in zip(slots, numbers):
            if self.number_in.get(s) != n:
                self.slot_of.pop(self.number_in.get(s), None)
                self.number_in[s] = n
                self.slot_of[n] = s   # (a frame number never comes back in another slot within a stream)
                a.ingest(s, n)
        fields, cells, l0_wanted = self.plan(slots, numbers, masks, cell_class)
        mine = fields[self.rank]
        a.search([f[:4] for f in mine])
        self.stats["chunks"] += 1
        self.stats["fields_searched"] += len(mine)
        # ---- list-0 fields that B cells on OTHER ranks read from their list-1 reference: every rank sends each peer exactly the fields
        # that peer asked for (the plan is the same on every rank, so both sides know the counts), one exchange per chunk
        if self.world > 1 or self.loop:
            for r in range(self.world):
                l0_wanted[r] = {k for k in l0_wanted[r] if (r,) + k not in self.l0_sent}
                self.l0_sent.update((r,) + k for k in l0_wanted[r])
            give = [[sorted(k for k in l0_wanted[r] if k[0] % self.world == o) for r in range(self.world)] for o in range(self.world)]  # [owner][receiver]
            n_send = [len(g) for g in give[self.rank]]
            n_recv = [len(give[o][self.rank]) for o in range(self.world)]
            if sum(len(g) for row in give for g in row):
                sbuf = X.zeros((max(sum(n_send), 1), a.n_mb, 2))
                keys = [k for g in give[self.rank] for k in g]
                if keys:
                    a.export_fields([(self.slot_of[n], 0, dm1) for n, dm1 in keys], sbuf)
                rbuf = X.send_recv(sbuf, n_send, n_recv)
                row = 0
                for o in range(self.world):
                    got = give[o][self.rank]
                    if got:
                        a.import_fields([(self.slot_of[n], 0, dm1) for n, dm1 in got], rbuf, list(range(row, row + len(got))))
                        self.stats["l0_fields_exchanged"] += len(got)
                    row += len(got)
                self.stats["bytes_l0_sent"] += sum(n_send) * a.n_mb * 8
                self.stats["bytes_l0_received"] += sum(n_recv) * a.n_mb * 8
                self.stats["bytes_l0_exchange"] += sum(n_recv) * a.n_mb * 8
                if self.loop and keys:
                    self._loop_checks = getattr(self, "_loop_checks", []) + [(sbuf[:len(keys)], rbuf[:len(keys)])]
        # ---- the cells of the frames this rank owns; rank 0 also queues the intra sums of every frame (it has all of them resident)
        my_cells = [c[:6] for c in cells[self.rank]]
        # summaries: a cell evaluated both ways travels as two entries, its own (flag 1) and its spare half (flag 4)
        halves = lambda cs: [h for c in cs for h in ([c[:5] + (1,), c[:5] + (4,)] if c[5] & 2 else [c[:6]])]  # noqa: E731
        sums = []
        if self.rank == 0:
            sums = [(s, s, s, 0, 0, 0) for s, n in zip(slots, numbers) if n not in self.sums_done]
            self.sums_done.update(numbers)
        a.spec_cells(sums + my_cells)
        self.stats["cells_evaluated"] += len(my_cells)
        if self.world == 1 and not self.loop:
            return
        # ---- summaries to rank 0
        sent = [halves(cs) for cs in cells]
        width = max([len(c) for c in sent[1:]] + ([len(sent[0])] if self.loop else [0]))
        if width:
            per = 8 + 2 * a.mb_h
            buf = X.zeros((width, per))
            if sent[self.rank] and (self.rank or self.loop):
                a.export_cells(sent[self.rank], buf)
            out = X.gather(buf)
            if self.rank == 0:
                remote = [(f[0], f[4], f[2], f[3]) for r in range(1, self.world) for f in fields[r]]
                a.fields_remote(remote)
                for r in range(0 if self.loop else 1, self.world):
                    if sent[r]:
                        a.import_cells(sent[r], out[r][:len(sent[r])])   # (loopback: every entry is skipped -- the cells are here)
                        self.stats["cells_imported"] += len(cells[r])
                self.stats["bytes_summaries"] += (self.world - 1) * width * per * 4
                if self.loop and sent[0]:
                    self._loop_checks = getattr(self, "_loop_checks", []) + [(buf[:len(sent[0])], out[0][:len(sent[0])])]
        elif self.rank == 0:
            a.fields_remote([(f[0], f[4], f[2], f[3]) for r in range(1, self.world) for f in fields[r]])

    def loopback_verify(self, sync):
        """loopback: what came back from every exchange equals what the export kernels wrote (the collectives ran behind them on the stream)"""
        import torch
        sync()
        for sent, got in getattr(self, "_loop_checks", []):
            assert torch.equal(sent.cpu(), got.cpu()), "loop-back exchange returned different data"
            self.stats["loopback_checks"] += 1
        self._loop_checks = []

    # ---- rank 0 ----
    def on_prefetch(self, slots, numbers):
        """the host logic's speculative submission on rank 0 (x264hip_prefetch_hook / the backend's prefetch entry)"""
        m0, m1, cc = self.a.classes()
        n = len(slots)
        self._send([CMD_CHUNK, n, m0, m1] + list(slots) + list(numbers) + [int(v) for v in cc])
        self._run_chunk(list(slots), list(numbers), (m0, m1), cc)

    def before_mbtree(self, cells):
        """cells: (slot_b, slot_p0, slot_p1, d0, d1) of the PROPAGATE steps MB-tree is about to run on rank 0: the per-block maps that
        are still with their owners are fetched (one FETCH command + one gather)"""
        if self.world == 1 or not cells:
            return
        cells = [tuple(c) for c in dict.fromkeys(cells)]
        miss = [c + (4 if m == 2 else 0,) for c, m in zip(cells, self.a.cells_missing([c + (0,) for c in cells])) if m]  # (.., half of the cell that is wanted)
        if not miss:
            return
        payload = [CMD_FETCH, len(miss)]
        for c in miss:
            payload += list(c) + [self.number_in[c[0]]]
        self._send(payload)
        self._fetch(miss, [self.number_in[c[0]] for c in miss])

    def _fetch(self, cells, numbers):
        a, X = self.a, self.a.exchange
        by_owner = [[c for c, n in zip(cells, numbers) if n % self.world == r] for r in range(self.world)]
        width = max(len(x) for x in by_owner[1:]) if self.world > 1 else 0
        if not width:
            return
        buf = X.zeros((width, 3, a.n_mb))
        if self.rank:
            for k, c in enumerate(by_owner[self.rank]):
                a.export_map(c, buf[k])
        out = X.gather(buf)
        if self.rank == 0:
            for r in range(1, self.world):
                for k, c in enumerate(by_owner[r]):
                    a.import_map(c[:5] + (0,), out[r][k])
                    self.stats["maps_fetched"] += 1
            self.stats["bytes_maps"] += (self.world - 1) * width * 3 * a.n_mb * 4
            self.stats["fetch_commands"] += 1

    def stop(self):
        self._send([CMD_STOP])

    # ---- ranks 1 .. world-1 ----
    def serve(self):
        import torch
        ns2 = (self.a.bframes + 2) ** 2
        while True:
            ln = int(self._bcast_cmd(torch.zeros(1, dtype=torch.int64))[0])
            cmd = self._bcast_cmd(torch.zeros(ln, dtype=torch.int64)).tolist()
            if cmd[0] == CMD_STOP:
                return
            if cmd[0] == CMD_FETCH:
                n = cmd[1]
                rec = [cmd[2 + 7 * k: 9 + 7 * k] for k in range(n)]
                self._fetch([tuple(r[:6]) for r in rec], [r[6] for r in rec])
                continue
            n = cmd[1]
            self._run_chunk(cmd[4:4 + n], cmd[4 + n:4 + 2 * n], (cmd[2], cmd[3]), cmd[4 + 2 * n:4 + 2 * n + ns2])


class Exchange:
    """The collectives of WindowShard over torch.distributed.  on_device: tensors live on this rank's GPU and every collective is
    issued under the context's own HIP stream (backend nccl = RCCL over xGMI), so exports, collectives and imports are ordered on
    the device and no host thread waits.  Otherwise (gloo: the CPU tests, and two ranks sharing one GPU) tensors are staged through
    host memory: `before` runs ahead of a collective (wait for the exports), `after` behind it (wait for the upload)."""

    def __init__(self, dist, rank, world, torch_device=None, stream=None, before=None, after=None):
        self.dist, self.rank, self.world, self.dev, self.stream, self.before, self.after = dist, rank, world, torch_device, stream, before, after
        self.on_device = stream is not None

    def _ctx(self):
        import contextlib
        import torch
        return torch.cuda.stream(self.stream) if self.on_device else contextlib.nullcontext()

    def zeros(self, shape):
        import torch
        with self._ctx():
            t = torch.zeros(shape, dtype=torch.int32, device=self.dev if self.dev is not None else "cpu")
        if not self.on_device and self.dev is not None:
            torch.cuda.synchronize(self.dev)  # the fill ran on torch's stream; the context's export kernels write on theirs
        return t

    def _host(self, t):
        if self.on_device or self.dev is None:
            return t
        if self.before:
            self.before()
        return t.cpu()

    def _back(self, ts):
        if self.on_device or self.dev is None:
            return ts
        ts = [t.to(self.dev) for t in ts]
        if self.after:
            self.after()
        return ts

    def all_gather(self, buf):
        import torch
        with self._ctx():
            h = self._host(buf)
            out = [torch.empty_like(h) for _ in range(self.world)]
            self.dist.all_gather(out, h)
            return self._back(out)

    def send_recv(self, sbuf, n_send, n_recv):
        """sbuf: rows for peer 0, then for peer 1, ... (n_send[r] rows each, first dimension); returns the rows received, peer after peer
        (n_recv[r] from rank r).  On the device (RCCL): ONE all_to_all_single with those split sizes on the context's stream.  Staged
        through the host (gloo has no all-to-all): one isend / irecv per peer with something to move."""
        import torch
        rows = max(sum(n_recv), 1)
        with self._ctx():
            if self.on_device:
                out = torch.empty((rows,) + tuple(sbuf.shape[1:]), dtype=sbuf.dtype, device=sbuf.device)
                self.dist.all_to_all_single(out[:sum(n_recv)], sbuf[:sum(n_send)], output_split_sizes=list(n_recv), input_split_sizes=list(n_send))
                return out
            h = self._host(sbuf)
            out = torch.zeros((rows,) + tuple(h.shape[1:]), dtype=h.dtype)
            ops, so, ro = [], 0, 0
            for r in range(self.world):
                if r == self.rank:
                    out[ro:ro + n_recv[r]] = h[so:so + n_send[r]]  # (loopback: a rank is its own peer)
                else:
                    if n_send[r]:
                        ops.append(self.dist.P2POp(self.dist.isend, h[so:so + n_send[r]].contiguous(), r))
                    if n_recv[r]:
                        ops.append(self.dist.P2POp(self.dist.irecv, out[ro:ro + n_recv[r]], r))
                so += n_send[r]; ro += n_recv[r]
            if ops:
                for w in self.dist.batch_isend_irecv(ops):
                    w.wait()
            return self._back([out])[0]

    def broadcast(self, t):
        """t from rank 0 to everybody, in place"""
        with self._ctx():
            if self.on_device or self.dev is None:
                self.dist.broadcast(t, src=0)
                return
            if self.before:
                self.before()
            h = t.cpu()
            self.dist.broadcast(h, src=0)
            if self.rank:
                t.copy_(h)
            if self.after:
                self.after()

    def gather(self, buf):
        import torch
        with self._ctx():
            h = self._host(buf)
            out = [torch.empty_like(h) for _ in range(self.world)] if self.rank == 0 else None
            self.dist.gather(h, out, dst=0)
            return self._back(out) if out is not None else None


class HipAdapter:
    """WindowShard over a device context (x264hip_ctx) of this rank's GPU.  frames: callable frame_number -> (device pointer, stride)
    of the full-resolution luma on THIS device (a rank reads the references of its frames' searches: with 2 (bframes + 1) + 1 >= world
    that is every frame of the window, so the input is broadcast once -- run_window_shard -- and every rank makes its own planes)."""

    def __init__(self, L, ctx_handle, cfg, frames, torch_device, dist=None, rank=0, world=1, exchange_on_device=True):
        import ctypes as C
        self.C, self.L, self.h, self.cfg, self.frames, self.dev = C, L, ctx_handle, cfg, frames, torch_device
        mb_w, self.mb_h = (cfg["width"] + 15) // 16, (cfg["height"] + 15) // 16
        self.n_mb, self.bframes = mb_w * self.mb_h, cfg["bframes"]
        self.own_ingest = False  # rank 0's frames are ingested by its host lookahead (put_frame)
        self.dist, self.rank, self.world, self.exchange_on_device = dist, rank, world, exchange_on_device
        self.exchange = None
        self._keep = []
        L.x264hip_frame_put.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.x264hip_export_field.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.x264hip_import_field.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        for f in ("x264hip_spec_cells", "x264hip_export_cells", "x264hip_import_cells"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_void_p] + ([C.c_void_p] if f != "x264hip_spec_cells" else [])
        L.x264hip_fields_remote.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.x264hip_cells_missing.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.x264hip_export_cell_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.x264hip_import_cell_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.x264hip_cell_classes.argtypes = [C.c_void_p, C.c_void_p]
        L.x264hip_stream_handle.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]

    def attach(self, ctx_handle, loopback=False):
        """the context exists now: bind the exchange to its stream"""
        import torch
        self.h = ctx_handle
        sp = self.C.c_void_p()
        self._ck(self.L.x264hip_stream_handle(self.h, self.C.byref(sp)), "stream_handle")
        self.ext = torch.cuda.ExternalStream(sp.value, device=self.dev)  # the context's stream as torch sees it (events, collectives)
        stream = self.ext if self.exchange_on_device and (self.world > 1 or loopback) else None
        self.stream = stream
        self.exchange = Exchange(self.dist, self.rank, self.world, self.dev, stream,
                                 before=lambda: self._ck(self.L.x264hip_synchronize(self.h), "synchronize"),
                                 after=lambda: torch.cuda.synchronize(self.dev))

    def _ck(self, rc, what):
        from . import lib
        lib._ck(rc, what)

    def _refs(self, cells):
        from . import lib
        arr = (lib.CellRef * len(cells))()
        for i, c in enumerate(cells):
            arr[i] = lib.CellRef(*[int(v) for v in c])
        return arr

    def _hold(self, t):
        """a tensor the context's stream reads or writes in kernels that have just been enqueued: kept until an event recorded on that
        stream behind them has completed (not for a number of calls: a slow stream must not lose its buffers)"""
        import torch
        ev = torch.cuda.Event()
        ev.record(self.ext)
        self._keep.append((ev, t))
        while self._keep and self._keep[0][0].query():
            self._keep.pop(0)
        return t

    def broadcast_frames(self, lo, hi):
        """the pictures of frames lo .. hi-1 from rank 0 to every rank (in place in the clip tensor); returns the bytes a rank receives"""
        part = self.bcast_clip[lo:hi]
        self.exchange.broadcast(part)
        return int(part.numel() * part.element_size()) if self.world > 1 else 0

    def ingest(self, slot, number):
        if not self.own_ingest:
            return
        ptr, stride = self.frames(number)
        self._ck(self.L.x264hip_frame_put(self.h, slot, self.C.c_void_p(ptr), stride, 1, None, None, 0, None), "frame_put")

    def classes(self):
        a, b = self.C.c_uint(), self.C.c_uint()
        self._ck(self.L.x264hip_field_classes(self.h, self.C.byref(a), self.C.byref(b)), "field_classes")
        cc = (self.C.c_ubyte * ((self.bframes + 2) ** 2))()
        self._ck(self.L.x264hip_cell_classes(self.h, cc), "cell_classes")
        return a.value, b.value, list(cc)

    def search(self, reqs):
        if not reqs:
            return
        n = len(reqs)
        arr = lambda k: (self.C.c_int * n)(*[r[k] for r in reqs])  # noqa: E731
        self._ck(self.L.x264hip_search_fields(self.h, n, arr(0), arr(1), arr(2), arr(3)), "search_fields")

    def export_fields(self, keys, out):
        for i, (slot, lst, dm1) in enumerate(keys):
            self._ck(self.L.x264hip_export_field(self.h, slot, lst, dm1, self.C.c_void_p(out[i].data_ptr())), "export_field")
        self._hold(out)

    def import_fields(self, keys, t, rows):
        assert t.is_contiguous()
        for (slot, lst, dm1), i in zip(keys, rows):
            self._ck(self.L.x264hip_import_field(self.h, slot, lst, dm1, self.C.c_void_p(t[i].data_ptr())), "import_field")
        self._hold(t)

    def spec_cells(self, cells):
        if cells:
            self._ck(self.L.x264hip_spec_cells(self.h, len(cells), self._refs(cells)), "spec_cells")

    def export_cells(self, cells, out):
        self._ck(self.L.x264hip_export_cells(self.h, len(cells), self._refs(cells), self.C.c_void_p(out.data_ptr())), "export_cells")
        self._hold(out)

    def import_cells(self, cells, t):
        assert t.is_contiguous()
        self._ck(self.L.x264hip_import_cells(self.h, len(cells), self._refs(cells), self.C.c_void_p(t.data_ptr())), "import_cells")
        self._hold(t)

    def fields_remote(self, keys):
        if not keys:
            return
        n = len(keys)
        arr = lambda k: (self.C.c_int * n)(*[int(r[k]) for r in keys])  # noqa: E731
        self._ck(self.L.x264hip_fields_remote(self.h, n, arr(0), arr(1), arr(2), arr(3)), "fields_remote")

    def cells_missing(self, cells):
        out = (self.C.c_ubyte * len(cells))()
        self._ck(self.L.x264hip_cells_missing(self.h, len(cells), self._refs(cells), out), "cells_missing")
        return [int(v) for v in out]  # 0 here, 1 with its owner, 2 the spare half is with its owner

    def export_map(self, cell, out):
        self._ck(self.L.x264hip_export_cell_map(self.h, self._refs([cell]), self.C.c_void_p(out.data_ptr())), "export_cell_map")
        self._hold(out)

    def import_map(self, cell, t):
        assert t.is_contiguous()
        self._ck(self.L.x264hip_import_cell_map(self.h, self._refs([cell]), self.C.c_void_p(t.data_ptr())), "import_cell_map")
        self._hold(t)


def run_window_shard(torch, lib, dist, rank, world, dev_index, cfg, dev_clip, exchange_on_device, qp_offsets=False, broadcast_input=False, vbv=False,
                     loopback=False, profile=False, command_log=None):
    """One pass of ONE stream over `world` ranks: returns (outputs on rank 0 | None, seconds, WindowShard.stats).
    dev_clip: [F, H, W] tensor of the whole clip resident on this rank's GPU (the same content on every rank), or, with
    broadcast_input, the clip on rank 0 and an uninitialised tensor of the same shape elsewhere: the pictures are then broadcast inside the
    timed region, chunk by chunk as the stream advances (W x H bytes per frame over xGMI, on the contexts' streams), before the ranks make
    their own lowres planes.
    loopback (world == 1, a one-rank process group): the whole exchange machinery runs with this rank as its own peer -- input broadcast,
    export kernels, the collectives on the context's stream, imports -- and what comes back is compared with what was sent."""
    import time
    F, W = dev_clip.shape[0], cfg["width"]
    ptrs = [dev_clip[i].data_ptr() for i in range(F)]
    L = lib.load()
    dev = torch.device("cuda", dev_index)
    loopback = bool(loopback) and world == 1 and dist is not None
    adapter = HipAdapter(L, None, cfg, lambda n: (ptrs[n], W), dev, dist, rank, world, exchange_on_device)
    adapter.bcast_clip = dev_clip if (broadcast_input and world > 1) or loopback else None
    cmd_group = None
    if world > 1 and exchange_on_device and dist.get_backend() == "nccl":
        import datetime
        # commands are a few int64 words: keep them off the GPU.  (A rank that fails leaves the protocol; the others then sit in a collective
        # until this timeout -- ten minutes, like the RCCL group's in bench.py -- instead of for ever.)
        cmd_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=10))
    ws = WindowShard(adapter, dist, rank, world, device=dev if (exchange_on_device and cmd_group is None) else None, cmd_group=cmd_group, loopback=loopback)
    ws.command_log = command_log
    # every rank opens the same context geometry; only rank 0 drives its lookahead
    # (one rank without loopback: nothing to spread -- the context's own speculative submission does the same work in fewer, larger launches
    # and evaluates every B cell both ways in one pass, x264hip_prefetch)
    hooked = world > 1 or loopback
    la = lib.Lookahead(cfg, device=dev_index, max_frames=F + 4, prefetch_hook=ws.on_prefetch if rank == 0 and hooked else None,
                       mbtree_hook=ws.before_mbtree if rank == 0 and world > 1 else None)
    adapter.attach(la.ctx_handle(), loopback=loopback)
    adapter.own_ingest = rank != 0
    try:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if profile:  # HIP events around the search and cell launches of this pass (the work the shard spreads): two events per launch and a
            lib.search_profile(L, la.ctx_handle(), 1)  # stream sync per 1024 pairs -- not in a pass whose time is reported
        t0 = time.perf_counter()
        outs = None
        if rank == 0:
            try:
                outs = la.run(device_ptrs=ptrs, stride=W, paced=False, qp_offsets=qp_offsets, vbv=vbv)
            finally:
                if world > 1:
                    ws.stop()  # also on failure: the other ranks are waiting for a command
        else:
            ws.serve()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if loopback:
            ws.loopback_verify(lambda: torch.cuda.synchronize())
        if profile:
            ms_s, nl_s, n_s = lib.search_profile(L, la.ctx_handle(), -1)
            ms_c, nl_c, n_c = lib.cell_profile(L, la.ctx_handle())
            ws.stats.update(device_ms_searches=round(ms_s, 3), search_launches=nl_s, device_ms_cells=round(ms_c, 3), cell_launches=nl_c, cells_in_launches=n_c)
        counters = np.zeros(16, np.uint64)
        import ctypes as C
        lib._ck(L.x264hip_counters(la.ctx_handle(), counters.ctypes.data_as(C.c_void_p), 16), "counters")
        ws.stats.update(searches_here=int(counters[0]), cells_here=int(counters[5]), cells_on_demand=int(counters[7]), remote_fields_searched_here=int(counters[8]),
                        remote_maps_recomputed_here=int(counters[9]), maps_imported=int(counters[10]), cells_imported_ctx=int(counters[11]))
        return outs, dt, ws.stats
    finally:
        la.close()
        if cmd_group is not None:
            dist.destroy_process_group(cmd_group)


# ---------------------------------------------------------------------------------------------------------------------------
# The same window shard through the library's C entry points (x264_amd/csrc/shard_host.cpp: x264hip_shard_open / _put_frames / _serve /
# _close over an x264hip_shard_transport).  Everything above orchestrates from Python and is what the CPU tests run over the oracle
# backend (no device there); on a device the C path is the product -- a C host calls exactly these entries --
# and the code below only binds it for the tests and bench.py.
# ---------------------------------------------------------------------------------------------------------------------------
def _ctypes_defs():
    import ctypes as C
    BCAST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
    SENDRECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p)
    GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
    ALLRED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
    DESTROY = C.CFUNCTYPE(None, C.c_void_p)

    class Transport(C.Structure):
        """x264hip_shard_transport of include/x264hip.h"""
        _fields_ = [("user", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("loopback", C.c_int), ("broadcast", BCAST), ("send_recv", SENDRECV),
                    ("gather", GATHER), ("allreduce_max_i32", ALLRED), ("destroy", DESTROY)]
    return C, Transport, (BCAST, SENDRECV, GATHER, ALLRED, DESTROY)


def rccl_unique_id(L):
    """128 bytes from x264hip_rccl_unique_id (rank 0); the caller hands them to the other ranks"""
    import ctypes as C
    from . import lib
    buf = (C.c_char * 128)()
    lib._ck(L.x264hip_rccl_unique_id(buf), "rccl_unique_id")
    return bytes(buf)


def rccl_transport(L, unique_id, rank, world, device, loopback=False):
    C, Transport, _ = _ctypes_defs()
    from . import lib
    t = Transport()
    L.x264hip_shard_transport_rccl.argtypes = [C.POINTER(Transport), C.c_char_p, C.c_int, C.c_int, C.c_int]
    lib._ck(L.x264hip_shard_transport_rccl(C.byref(t), unique_id, rank, world, device), "shard_transport_rccl")
    t.loopback = int(bool(loopback) and world == 1)
    return t


class HostStagedTransport:
    """An x264hip_shard_transport whose four operations go through host memory and torch.distributed (gloo): what lets two ranks share
    ONE GPU in the tests (RCCL refuses two ranks on a device).  Every operation waits for the stream, moves the bytes over gloo and
    copies the result back before it returns -- the semantics of the device transport without its asynchrony."""

    def __init__(self, dist, rank, world):
        import ctypes as C
        self.C, self.dist, self.rank, self.world = C, dist, rank, world
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        _, Transport, (BCAST, SENDRECV, GATHER, ALLRED, DESTROY) = _ctypes_defs()
        self._fns = (BCAST(self._broadcast), SENDRECV(self._send_recv), GATHER(self._gather), ALLRED(self._allreduce))
        self.struct = Transport(None, rank, world, 0, self._fns[0], self._fns[1], self._fns[2], self._fns[3], DESTROY())
        self.calls = dict(broadcast=0, send_recv=0, gather=0, allreduce=0)
        self.command_log = None  # a list: every command block rank 0 broadcasts is decoded into it (shard_host.cpp send_cmd: word 0 = words used)

    def _down(self, ptr, n, stream):
        import torch
        assert self.hip.hipStreamSynchronize(stream) == 0
        t = torch.empty(max(n, 1), dtype=torch.uint8)
        if n:
            assert self.hip.hipMemcpy(t.data_ptr(), ptr, n, 2) == 0  # hipMemcpyDeviceToHost
        return t

    def _up(self, ptr, t, n):
        if n:
            assert self.hip.hipMemcpy(ptr, t.data_ptr(), n, 1) == 0  # hipMemcpyHostToDevice

    def _guard(self, f, *a):
        try:
            f(*a)
            return 0
        except Exception:  # never let an exception cross the C ABI
            import traceback
            traceback.print_exc()
            return -1

    def _broadcast(self, user, buf, n, root, stream):
        def go():
            import torch
            self.calls["broadcast"] += 1
            t = self._down(buf, n, stream)
            if self.command_log is not None and self.rank == root and n == SHARD_CMD_WORDS * 8:
                w = t.view(torch.int64)
                self.command_log.append(w[1:1 + int(w[0])].tolist())
            self.dist.broadcast(t, src=root)
            if self.rank != root:
                self._up(buf, t, n)
        return self._guard(go)

    def _send_recv(self, user, sbuf, sb, rbuf, rb, stream):
        def go():
            import torch
            self.calls["send_recv"] += 1
            sb_, rb_ = [sb[r] for r in range(self.world)], [rb[r] for r in range(self.world)]
            h = self._down(sbuf, sum(sb_), stream)
            out = torch.zeros(max(sum(rb_), 1), dtype=torch.uint8)
            ops, so, ro = [], 0, 0
            for r in range(self.world):
                if r == self.rank:
                    out[ro:ro + rb_[r]] = h[so:so + sb_[r]]
                else:
                    if sb_[r]:
                        ops.append(self.dist.P2POp(self.dist.isend, h[so:so + sb_[r]].contiguous(), r))
                    if rb_[r]:
                        ops.append(self.dist.P2POp(self.dist.irecv, out[ro:ro + rb_[r]], r))
                so += sb_[r]; ro += rb_[r]
            if ops:
                for w in self.dist.batch_isend_irecv(ops):
                    w.wait()
            self._up(rbuf, out, sum(rb_))
        return self._guard(go)

    def _gather(self, user, sbuf, rbuf, n, root, stream):
        def go():
            import torch
            self.calls["gather"] += 1
            h = self._down(sbuf, n, stream)
            out = [torch.empty_like(h) for _ in range(self.world)] if self.rank == root else None
            self.dist.gather(h, out, dst=root)
            if self.rank == root:
                self._up(rbuf, torch.cat(out), n * self.world)
        return self._guard(go)

    def _allreduce(self, user, buf, n, stream):
        def go():
            import torch
            self.calls["allreduce"] += 1
            t = self._down(buf, 4 * n, stream).view(torch.int32)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            self._up(buf, t.view(torch.uint8), 4 * n)
        return self._guard(go)


SHARD_CMD_WORDS = 8192  # shard_host.cpp CMD_WORDS: a command travels as one block of that many int64 words
SHARD_STAT_NAMES = ("chunks", "fields_searched", "cells_evaluated", "l0_fields_exchanged", "cells_imported", "maps_fetched", "fetch_commands",
                    "bytes_input_broadcast", "bytes_l0_received", "bytes_summaries", "bytes_maps", "maps_fetched_spare")


class CShard:
    """x264hip_shard of one rank.  Rank 0: put_frames(), then frames come out of .lookahead.get(); other ranks: serve()."""

    def __init__(self, L, cfg, transport, device=0, max_frames=0):
        import ctypes as C
        from . import lib
        self.C, self.L, self.cfg, self._t = C, L, cfg, transport
        self.params = lib.make_la_params(cfg, None, max_frames)
        self.h = C.c_void_p()
        tstruct = transport.struct if hasattr(transport, "struct") else transport
        L.x264hip_shard_open.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(lib.LaParams), C.c_void_p]
        lib._ck(L.x264hip_shard_open(C.byref(self.h), device, C.byref(self.params), C.byref(tstruct)), "shard_open")
        L.x264hip_shard_lookahead.restype = C.c_void_p
        L.x264hip_shard_lookahead.argtypes = [C.c_void_p]
        L.x264hip_shard_ctx.restype = C.c_void_p
        L.x264hip_shard_ctx.argtypes = [C.c_void_p]
        la = lib.Lookahead.__new__(lib.Lookahead)  # a view of the shard's own lookahead: get() / stats() as usual, never closed from here
        la.L, la.cfg, la.params, la.h = L, cfg, self.params, C.c_void_p(L.x264hip_shard_lookahead(self.h))
        la.dtype = np.uint8 if cfg["bit_depth"] == 8 else np.uint16
        la._n_put = 0
        self.lookahead = la
        self._n = 0

    def ctx_handle(self):
        return self.C.c_void_p(self.L.x264hip_shard_ctx(self.h))

    def put_frames(self, device_ptrs, stride=None):
        from . import lib
        n = len(device_ptrs)
        arr = (self.C.c_void_p * n)(*device_ptrs)
        self.L.x264hip_shard_put_frames.argtypes = [self.C.c_void_p, self.C.c_int, self.C.c_int, self.C.c_void_p, self.C.c_int]
        lib._ck(self.L.x264hip_shard_put_frames(self.h, self._n, n, arr, stride or self.cfg["width"]), "shard_put_frames")
        self._n += n

    def status(self):
        self.L.x264hip_shard_status.argtypes = [self.C.c_void_p]
        return self.L.x264hip_shard_status(self.h)

    def reset(self):
        """rank 0: a new sequence on every rank (x264hip_shard_reset)"""
        from . import lib
        self.L.x264hip_shard_reset.argtypes = [self.C.c_void_p]
        lib._ck(self.L.x264hip_shard_reset(self.h), "shard_reset")
        self._n = 0
        self.lookahead._n_put = 0

    def serve(self):
        self.L.x264hip_shard_serve.argtypes = [self.C.c_void_p]
        return self.L.x264hip_shard_serve(self.h)

    def stats(self):
        out = np.zeros(12, np.uint64)
        self.L.x264hip_shard_stats.argtypes = [self.C.c_void_p, self.C.c_void_p, self.C.c_int]
        self.L.x264hip_shard_stats(self.h, out.ctypes.data_as(self.C.c_void_p), 12)
        return {k: int(v) for k, v in zip(SHARD_STAT_NAMES, out)}

    def loopback_verify(self):
        from . import lib
        n = self.C.c_int(0)
        self.L.x264hip_shard_loopback_verify.argtypes = [self.C.c_void_p, self.C.POINTER(self.C.c_int)]
        lib._ck(self.L.x264hip_shard_loopback_verify(self.h, self.C.byref(n)), "shard_loopback_verify")
        return n.value

    def close(self):
        if self.h:
            self.L.x264hip_shard_close.argtypes = [self.C.c_void_p]
            self.L.x264hip_shard_close(self.h)
            self.h = None
            self.lookahead.h = None


def run_c_window_shard(torch, lib, rank, world, dev_index, cfg, dev_clip, transport, qp_offsets=False, vbv=False, passes=1):
    """One pass of ONE stream over `world` ranks through the C entry points: (outputs on rank 0 | None, seconds, stats, rc of serve()).
    dev_clip: the clip on rank 0's GPU ([F, H, W]); the other ranks receive the pictures chunk by chunk inside the timed region.
    passes > 1: the same clip again after x264hip_shard_reset (outputs: a list per pass)."""
    import time
    L = lib.load()
    F = dev_clip.shape[0] if dev_clip is not None else 0
    sh = CShard(L, cfg, transport, device=dev_index, max_frames=(F or cfg.get("_frames", 0)) + 4)
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs, rc = None, 0
        if rank == 0:
            every = []
            for k in range(passes):
                if k:
                    sh.reset()
                sh.put_frames([dev_clip[i].data_ptr() for i in range(F)], cfg["width"])
                outs = []
                while True:
                    o = sh.lookahead.get(True, qp_offsets, vbv)
                    if o is None:
                        break
                    outs.append(o)
                every.append(outs)
            if passes > 1:
                outs = every
            lib._ck(L.x264hip_synchronize(sh.ctx_handle()), "synchronize")
            lib._ck(sh.status(), "shard_status")  # a rank that failed anywhere fails the pass here at the latest
        else:
            rc = sh.serve()
        dt = time.perf_counter() - t0
        checks = sh.loopback_verify() if getattr(transport, "loopback", 0) else 0
        st = sh.stats()
        st["loopback_checks"] = checks
        st["bytes_l0_exchange"] = st["bytes_l0_received"]
        counters = np.zeros(16, np.uint64)
        import ctypes as C
        lib._ck(L.x264hip_counters(sh.ctx_handle(), counters.ctypes.data_as(C.c_void_p), 16), "counters")
        st.update(searches_here=int(counters[0]), cells_here=int(counters[5]), cells_on_demand=int(counters[7]), remote_fields_searched_here=int(counters[8]),
                  remote_maps_recomputed_here=int(counters[9]), maps_imported=int(counters[10]), cells_imported_ctx=int(counters[11]))
        return outs, dt, st, rc
    finally:
        sh.close()
