"""Test-only evaluation backend: plugs the CPU oracle into the product's host lookahead logic
(x264hip_lookahead_open_backend) so the host control flow can be tested without a GPU."""
import ctypes as C

import numpy as np

from oracle.oraclelib import Oracle, Weight as OWeight
from x264_amd import lib


class OracleBackend:
    def __init__(self, cfg, cost_mv=None, speculative=False):
        """speculative=True also offers the optional prefetch entries (as no-op / recording stubs), so that the host's
        ahead-of-time submission paths run on CPU: weight pairs announced through prefetch_weight_costs are remembered
        and every later weighted weight_cost request is checked against them."""
        self.cfg = cfg
        self.speculative = speculative
        self.announced = set()
        self.weighted_requests = self.weighted_predicted = 0
        self.o = Oracle(cfg["bit_depth"])
        mb_w, mb_h = (cfg["width"] + 15) // 16, (cfg["height"] + 15) // 16
        self.ocfg = self.o.make_cfg(mb_w, mb_h, me_method=cfg["la_me_method"], subpel_refine=cfg["la_subpel_refine"],
                                    me_range=cfg["me_range"], mv_range=cfg["mv_range"], subme=cfg["subme"],
                                    mbcmp_satd=cfg["mbcmp_satd"], fpelcmp_satd=cfg["fpelcmp_satd"],
                                    weighted_bipred=cfg["weighted_bipred"], aq_mode=cfg["aq_mode"], lam=cfg["lam"],
                                    bframe_bias=cfg["bframe_bias"], cost_mv=cost_mv, n_slices=cfg.get("lookahead_threads", 1),
                                    do_edges=cfg.get("do_edges", 1))
        self.slots = {}
        self.n_eval = 0
        self.requested = set()
        self.on_prefetch = None
        self.spec_used = self.searched_here = 0   # unweighted fields taken from the speculative store / searched by this backend itself
        # window shard (x264_amd/shard.py), mirroring the device context: fields known by name only (searched on the owner rank),
        # cell summaries imported from the owner, per-block maps fetched before MB-tree reads them
        self.on_mbtree = None
        self.cells_from_owner = self.cells_here = self.remote_fields_searched_here = self.maps_recomputed_here = 0
        self.variant_req = {}
        self.gop_hints = []
        self.verify_imported = False
        self.struct = lib.Backend(None, lib.FRAME_PUT_FN(self._put), lib.FRAME_STATS_FN(self._stats),
                                  lib.WEIGHT_COST_FN(self._wcost), lib.FRAME_COST_FN(self._cost),
                                  lib.PREFETCH_FN(self._prefetch) if speculative else lib.PREFETCH_FN(0),
                                  lib.MBTREE_FN(self._mbtree), lib.QP_OFFSETS_FN(self._qp), lib.PUT_BATCH_FN(0),
                                  lib.PREFETCH_WEIGHTS_FN(self._prefetch_weights) if speculative else lib.PREFETCH_WEIGHTS_FN(0),
                                  lib.RECALC_FN(self._recalc), lib.ROW_SATDS_FN(self._rows), lib.FRAME_PUT_YUV_FN(self._put_yuv), lib.ADD_QOFFS_FN(self._add_qoffs), lib.PUT_BATCH_YUV_FN(0),
                                  lib.GOP_HINT_FN(self._gop_hint) if speculative else lib.GOP_HINT_FN(0), lib.FLUSH_FN(0),
                                  lib.PREFETCH_WEIGHTS_FN(self._prefetch_weighted) if speculative else lib.PREFETCH_WEIGHTS_FN(0))
        self.announced_fields = set()
        self.weighted_searches = self.weighted_searches_predicted = 0

    def _gop_hint(self, user, anchor, period):
        self.gop_hints.append((anchor, period))
        return 0

    def _prefetch(self, user, slots, numbers, n):
        if self.on_prefetch is not None:  # window sharding (x264_amd/shard.py): the speculative searches of this chunk, spread over ranks
            self.on_prefetch([slots[i] for i in range(n)], [numbers[i] for i in range(n)])
        return 0

    def _prefetch_weighted(self, user, n, sf, sr, w):
        """x264hip_prefetch_weighted_fields: remembered, and every weighted first-trigger search is held against it (_cost)"""
        for i in range(n):
            assert w[i].on
            self.announced_fields.add((sf[i], sr[i], w[i].scale, w[i].denom, w[i].offset))
        return 0

    def _prefetch_weights(self, user, n, sf, sr, w):
        for i in range(n):
            self.announced.add((sf[i], sr[i], w[i].on, w[i].scale, w[i].denom, w[i].offset))
        return 0

    def _plane(self, addr, stride, w, h):
        dt = self.o.dtype
        buf = (C.c_char * (stride * h * np.dtype(dt).itemsize)).from_address(addr)
        return np.frombuffer(buf, dtype=dt).reshape(h, stride)[:, :w].copy()

    def _put_yuv(self, user, slot, luma, stride, cb, cr, cstride, is_device):
        c = self.cfg
        fmt = c.get("chroma_format", 1)
        cw = c["width"] if fmt == 3 else (c["width"] + 1) // 2
        ch = c["height"] if fmt >= 2 else (c["height"] + 1) // 2
        return self._put(user, slot, luma, stride, is_device, self._plane(cb, cstride, cw, ch), self._plane(cr, cstride, cw, ch))

    def _put(self, user, slot, luma, stride, is_device, cb=None, cr=None):
        c = self.cfg
        return self.put_array(slot, self._plane(luma, stride, c["width"], c["height"]), cb, cr)

    def put_array(self, slot, img, cb=None, cr=None):
        c = self.cfg
        pl = self.o.lowres_init(self.ocfg, img)
        inv, qp, s, ssd = self.o.aq_frame(img, self.ocfg.mb_w, self.ocfg.mb_h, c["aq_mode"], c["aq_strength"], cb, cr,
                                            chroma_format=c.get("chroma_format", 1))
        n = self.ocfg.mb_w * self.ocfg.mb_h
        self.slots[slot] = dict(planes=pl, inv=inv, sum=s, ssd=ssd, intra=self.o.intra_costs(self.ocfg, pl), fields={}, spec={}, maps={}, rows={}, remote=set(), ident={}, sums={}, sums_spare={}, map_remote={},
                                prop=np.zeros(n, np.uint16), qp_aq=qp.copy(), qp=qp.copy(), img=img, cb=cb, cr=cr)
        return 0

    def _stats(self, user, slot, psum, pssd):
        psum[0] = self.slots[slot]["sum"]
        pssd[0] = self.slots[slot]["ssd"]
        return 0

    def _wcost(self, user, sf, sr, w, out):
        wt = OWeight(w[0].on, w[0].scale, w[0].denom, w[0].offset) if w else None
        if w and w[0].on:
            self.weighted_requests += 1
            self.weighted_predicted += (sf, sr, w[0].on, w[0].scale, w[0].denom, w[0].offset) in self.announced
        out[0] = self.o.weight_cost(self.ocfg, self.slots[sf]["planes"], self.slots[sr]["planes"], wt, self.slots[sf]["intra"])
        return 0

    def _field(self, B, R, lst, dm1):
        """the field's data, searched here if this backend only knows it by name (window shard)"""
        key = (lst, dm1)
        if key in B["remote"]:
            B["remote"].discard(key)
            B["fields"][key] = self.o.search_field(self.ocfg, B["planes"], R["planes"])
            self.remote_fields_searched_here += 1
        return B["fields"][key]

    def _cost(self, user, s0, s1, sb, d0, d1, do_search, w, with_intra, ref1_valid, out):
        o, cfg = self.o, self.ocfg
        B, F0, F1 = self.slots[sb], self.slots[s0], self.slots[s1]
        self.n_eval += 1
        self.requested.add((int(d0), int(d1)))  # the cell classes the decisions asked for (held against x264hip_lookahead_classes)
        if d0 == 0 and d1 == 0:
            lc, rows, rows_i, co = o.cell(cfg, B["planes"], None, None, 128, None, None, None, None, None, B["intra"], B["inv"],
                                          bool(with_intra), alias_intra=True)
        else:
            if do_search[0]:
                wt, wplane = None, None
                if w and w[0].on:
                    wt = OWeight(w[0].on, w[0].scale, w[0].denom, w[0].offset)
                    wplane = o.weight_plane(cfg, F0["planes"][0], wt)
                    self.weighted_searches += 1
                    self.weighted_searches_predicted += (sb, s0, w[0].scale, w[0].denom, w[0].offset) in self.announced_fields
                if wt is None and ((0, d0 - 1) in B["spec"] or (0, d0 - 1) in B["remote"]):  # a speculative (possibly imported, possibly remote) field
                    if (0, d0 - 1) in B["spec"]:
                        B["fields"][(0, d0 - 1)] = B["spec"][(0, d0 - 1)]
                    self.spec_used += 1
                else:
                    B["fields"][(0, d0 - 1)] = o.search_field(cfg, B["planes"], F0["planes"], wt, wplane)
                    B["remote"].discard((0, d0 - 1))
                    # (a weighted field is a different field from the one cells of other ranks were evaluated with)
                    B["ident"][(0, d0 - 1)] = "unweighted" if wt is None else object()
                    self.searched_here += wt is None
            if d1 > 0 and do_search[1]:
                if (1, d1 - 1) in B["spec"] or (1, d1 - 1) in B["remote"]:
                    if (1, d1 - 1) in B["spec"]:
                        B["fields"][(1, d1 - 1)] = B["spec"][(1, d1 - 1)]
                    self.spec_used += 1
                else:
                    B["fields"][(1, d1 - 1)] = o.search_field(cfg, B["planes"], F1["planes"])
                    B["ident"][(1, d1 - 1)] = "unweighted"
                    self.searched_here += 1
            if d1 > 0:
                rq = self.variant_req.setdefault((d0, d1), [0, 0])
                rq[1 if ref1_valid else 0] += 1
            imp = B["sums"].get((d0, d1))
            ids = (B["ident"].get((0, d0 - 1)), B["ident"].get((1, d1 - 1)) if d1 else None,
                   F1["ident"].get((0, d0 + d1 - 1)) if d1 and ref1_valid else None)
            if d1 and not (imp is not None and imp["ids"] == ids and imp["with_l0"] == bool(ref1_valid)):
                imp = B["sums_spare"].get((d0, d1))  # the owner evaluated the cell both ways: the other half (without the list-1 reference's vectors)
            if imp is not None and imp["ids"] == ids and (not d1 or imp["with_l0"] == bool(ref1_valid)):
                # the owner rank's evaluation of this cell: sums and row sums only, the per-block map stays there
                self.cells_from_owner += 1
                rows, rows_i, co = imp["rows"], imp["rows_i"], imp["co"]
                if self.verify_imported:  # (debugging aid: the owner's sums against an evaluation of the cell here)
                    rem = (set(B["remote"]), set(F1["remote"]))
                    n_before = self.remote_fields_searched_here
                    fb, f1 = dict(B["fields"]), dict(F1["fields"])
                    m0, c0 = self._field(B, F0, 0, d0 - 1)
                    dsf = (d0 * 256 + (d0 + d1) // 2) // (d0 + d1)
                    if d1 > 0:
                        m1, c1 = self._field(B, F1, 1, d1 - 1)
                        r1 = self._field(F1, F0, 0, d0 + d1 - 1)[0] if ref1_valid else None
                        chk = o.cell(cfg, B["planes"], F0["planes"], F1["planes"], dsf, m0, c0, m1, c1, r1, B["intra"], B["inv"], True)
                    else:
                        chk = o.cell(cfg, B["planes"], F0["planes"], None, dsf, m0, c0, None, None, None, B["intra"], B["inv"], True)
                    B["remote"], F1["remote"] = rem
                    self.remote_fields_searched_here = n_before
                    B["fields"], F1["fields"] = fb, f1
                    same = (chk[3].cost_est, chk[3].cost_est_aq, chk[3].intra_mbs) == (co.cost_est, co.cost_est_aq, co.intra_mbs) and np.array_equal(chk[1], rows)
                    assert same, ("imported cell differs", sb, d0, d1, ref1_valid, chk[3].cost_est, co.cost_est, chk[3].intra_mbs, co.intra_mbs)
                B["map_remote"][(d0, d1)] = dict(s0=s0, s1=s1, with_l0=imp["with_l0"], spare=imp.get("spare", False))
                B["maps"].pop((d0, d1), None)
                lc = None
            else:
                self.cells_here += 1
                m0, c0 = self._field(B, F0, 0, d0 - 1)
                dsf = (d0 * 256 + (d0 + d1) // 2) // (d0 + d1)
                if d1 > 0:
                    m1, c1 = self._field(B, F1, 1, d1 - 1)
                    r1 = self._field(F1, F0, 0, d0 + d1 - 1)[0] if ref1_valid else None
                    lc, rows, rows_i, co = o.cell(cfg, B["planes"], F0["planes"], F1["planes"], dsf, m0, c0, m1, c1, r1, B["intra"],
                                                  B["inv"], bool(with_intra))
                else:
                    lc, rows, rows_i, co = o.cell(cfg, B["planes"], F0["planes"], None, dsf, m0, c0, None, None, None, B["intra"],
                                                  B["inv"], bool(with_intra))
                B["map_remote"].pop((d0, d1), None)
        if lc is not None:
            B["maps"][(d0, d1)] = lc
        if d0 or d1:
            B["rows"][(d0, d1)] = rows.copy()
        if with_intra:
            B["rows"][(0, 0)] = rows_i.copy()
        out[0].cost_est, out[0].cost_est_aq, out[0].intra_mbs = co.cost_est, co.cost_est_aq, co.intra_mbs
        out[0].intra_cost_est, out[0].intra_cost_est_aq = co.intra_cost_est, co.intra_cost_est_aq
        return 0

    def cell_map(self, sb, d0, d1):
        """the per-block map of an evaluated cell, recomputed here when only its sums came from the owner rank (what the device context
        does in ensure_cell_local)"""
        B = self.slots[sb]
        if (d0, d1) not in B["maps"]:
            info = B["map_remote"].pop((d0, d1))
            F0, F1 = self.slots[info["s0"]], self.slots[info["s1"]]
            m0, c0 = self._field(B, F0, 0, d0 - 1)
            dsf = (d0 * 256 + (d0 + d1) // 2) // (d0 + d1)
            if d1:
                m1, c1 = self._field(B, F1, 1, d1 - 1)
                r1 = self._field(F1, F0, 0, d0 + d1 - 1)[0] if info["with_l0"] else None
                lc = self.o.cell(self.ocfg, B["planes"], F0["planes"], F1["planes"], dsf, m0, c0, m1, c1, r1, B["intra"], B["inv"], False)[0]
            else:
                lc = self.o.cell(self.ocfg, B["planes"], F0["planes"], None, dsf, m0, c0, None, None, None, B["intra"], B["inv"], False)[0]
            B["maps"][(d0, d1)] = lc
            self.maps_recomputed_here += 1
        return B["maps"][(d0, d1)]

    def _mbtree(self, user, ops, n):
        L = self.o.lib
        cfg = self.ocfg
        nmb = cfg.mb_w * cfg.mb_h
        if self.on_mbtree is not None:  # window shard: fetch the maps the PROPAGATE steps read from their owner ranks
            self.on_mbtree([(ops[k].slot_b, ops[k].slot_p0, ops[k].slot_p1, ops[k].dist_p0, ops[k].dist_p1) for k in range(n) if ops[k].type == 1])
        for k in range(n):
            op = ops[k]
            B = self.slots[op.slot_b]
            if op.type == 3:      # X264HIP_MBT_SWAP: the two frames exchange their accumulators
                A = self.slots[op.slot_p0]
                B["prop"], A["prop"] = A["prop"], B["prop"]
            elif op.type == 4:    # X264HIP_MBT_RESET_QP
                B["qp"][:] = B["qp_aq"]
            elif op.type == 0:
                B["prop"][:] = 0
            elif op.type == 1:
                F0, F1 = self.slots[op.slot_p0], self.slots[op.slot_p1]
                lc = self.cell_map(op.slot_b, op.dist_p0, op.dist_p1)
                mvf = B.get("mv_only", {})  # vectors that arrived with a fetched map (their costs stayed with the owner)
                m0 = mvf[(0, op.dist_p0 - 1)] if (0, op.dist_p0 - 1) in B["remote"] else B["fields"][(0, op.dist_p0 - 1)][0]
                m1 = None
                if op.dist_p1 > 0:
                    m1 = mvf[(1, op.dist_p1 - 1)] if (1, op.dist_p1 - 1) in B["remote"] else B["fields"][(1, op.dist_p1 - 1)][0]
                pin = B["prop"].copy() if op.referenced else None
                L.or_mbtree_propagate.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_float]
                L.or_mbtree_propagate(cfg.mb_w, cfg.mb_h, B["intra"].ctypes.data, lc.ctypes.data, B["inv"].ctypes.data,
                                      pin.ctypes.data if pin is not None else None, m0.ctypes.data,
                                      m1.ctypes.data if m1 is not None else None, F0["prop"].ctypes.data,
                                      F1["prop"].ctypes.data if m1 is not None else None, op.bipred_weight, op.fps_factor)
            else:
                L.or_mbtree_finish.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_float, C.c_float]
                L.or_mbtree_finish(nmb, B["intra"].ctypes.data, B["inv"].ctypes.data, B["prop"].ctypes.data, B["qp_aq"].ctypes.data,
                                   B["qp"].ctypes.data, op.fps_factor_i, op.weightdelta, op.strength)
        return 0

    def _qp(self, user, slot, dst):
        q = self.slots[slot]["qp"]
        C.memmove(dst, q.ctypes.data, q.nbytes)
        return 0

    def _recalc(self, user, slot_b, d0, d1, use_aq, score):
        """slicetype_frame_cost_recalculate: rewrites the cell's row sums like the reference"""
        B = self.slots[slot_b]
        cfg = self.ocfg
        rows = np.zeros(cfg.mb_h, np.int32)
        fn = self.o.f("frame_cost_recalculate", C.c_int)
        q = B["qp_aq"] if use_aq else B["qp"]
        lc = self.cell_map(slot_b, d0, d1) if (d0 or d1) else B["intra"]  # lowres_costs[0][0] is the intra cost array itself (frame.c:283)
        score[0] = fn(cfg.mb_w, cfg.mb_h, lc.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p),
                      rows.ctypes.data_as(C.c_void_p))
        B["rows"][(d0, d1)] = rows
        return 0

    def _rows(self, user, slot, d0, d1, dst):
        r = self.slots[slot]["rows"][(d0, d1)]
        C.memmove(dst, r.ctypes.data, r.nbytes)
        return 0

    def _add_qoffs(self, user, slot, q):
        """x264_picture_t.prop.quant_offsets: the oracle's AQ of the stored picture again, with the offsets"""
        B, c = self.slots[slot], self.cfg
        n = self.ocfg.mb_w * self.ocfg.mb_h
        offs = np.ctypeslib.as_array(q, shape=(n,)).copy()
        inv, qp, _, _ = self.o.aq_frame(B["img"], self.ocfg.mb_w, self.ocfg.mb_h, c["aq_mode"], c["aq_strength"], B["cb"], B["cr"],
                                        chroma_format=c.get("chroma_format", 1), quant_offsets=offs)
        B["inv"], B["qp_aq"], B["qp"] = inv, qp.copy(), qp.copy()
        return 0


def _pack_mv(mv):
    return (mv[:, 0].astype(np.int32) & 0xFFFF) | (mv[:, 1].astype(np.int32) << 16)


def _unpack_mv(w):
    return np.ascontiguousarray(np.stack([(w & 0xFFFF).astype(np.uint16).view(np.int16), (w >> 16).astype(np.int16)], axis=1).astype(np.int16))


class OracleShardAdapter:
    """x264_amd.shard.WindowShard over the oracle backend (CPU tests of the multi-rank protocol): the same duck-typed interface as
    HipAdapter.  Fields live in the backend's speculative store, cells evaluated for other ranks in `owned`; on rank 0 imported
    summaries, fields known by name only and fetched maps are kept the way the device context keeps them."""

    def __init__(self, be, clip, own_ingest, dist=None, rank=0, world=1):
        from x264_amd.shard import Exchange
        self.be, self.clip, self.own_ingest = be, clip, own_ingest
        self.n_mb, self.mb_h, self.bframes = be.ocfg.mb_w * be.ocfg.mb_h, be.ocfg.mb_h, be.cfg["bframes"]
        self.exchange = Exchange(dist, rank, world)
        self.owned = {}
        self.owned_spare = {}

    def ingest(self, slot, number):
        for d in (self.owned, self.owned_spare):
            for k in [k for k in d if k[0] == slot]:
                del d[k]  # the slot holds another frame now
        if self.own_ingest:
            self.be.put_array(slot, self.clip[number])

    def classes(self):
        ns = self.bframes + 2
        cc = [0] * (ns * ns)
        for d0 in range(1, ns):
            for d1 in range(0, ns - d0):
                rq = self.be.variant_req.get((d0, d1), [0, 0])
                cc[d0 * ns + d1] = 3 if d1 else 1  # B cells: both ways in one pass (x264hip_cell_classes)
        return (1 << (self.bframes + 1)) - 1, (1 << (self.bframes + 1)) - 1, cc

    def search(self, reqs):
        for sb, sr, lst, dm1 in reqs:
            B, R = self.be.slots[sb], self.be.slots[sr]
            if (lst, dm1) not in B["spec"] and (lst, dm1) not in B["fields"]:
                B["spec"][(lst, dm1)] = self.be.o.search_field(self.be.ocfg, B["planes"], R["planes"])
                B["ident"][(lst, dm1)] = "unweighted"

    def _have(self, B, key):
        return B["spec"].get(key) or B["fields"][key]

    def export_fields(self, keys, out):
        import torch
        for i, (slot, lst, dm1) in enumerate(keys):
            mv, cost = self._have(self.be.slots[slot], (lst, dm1))
            out[i].copy_(torch.from_numpy(np.stack([_pack_mv(mv), cost.astype(np.int32)], axis=1)))

    def import_fields(self, keys, t, rows):
        a = t.numpy()
        for (slot, lst, dm1), i in zip(keys, rows):
            B = self.be.slots[slot]
            if ((lst, dm1) in B["spec"] or (lst, dm1) in B["fields"]) and (lst, dm1) not in B["remote"]:
                continue
            B["remote"].discard((lst, dm1))
            B["spec"][(lst, dm1)] = (_unpack_mv(a[i, :, 0]), np.ascontiguousarray(a[i, :, 1], np.int32))
            B["ident"].setdefault((lst, dm1), "unweighted")

    def _cell(self, c):
        sb, s0, s1, d0, d1, with_l0 = c
        be, o = self.be, self.be.o
        B, F0, F1 = be.slots[sb], be.slots[s0], be.slots[s1]
        m0, c0 = self._have(B, (0, d0 - 1))
        dsf = (d0 * 256 + (d0 + d1) // 2) // (d0 + d1)
        if d1:
            m1, c1 = self._have(B, (1, d1 - 1))
            r1 = self._have(F1, (0, d0 + d1 - 1))[0] if with_l0 else None
            return o.cell(be.ocfg, B["planes"], F0["planes"], F1["planes"], dsf, m0, c0, m1, c1, r1, B["intra"], B["inv"], True)
        return o.cell(be.ocfg, B["planes"], F0["planes"], None, dsf, m0, c0, None, None, None, B["intra"], B["inv"], True)

    def spec_cells(self, cells):
        for c in cells:
            if c[3] and (c[0], c[3], c[4]) not in self.owned and not self.be.on_prefetch:  # (rank 0 evaluates its own cells on demand)
                self.owned[(c[0], c[3], c[4])] = self._cell(tuple(c[:5]) + (c[5] & 1,))
                if c[5] & 2:  # both ways: the spare half is the evaluation without the list-1 reference's vectors
                    self.owned_spare[(c[0], c[3], c[4])] = self._cell(tuple(c[:5]) + (0,))

    def export_cells(self, cells, out):
        import torch
        for i, c in enumerate(cells):
            lc, rows, rows_i, co = (self.owned_spare if c[5] & 4 else self.owned)[(c[0], c[3], c[4])]
            row = np.zeros(8 + 2 * self.mb_h, np.int32)
            row[:5] = (co.cost_est, co.cost_est_aq, co.intra_mbs, co.intra_cost_est, co.intra_cost_est_aq)
            row[8:8 + self.mb_h], row[8 + self.mb_h:] = rows, rows_i
            out[i].copy_(torch.from_numpy(row))

    def import_cells(self, cells, t):
        from oracle.oraclelib import CellOut
        a = t.numpy()
        for i, (sb, s0, s1, d0, d1, flags) in enumerate(cells):
            B, F1 = self.be.slots[sb], self.be.slots[s1]
            spare, with_l0 = bool(flags & 4), bool(flags & 1) and not flags & 4
            co = CellOut()
            co.cost_est, co.cost_est_aq, co.intra_mbs, co.intra_cost_est, co.intra_cost_est_aq = (int(v) for v in a[i, :5])
            ids = (B["ident"].get((0, d0 - 1)), B["ident"].get((1, d1 - 1)) if d1 else None, F1["ident"].get((0, d0 + d1 - 1)) if d1 and with_l0 else None)
            B["sums_spare" if spare else "sums"][(d0, d1)] = dict(co=co, rows=a[i, 8:8 + self.mb_h].copy(), rows_i=a[i, 8 + self.mb_h:].copy(), with_l0=with_l0, ids=ids,
                                                                   spare=spare)

    def fields_remote(self, keys):
        for slot, number, lst, dm1 in keys:
            B = self.be.slots[slot]
            if (lst, dm1) in B["spec"] or (lst, dm1) in B["fields"] or (lst, dm1) in B["remote"]:
                continue
            B["remote"].add((lst, dm1))
            B["ident"][(lst, dm1)] = "unweighted"

    def cells_missing(self, cells):
        info = lambda c: self.be.slots[c[0]]["map_remote"].get((c[3], c[4]))  # noqa: E731
        return [0 if info(c) is None else 2 if info(c).get("spare") else 1 for c in cells]

    def export_map(self, cell, out):
        import torch
        sb, s0, s1, d0, d1 = cell[:5]
        B = self.be.slots[sb]
        lc = (self.owned_spare if len(cell) > 5 and cell[5] & 4 else self.owned)[(sb, d0, d1)][0]
        m = np.zeros((3, self.n_mb), np.int32)
        m[0] = lc
        m[1] = _pack_mv(self._have(B, (0, d0 - 1))[0])
        if d1:
            m[2] = _pack_mv(self._have(B, (1, d1 - 1))[0])
        out.copy_(torch.from_numpy(m))

    def import_map(self, cell, t):
        sb, s0, s1, d0, d1 = cell[:5]
        B = self.be.slots[sb]
        if (d0, d1) not in B["map_remote"]:
            return
        a = t.numpy()
        B["map_remote"].pop((d0, d1))
        B["maps"][(d0, d1)] = a[0].astype(np.uint16)
        mvf = B.setdefault("mv_only", {})
        if (0, d0 - 1) in B["remote"]:
            mvf[(0, d0 - 1)] = _unpack_mv(a[1])
        if d1 and (1, d1 - 1) in B["remote"]:
            mvf[(1, d1 - 1)] = _unpack_mv(a[2])
