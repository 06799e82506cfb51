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
        self.on_prefetch = None
        self.spec_used = self.searched_here = 0   # unweighted fields taken from the speculative store / searched by this backend itself
        self.struct = lib.Backend(None, lib.FRAME_PUT_FN(self._put), lib.FRAME_STATS_FN(self._stats),
                                  lib.WEIGHT_COST_FN(self._wcost), lib.FRAME_COST_FN(self._cost),
                                  lib.PREFETCH_FN(self._prefetch) if speculative else lib.PREFETCH_FN(0),
                                  lib.MBTREE_FN(self._mbtree), lib.QP_OFFSETS_FN(self._qp), lib.PUT_BATCH_FN(0),
                                  lib.PREFETCH_WEIGHTS_FN(self._prefetch_weights) if speculative else lib.PREFETCH_WEIGHTS_FN(0),
                                  lib.RECALC_FN(self._recalc), lib.ROW_SATDS_FN(self._rows), lib.FRAME_PUT_YUV_FN(self._put_yuv), lib.ADD_QOFFS_FN(self._add_qoffs), lib.PUT_BATCH_YUV_FN(0))

    def _prefetch(self, user, slots, numbers, n):
        if self.on_prefetch is not None:  # window sharding (x264_amd/shard.py): the speculative searches of this chunk, spread over ranks
            self.on_prefetch([slots[i] for i in range(n)], [numbers[i] for i in range(n)])
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
        self.slots[slot] = dict(planes=pl, inv=inv, sum=s, ssd=ssd, intra=self.o.intra_costs(self.ocfg, pl), fields={}, spec={}, maps={}, rows={},
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

    def _cost(self, user, s0, s1, sb, d0, d1, do_search, w, with_intra, ref1_valid, out):
        o, cfg = self.o, self.ocfg
        B, F0, F1 = self.slots[sb], self.slots[s0], self.slots[s1]
        self.n_eval += 1
        if d0 == 0 and d1 == 0:
            lc, rows, rows_i, co = o.cell(cfg, B["planes"], None, None, 128, None, None, None, None, None, B["intra"], B["inv"],
                                          bool(with_intra), alias_intra=True)
        else:
            if do_search[0]:
                wt, wplane = None, None
                if w and w[0].on:
                    wt = OWeight(w[0].on, w[0].scale, w[0].denom, w[0].offset)
                    wplane = o.weight_plane(cfg, F0["planes"][0], wt)
                if wt is None and (0, d0 - 1) in B["spec"]:  # a speculative (possibly imported) field, like the device's claimed fields
                    B["fields"][(0, d0 - 1)] = B["spec"][(0, d0 - 1)]
                    self.spec_used += 1
                else:
                    B["fields"][(0, d0 - 1)] = o.search_field(cfg, B["planes"], F0["planes"], wt, wplane)
                    self.searched_here += wt is None
            if d1 > 0 and do_search[1]:
                if (1, d1 - 1) in B["spec"]:
                    B["fields"][(1, d1 - 1)] = B["spec"][(1, d1 - 1)]
                    self.spec_used += 1
                else:
                    B["fields"][(1, d1 - 1)] = o.search_field(cfg, B["planes"], F1["planes"])
                    self.searched_here += 1
            m0, c0 = B["fields"][(0, d0 - 1)]
            dsf = (d0 * 256 + (d0 + d1) // 2) // (d0 + d1)
            if d1 > 0:
                m1, c1 = B["fields"][(1, d1 - 1)]
                r1 = F1["fields"][(0, d0 + d1 - 1)][0] if ref1_valid else None
                lc, rows, rows_i, co = o.cell(cfg, B["planes"], F0["planes"], F1["planes"], dsf, m0, c0, m1, c1, r1, B["intra"],
                                              B["inv"], bool(with_intra))
            else:
                lc, rows, rows_i, co = o.cell(cfg, B["planes"], F0["planes"], None, dsf, m0, c0, None, None, None, B["intra"],
                                              B["inv"], bool(with_intra))
        B["maps"][(d0, d1)] = lc
        if d0 or d1:
            B["rows"][(d0, d1)] = rows.copy()
        if with_intra:
            B["rows"][(0, 0)] = rows_i.copy()
        out[0].cost_est, out[0].cost_est_aq, out[0].intra_mbs = co.cost_est, co.cost_est_aq, co.intra_mbs
        out[0].intra_cost_est, out[0].intra_cost_est_aq = co.intra_cost_est, co.intra_cost_est_aq
        return 0

    def _mbtree(self, user, ops, n):
        L = self.o.lib
        cfg = self.ocfg
        nmb = cfg.mb_w * cfg.mb_h
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
                lc = B["maps"][(op.dist_p0, op.dist_p1)]
                m0 = B["fields"][(0, op.dist_p0 - 1)][0]
                m1 = B["fields"][(1, op.dist_p1 - 1)][0] if op.dist_p1 > 0 else None
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
        lc = B["maps"][(d0, d1)] if (d0 or d1) else B["intra"]  # lowres_costs[0][0] is the intra cost array itself (frame.c:283)
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


class OracleShardAdapter:
    """x264_amd.shard.WindowShard over the oracle backend (CPU tests of the multi-rank protocol): the same duck-typed interface as
    HipAdapter, fields kept in the backend's speculative store."""

    def __init__(self, be, clip, own_ingest):
        self.be, self.clip, self.own_ingest = be, clip, own_ingest
        self.n_mb, self.bframes = be.ocfg.mb_w * be.ocfg.mb_h, be.cfg["bframes"]

    def ingest(self, slot, number):
        if self.own_ingest:
            self.be.put_array(slot, self.clip[number])

    def classes(self):
        return (1 << (self.bframes + 1)) - 1, (1 << (self.bframes + 1)) - 1

    def search(self, reqs):
        for sb, sr, lst, dm1 in reqs:
            B, R = self.be.slots[sb], self.be.slots[sr]
            if (lst, dm1) not in B["spec"] and (lst, dm1) not in B["fields"]:
                B["spec"][(lst, dm1)] = self.be.o.search_field(self.be.ocfg, B["planes"], R["planes"])

    def export(self, keys):
        import torch
        out = np.zeros((len(keys), self.n_mb, 2), np.int32)
        for i, (slot, lst, dm1) in enumerate(keys):
            B = self.be.slots[slot]
            mv, cost = B["spec"].get((lst, dm1)) or B["fields"][(lst, dm1)]
            out[i, :, 0] = (mv[:, 0].astype(np.int32) & 0xFFFF) | (mv[:, 1].astype(np.int32) << 16)
            out[i, :, 1] = cost
        return torch.from_numpy(out)

    def import_(self, keys, t):
        a = t.numpy()
        for i, (slot, lst, dm1) in enumerate(keys):
            B = self.be.slots[slot]
            if (lst, dm1) in B["spec"] or (lst, dm1) in B["fields"]:
                continue
            w = a[i, :, 0]
            mv = np.stack([(w & 0xFFFF).astype(np.uint16).view(np.int16), (w >> 16).astype(np.int16)], axis=1).astype(np.int16)
            B["spec"][(lst, dm1)] = (np.ascontiguousarray(mv), np.ascontiguousarray(a[i, :, 1], np.int32))

    def finish(self, slots, numbers):
        pass
