#!/usr/bin/env python
"""bench.py -- lookahead frames/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the whole lookahead hot path (lowres + AQ + intra + motion searches + cost cells +
slice-type decision) over one clip of synthetic frames that is already resident in HBM when the timed
region starts.  Workload at N=1: BASELINE.json configs[1] -- 1920x1080 8-bit 4:2:0, --preset slow --me dia.
For N>1 every rank runs the same-sized, independent sequence segment (GOP-parallel sharding);
the only exchange is an RCCL all-gather of the per-frame cost summaries.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_search(cfg):
    """one (frame, list, distance) search reads the source plane once and the four half-pel
    planes of its reference once (5*S) and writes mv (4 B) + mv cost (4 B) per block."""
    mb_w, mb_h = (cfg["width"] + 15) // 16, (cfg["height"] + 15) // 16
    S = (8 * mb_w) * (8 * mb_h) * (1 if cfg["bit_depth"] == 8 else 2)
    return 5 * S + 8 * mb_w * mb_h


def primitives_bench(torch, libmod, cfg, iters=30):
    """SAD / SATD GB/s of the batched vtable primitives over all blocks of a frame pair, and lowres init GB/s."""
    import ctypes as C
    W, H = cfg["width"], cfg["height"]
    out = {}
    ctx = libmod.Context(W, H, bit_depth=8, max_frames=2, mv_range=cfg["mv_range"])
    try:
        g = torch.Generator(device="cuda").manual_seed(1)
        # one frame pair, and a 4x4 mosaic of frame pairs (265 MB: larger than the 256 MB Infinity Cache, so the
        # figure is an HBM rate and the launch overhead of a 3 us kernel does not dominate it)
        for tag, mul in (("", 1), ("_x16", 4)):
            Wp, Hp = (W // 16) * 16 * mul, (H // 16) * 16 * mul
            stride = Wp + 64
            fenc = torch.randint(0, 256, (Hp + 64, stride), dtype=torch.uint8, device="cuda", generator=g)
            ref = torch.randint(0, 256, (Hp + 64, stride), dtype=torch.uint8, device="cuda", generator=g)
            org = 32 * stride + 32
            for size_idx, size in ((0, 16), (3, 8), (6, 4)):
                bw, bh = Wp // size, Hp // size
                mv = torch.randint(-16, 17, (bw * bh, 2), dtype=torch.int16, device="cuda", generator=g)
                res = torch.zeros(bw * bh, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                for satd in (0, 1):
                    def run():
                        ctx.pixel_cmp_batch(satd, size_idx, fenc.data_ptr() + org, ref.data_ptr() + org, stride, bw, bh, mv.data_ptr(), res.data_ptr())
                    run(); ctx.synchronize()
                    dt = None
                    for _ in range(3):  # best of three timed loops: a one-off stall (clock ramp, another process on the box) would otherwise be the figure
                        t0 = time.perf_counter()
                        for _ in range(iters):
                            run()
                        ctx.synchronize()
                        d = (time.perf_counter() - t0) / iters
                        dt = d if dt is None else min(dt, d)
                    nbytes = bw * bh * (2 * size * size + 4)  # SURVEY 8(d) per-block form: both blocks + the result
                    out["%s_%dx%d%s_GBps" % ("satd" if satd else "sad", size, size, tag)] = round(nbytes / dt / 1e9, 1)
            del fenc, ref
        # The same blocks as SIXTEEN separate frame pairs in one launch of the public multi-pair entry (x264hip_pixel_cmp_batch_multi: pair =
        # blockIdx.z): what a caller with a window of frames has -- the single 4K pair above is a launch of a few microseconds
        NP = 16
        Wp, Hp = (W // 16) * 16, (H // 16) * 16
        stride = Wp + 64
        org = 32 * stride + 32
        fencs = [torch.randint(0, 256, (Hp + 64, stride), dtype=torch.uint8, device="cuda", generator=g) for _ in range(NP)]
        refs = [torch.randint(0, 256, (Hp + 64, stride), dtype=torch.uint8, device="cuda", generator=g) for _ in range(NP)]
        for size_idx, size in ((0, 16), (3, 8), (6, 4)):
            bw, bh = Wp // size, Hp // size
            mvs = [torch.randint(-16, 17, (bw * bh, 2), dtype=torch.int16, device="cuda", generator=g) for _ in range(NP)]
            ress = [torch.zeros(bw * bh, dtype=torch.int32, device="cuda") for _ in range(NP)]
            torch.cuda.synchronize()
            fp_, rp_, mp_, op_ = ([t.data_ptr() + o_ for t in lst] for lst, o_ in ((fencs, org), (refs, org), (mvs, 0), (ress, 0)))
            for satd in (0, 1):
                def run():
                    ctx.pixel_cmp_batch_multi(satd, size_idx, fp_, rp_, stride, bw, bh, mp_, op_)
                run(); ctx.synchronize()
                dt = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(iters):
                        run()
                    ctx.synchronize()
                    d = (time.perf_counter() - t0) / iters
                    dt = d if dt is None else min(dt, d)
                out["%s_%dx%d_16pairs_GBps" % ("satd" if satd else "sad", size, size)] = round(NP * bw * bh * (2 * size * size + 4) / dt / 1e9, 1)
        del fencs, refs, mvs, ress
        # hpel_filter (first piece) over a 4K plane: W*H read + 3*W*H written
        hs = W + 64
        src = torch.randint(0, 256, (H + 16, hs), dtype=torch.uint8, device="cuda", generator=g)
        dst = torch.empty((3, H + 16, hs), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ho = 8 * hs + 32
        def run_hpel():
            ctx.hpel_filter(dst[0].data_ptr() + ho, dst[1].data_ptr() + ho, dst[2].data_ptr() + ho, src.data_ptr() + ho, hs, W, H)
        run_hpel(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            run_hpel()
        ctx.synchronize()
        out["hpel_filter_GBps"] = round(4 * W * H * iters / (time.perf_counter() - t0) / 1e9, 1)
        # eight such planes in one launch (x264hip_hpel_filter_multi)
        srcs = [src] + [torch.randint(0, 256, (H + 16, hs), dtype=torch.uint8, device="cuda", generator=g) for _ in range(7)]
        dsts = [dst] + [torch.empty((3, H + 16, hs), dtype=torch.uint8, device="cuda") for _ in range(7)]
        torch.cuda.synchronize()
        hp_ = [[d_[k].data_ptr() + ho for d_ in dsts] for k in range(3)] + [[s_.data_ptr() + ho for s_ in srcs]]
        def run_hpel_multi():
            ctx.hpel_filter_multi(hp_[0], hp_[1], hp_[2], hp_[3], hs, W, H)
        run_hpel_multi(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            run_hpel_multi()
        ctx.synchronize()
        out["hpel_filter_8planes_GBps"] = round(8 * 4 * W * H * iters / (time.perf_counter() - t0) / 1e9, 1)
        del src, dst, srcs, dsts
        # the same over an 8K plane: a 4K plane is ~15 us of work, of which ~8 us are the launch and the first loads / last stores of a wave
        W8, H8 = 2 * W, 2 * H
        hs = W8 + 64
        src = torch.randint(0, 256, (H8 + 16, hs), dtype=torch.uint8, device="cuda", generator=g)
        dst = torch.empty((3, H8 + 16, hs), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ho = 8 * hs + 32
        def run_hpel8():
            ctx.hpel_filter(dst[0].data_ptr() + ho, dst[1].data_ptr() + ho, dst[2].data_ptr() + ho, src.data_ptr() + ho, hs, W8, H8)
        run_hpel8(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            run_hpel8()
        ctx.synchronize()
        out["hpel_filter_8k_GBps"] = round(4 * W8 * H8 * iters / (time.perf_counter() - t0) / 1e9, 1)
        del src, dst
        # a 4K plane of 10-bit samples (2 bytes each): the high-bit-depth streaming kernel
        try:
            ctx10 = libmod.Context(W, H, bit_depth=10, max_frames=2, mv_range=cfg["mv_range"])
            try:
                hs = W + 64
                src = torch.randint(0, 1024, (H + 16, hs), dtype=torch.int16, device="cuda", generator=g)
                dst = torch.empty((3, H + 16, hs), dtype=torch.int16, device="cuda")
                torch.cuda.synchronize()
                ho = (8 * hs + 32) * 2
                def run_hpel10():
                    ctx10.hpel_filter(dst[0].data_ptr() + ho, dst[1].data_ptr() + ho, dst[2].data_ptr() + ho, src.data_ptr() + ho, hs, W, H)
                run_hpel10(); ctx10.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    run_hpel10()
                ctx10.synchronize()
                out["hpel_filter_10bit_GBps"] = round(2 * 4 * W * H * iters / (time.perf_counter() - t0) / 1e9, 1)
                del src, dst
            finally:
                ctx10.close()
        except Exception as e:  # pragma: no cover
            out["hpel_filter_10bit_error"] = repr(e)
        # frame-level sub4x4_dct + quant_4x4 (first piece): 2*W*H read, 2*W*H (int16 coefficients) + W*H/16 written
        fe = torch.randint(0, 256, (H, W), dtype=torch.uint8, device="cuda", generator=g)
        fp = torch.randint(0, 256, (H, W), dtype=torch.uint8, device="cuda", generator=g)
        co = torch.empty((H // 4, W // 4, 16), dtype=torch.int16, device="cuda")
        nzb = torch.empty((H // 4, W // 4), dtype=torch.uint8, device="cuda")
        mfq = np.full(16, 13107, np.uint16); bq = np.full(16, 21845, np.uint16)
        torch.cuda.synchronize()
        def run_dq():
            ctx.frame_dct_quant4x4(fe.data_ptr(), W, fp.data_ptr(), W, W, H, mfq, bq, co.data_ptr(), nzb.data_ptr())
        run_dq(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            run_dq()
        ctx.synchronize()
        out["frame_dct_quant4x4_GBps"] = round((4 * W * H + W * H // 16) * iters / (time.perf_counter() - t0) / 1e9, 1)
        # eight plane pairs in one launch (x264hip_frame_dct_quant4x4_multi)
        fes = [fe] + [torch.randint(0, 256, (H, W), dtype=torch.uint8, device="cuda", generator=g) for _ in range(7)]
        fps = [fp] + [torch.randint(0, 256, (H, W), dtype=torch.uint8, device="cuda", generator=g) for _ in range(7)]
        cos = [co] + [torch.empty((H // 4, W // 4, 16), dtype=torch.int16, device="cuda") for _ in range(7)]
        nzs = [nzb] + [torch.empty((H // 4, W // 4), dtype=torch.uint8, device="cuda") for _ in range(7)]
        torch.cuda.synchronize()
        dq_ = [[t.data_ptr() for t in lst] for lst in (fes, fps, cos, nzs)]
        def run_dq_multi():
            ctx.frame_dct_quant4x4_multi(dq_[0], W, dq_[1], W, W, H, mfq, bq, dq_[2], dq_[3])
        run_dq_multi(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            run_dq_multi()
        ctx.synchronize()
        out["frame_dct_quant4x4_8planes_GBps"] = round(8 * (4 * W * H + W * H // 16) * iters / (time.perf_counter() - t0) / 1e9, 1)
        del fe, fp, co, nzb, fes, fps, cos, nzs
        # the build's own copy kernel over 512 MB (read + write counted): the measured HBM rate next to the 8 TB/s vendor peak
        a = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
        b = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.device_copy(b.data_ptr(), a.data_ptr(), a.numel()); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.device_copy(b.data_ptr(), a.data_ptr(), a.numel())
        ctx.synchronize()
        out["device_copy_GBps"] = round(2 * a.numel() * 10 / (time.perf_counter() - t0) / 1e9, 1)
        del a, b
        # main-encode motion search: every 16x16 macroblock of a 1920x1072 frame pair searched in one batch, searches/s.
        # a wave per request for every method: block costs across the wave, 64 candidates per step in the exhaustive scans (me_range 16)
        try:
            mw, mh, padh, padv, mvr = 1920, 1072, 32, 32, 64
            pw, ph = mw + 2 * padh, mh + 2 * padv
            base = torch.rand((1, 1, mh + 16, mw + 16), generator=g, device="cuda")
            base = torch.nn.functional.avg_pool2d(base, 5, stride=1, padding=2)[0, 0]
            base = (base - base.min()) / float(base.max() - base.min()) * 255.0
            fenc_l = (base[3:3 + mh, 2:2 + mw] + torch.randint(-3, 4, (mh, mw), generator=g, device="cuda")).clamp(0, 255).to(torch.uint8).contiguous()
            ref_l = (base[:mh, :mw] + torch.randint(-3, 4, (mh, mw), generator=g, device="cuda")).clamp(0, 255).to(torch.uint8).contiguous()
            planes = torch.zeros((4, ph, pw), dtype=torch.uint8, device="cuda")
            integ = torch.zeros((2 * ph, pw), dtype=torch.int16, device="cuda")
            table, centre = libmod.cost_mv_table(mvr, 1)
            cmv = torch.from_numpy(table.view(np.int16)).cuda()
            torch.cuda.synchronize()
            org = padv * pw + padh
            ctx.frame_filter(ref_l.data_ptr(), mw, mw, mh, [planes[k].data_ptr() + org for k in range(4)], pw, padh, padv, integ.data_ptr(), integ.data_ptr() + 2 * ph * pw)
            ctx.synchronize()
            mbw, mbh = mw // 16, mh // 16
            for name, method in (("esa", 3), ("tesa", 4), ("umh", 2), ("hex", 1), ("dia", 0)):
                reqs = []
                for my in range(mbh):
                    for mx in range(mbw):
                        q = libmod.MeRequest()
                        q.i_pixel, q.me_method, q.subpel_refine, q.me_range, q.mbcmp_satd, q.fpelcmp_satd = 0, method, 7, 16, 1, int(method == 4)
                        q.x, q.y = 16 * mx, 16 * my
                        fm = 4 * mvr
                        smin = [max(4 * (-16 * mx - 24), -fm), max(4 * (-16 * my - 24), -fm)]
                        smax = [min(4 * (16 * (mbw - mx - 1) + 24), fm - 1), min(4 * (16 * (mbh - my - 1) + 24), fm - 1)]
                        for k in range(2):
                            q.mvp[k] = 0
                            q.spel_min[k], q.spel_max[k] = smin[k], smax[k]
                            q.lim_min[k], q.lim_max[k] = (smin[k] >> 2) + 6, (smax[k] >> 2) - 6
                        q.n_mvc = 0
                        reqs.append(q)
                reqs = reqs * 4  # every macroblock four times in one batch (32 160 requests): the fixed cost of a call (table upload, launch, read-back) is amortised
                n_req = len(reqs)
                args_ = (ctx.me_requests(reqs), fenc_l.data_ptr(), mw, [planes[k].data_ptr() + org for k in range(4)], pw, integ.data_ptr() + org * 2, ph * pw, cmv.data_ptr() + 2 * centre)
                ctx.me_search_batch(*args_)
                t_call = dev_ms = None
                for _ in range(3):  # best of three calls, call time and device time each
                    t0 = time.perf_counter()
                    ctx.me_search_batch(*args_)
                    d = time.perf_counter() - t0
                    t_call = d if t_call is None else min(t_call, d)
                    m = ctx.last_search_ms()[0]
                    dev_ms = m if dev_ms is None else min(dev_ms, m)
                # device rate: the batch's kernels between HIP events on the context's stream, requests and planes resident (like `value`);
                # call rate: the whole C call -- request table translated and uploaded, kernels, results read back
                # (key names: `_searches_per_s` has been the rate of the whole call since round 2 -- in round 3 it briefly named the device
                # rate --, the device rate has its own key)
                out["me_full_%s_16x16_device_searches_per_s" % name] = round(n_req / (dev_ms * 1e-3))
                out["me_full_%s_16x16_searches_per_s" % name] = round(n_req / t_call)
                # the same table resident on the device (x264hip_me_search_batch_dev): the call enqueues, the wait is x264hip_synchronize
                import ctypes as _C
                raw = np.frombuffer(bytes(_C.string_at(_C.addressof(args_[0]), _C.sizeof(args_[0]))), dtype=np.uint8)
                rq_dev = torch.from_numpy(raw.copy()).cuda()
                res_dev = torch.zeros((n_req, 4), dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                t_dev = None
                for _ in range(4):
                    t0 = time.perf_counter()
                    ctx.me_search_batch_dev(n_req, rq_dev.data_ptr(), *args_[1:], method, 16, res_dev.data_ptr())
                    ctx.synchronize()
                    d = time.perf_counter() - t0
                    t_dev = d if t_dev is None else min(t_dev, d)
                out["me_full_%s_16x16_resident_call_searches_per_s" % name] = round(n_req / t_dev)
            out["me_full_note"] = ("me_full_*_searches_per_s = the whole x264hip_me_search_batch call (request table translated and uploaded, kernels, read-back), "
                                   "32 160 requests (116 bytes each over PCIe); me_full_*_resident_call_searches_per_s = x264hip_me_search_batch_dev + x264hip_synchronize, table and results on the device; me_full_*_device_searches_per_s = the kernels between HIP events.  In round 3's line the unsuffixed key was the "
                                   "device rate and *_call_searches_per_s the call rate")
            del planes, integ, fenc_l, ref_l
        except Exception as e:  # pragma: no cover
            out["me_full_error"] = repr(e)
        luma = torch.randint(0, 256, (H, W), dtype=torch.uint8, device="cuda", generator=g)
        torch.cuda.synchronize()
        ctx.frame_put(0, None, device_ptr=luma.data_ptr(), stride=W); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            ctx.frame_put(0, None, device_ptr=luma.data_ptr(), stride=W)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / iters
        S = (8 * ctx.mb_w) * (8 * ctx.mb_h)
        out["frame_ingest_GBps"] = round((W * H + 4 * S) / dt / 1e9, 1)  # lowres init bytes; AQ + intra ride along
    finally:
        ctx.close()
    return out


def cpu_baseline(cfg, frames, preset, opts, lookahead_threads=None):
    """Reference C path (oracle/_ref, built from /root/reference) timed on this box's host cores;
    falls back to the CPU port (oracle restatement behind the host logic) when the reference build is absent.
    lookahead_threads: None = the configuration as given (1 thread for the parity configuration); N > 1 = the reference's own
    lookahead threading (i_lookahead_threads bands of a frame searched concurrently, encoder/encoder.c:1273-1300,
    slicetype.c:917-944) -- a different configuration as far as results go (bands do not see each other's vectors)."""
    n = len(frames)
    try:
        from oracle import refharness
        if refharness.available(cfg["bit_depth"]):
            nt = cfg["lookahead_threads"]
            if lookahead_threads and lookahead_threads > 1:
                nt = lookahead_threads
                opts += ",threads=%d,sync-lookahead=0,lookahead-threads=%d" % (max(nt, 2), nt)
            r = refharness.Ref(cfg["width"], cfg["height"], preset, opts=opts, bit_depth=cfg["bit_depth"])
            try:
                res = r.lookahead_run(frames)
            finally:
                r.close()
            return dict(value=round(n / res["seconds"], 2), unit="frames/s", cores=nt, kind="reference", host_cores=os.cpu_count(),
                        sample="%d frames %dx%d, reference C path (--disable-asm, no AVX2: no assembler in the build image), "
                               "%d lookahead thread(s); lowres init + lookahead only, AQ excluded (%.2f s)" %
                               (n, cfg["width"], cfg["height"], nt, res["seconds"]))
    except Exception as e:  # pragma: no cover
        print("cpu_baseline: reference unavailable (%s), using the port" % e, file=sys.stderr)
    if lookahead_threads and lookahead_threads > 1:
        return None  # the port is single-threaded
    from tests.oracle_backend import OracleBackend
    from x264_amd import lib
    be = OracleBackend(cfg)
    la = lib.Lookahead(cfg, backend=be.struct)
    try:
        t0 = time.perf_counter()
        la.run(frames)
        dt = time.perf_counter() - t0
    finally:
        la.close()
    return dict(value=round(n / dt, 2), unit="frames/s", cores=1, kind="port", sample="%d frames %dx%d, oracle restatement, 1 thread" % (n, cfg["width"], cfg["height"]))


def make_clip_device(torch, W, H, F, seed, bit_depth=8, scene_cuts=(), pan=(5, 3), noise=3, texture=0.18, fade=None, still=None):
    """The synthetic recipe of x264_amd/synth.py (smooth random field, per-frame pan, +-noise, inversion at scene cuts) generated
    on the device with torch: used for the 4K workload, where the numpy generator would take longer than the benchmark."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    fh, fw = H + 128 + 8, W + 256 + 8
    field = torch.rand((1, 1, fh, fw), generator=g, device="cuda", dtype=torch.float32)
    for _ in range(4):
        field = torch.nn.functional.avg_pool2d(field, 5, stride=1, padding=2, count_include_pad=False)
    field = field[0, 0]
    field = (field - field.min()) / max(float(field.max() - field.min()), 1e-9)
    field = (1.0 - texture) * field + texture * torch.rand(field.shape, generator=g, device="cuda")
    field = 16.0 + field * 219.0
    cuts = sorted(set(int(c) for c in scene_cuts))
    scale, maxv = 1 << (bit_depth - 8), (1 << bit_depth) - 1
    out = torch.empty((F, H, W), dtype=torch.uint8 if bit_depth == 8 else torch.int16, device="cuda")
    from x264_amd.synth import pan_offsets
    offs = pan_offsets(F, pan, still)
    for i in range(F):
        dx, dy = offs[i]
        img = field[dy:dy + H, dx:dx + W]
        if sum(1 for c in cuts if c <= i) & 1:
            img = 255.0 - img
        if fade is not None:  # (start, length, gain_end, offset_end): linear fade, held afterwards (exercises weighted prediction)
            t = min(max((i - fade[0] + 1) / float(fade[1]), 0.0), 1.0)
            img = img * (1.0 + (fade[2] - 1.0) * t) + fade[3] * t
        img = img + torch.randint(-noise, noise + 1, img.shape, generator=g, device="cuda")
        out[i] = torch.clamp(torch.round(img * scale), 0, maxv).to(out.dtype)
    return out


def outputs_signature(outs, nb):
    """(frame, type) in coded order + every cost cell: what two runs of the same clip must agree on"""
    return [(o.frame, o.type, tuple(o.cost_est[i][j] for i in range(nb) for j in range(nb))) for o in outs]


class Workload:
    """S independent GOP segments of F frames resident in HBM on this rank's GPU, one lookahead context (and host thread) each."""

    def __init__(self, torch, lib, shard, cfg, dev_index, rank, S, F, seg_dev, paced, dist=None, backend="nccl", world=1):
        import concurrent.futures
        self.torch, self.lib, self.shard, self.cfg, self.rank, self.S, self.F, self.world = torch, lib, shard, cfg, rank, S, F, world
        self.W = cfg["width"]
        self.seg_dev = seg_dev
        self.seg_ptrs = [[dv[i].data_ptr() for i in range(F)] for dv in seg_dev]
        self.las = [lib.Lookahead(cfg, device=dev_index, max_frames=F + 4) for _ in range(S)]
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=S) if S > 1 else None
        self.paced = paced
        self.dist, self.backend = dist, backend
        self.gathered = None

    def run_segment(self, sgi, n_steps, paced):
        """n_steps passes of segment slot sgi, back to back.  The segment slots of a GPU are independent pipelines: they do
        not wait for each other between steps, so one slot's decision phases drift against the other's kernel phases
        instead of being re-aligned at every step (every step of every slot still completes inside the timed region)."""
        la = self.las[sgi]
        summaries, outs = [], None
        for k in range(n_steps):
            la.reset()
            outs = la.run_frames(self.seg_ptrs[sgi], stride=self.W, paced=paced)  # one C call per pass (x264hip_lookahead_run_frames)
            assert len(outs) == self.F
            summaries.append(self.shard.summarize(outs, (self.rank * self.S + sgi) * self.F))
        return outs, summaries

    def run_steps(self, n_steps, paced=None):
        """returns the outputs of the last step of every segment slot"""
        paced = self.paced if paced is None else paced
        if n_steps <= 0:
            return None
        if self.pool is None:
            res = [self.run_segment(0, n_steps, paced)]
        else:
            res = list(self.pool.map(lambda sgi: self.run_segment(sgi, n_steps, paced), range(self.S)))  # ctypes calls release the GIL
        for k in range(n_steps):
            host = np.concatenate([r[1][k] for r in res])
            if self.dist is not None:
                # the only exchange of the path: per-frame summaries (16 B per frame), one all_gather per step
                self.gathered = self.shard.gather_summaries(host, self.dist, device="cuda" if self.backend == "nccl" else None)
        return [r[0] for r in res]

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, steps, warmup, paced=None):
        self.run_steps(warmup, paced)
        self.barrier()
        t0 = time.perf_counter()
        outs = self.run_steps(steps, paced)
        self.barrier()
        return time.perf_counter() - t0, outs

    def close(self):
        for la in self.las:
            la.close()
        if self.pool is not None:
            self.pool.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=160, help="frames per step and per GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--preset", default="slow")
    ap.add_argument("--bit-depth", type=int, default=8, choices=(8, 10))
    ap.add_argument("--me", default="dia")
    ap.add_argument("--me-range", type=int, default=0)
    ap.add_argument("--threads", type=int, default=1,
                    help="x264 --threads of the mirrored configuration: > 1 turns on the reference's automatic lookahead bands "
                         "(i_lookahead_threads, encoder.c:1273-1300); 1 = the --threads 1 parity configuration")
    ap.add_argument("--paced", action="store_true", help="encoder-paced put/get instead of the deep-prefetch batch")
    ap.add_argument("--inflight", type=int, default=8,
                    help="independent GOP segments in flight per GPU (one host thread + one context each): the decisions of one "
                         "segment overlap the device work of the other")
    ap.add_argument("--shard", default="segments", choices=("segments", "window"),
                    help="N > 1: 'segments' = an independent GOP segment per rank (weak scaling, the default line); 'window' = ONE stream whose "
                         "lookahead window is sharded over the ranks (SURVEY 8e / BASELINE configs[3], strong scaling) as the primary value")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-primitives", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the additional lines (paced figure, 4K configs[2], threaded CPU baseline)")
    ap.add_argument("--no-check", action="store_true", help="skip the untimed verification pass in the other batching mode (profiling runs only: "
                                                              "its launches would mix into the per-kernel counters)")
    ap.add_argument("--device-clip", action="store_true", help="generate the clips on the device (no CPU baseline possible)")
    ap.add_argument("--cpu-frames", type=int, default=96)
    args = ap.parse_args()

    # Each context uses two streams (searches/cells and MB-tree).  The HIP runtime multiplexes a process's streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4): with two contexts in flight both MB-tree streams land on the same
    # queue and serialise behind each other, so give every stream its own queue.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(2 * max(1, args.inflight) + 2))
    import torch
    from x264_amd import lib, shard
    from x264_amd.synth import make_clip

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # --gpus N is a claim about the launch (python -m torch.distributed.run --nproc-per-node N ...): a line that says n_gpus = N must
        # come from N ranks
        print("bench.py: --gpus %d but WORLD_SIZE is %d (launch with python -m torch.distributed.run --nnodes=1 --nproc-per-node %d ...)" %
              (args.gpus, world, args.gpus), file=sys.stderr)
        raise SystemExit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the lookahead path has no CPU fallback")
    # one process per GPU; the modulo only matters for the single-GPU test rig (X264HIP_BENCH_BACKEND=gloo)
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dist = None
    backend = os.environ.get("X264HIP_BENCH_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            import datetime
            # (a collective that never completes ends the run after ten minutes instead of hanging it)
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=datetime.timedelta(minutes=10))  # RCCL over xGMI
        else:
            import datetime
            dist.init_process_group(backend, timeout=datetime.timedelta(minutes=10))
        if dist.get_world_size() != args.gpus:
            print("bench.py: the process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus), file=sys.stderr)
            raise SystemExit(2)

    W, H, F = args.width, args.height, args.frames
    over = dict(me=args.me, threads=args.threads)
    if args.me_range:
        over["me_range"] = args.me_range
    for kv in filter(None, os.environ.get("X264HIP_BENCH_CFG", "").split(",")):  # experiments only, e.g. mb_tree=0: the line is then NOT the BASELINE workload
        k, v = kv.split("=")
        over[k] = int(v)
    cfg = lib.la_config(W, H, args.preset, bit_depth=args.bit_depth, **over)
    nb = cfg["bframes"] + 2
    S = max(1, args.inflight)
    # every (rank, segment) gets its own part of the synthetic sequence (different seed = different content)
    seg_frames, seg_dev = [], []
    # the camera all but stops under the fade (x264_amd/synth.py pan_offsets): the lookahead's weight analysis then keeps its weights and
    # the weighted searches are part of the timed region.  X264HIP_BENCH_NO_STILL=1: the clip of rounds 1-5 (the fade under a fast pan,
    # no weight ever kept) for comparisons with their figures
    still = None if os.environ.get("X264HIP_BENCH_NO_STILL") else (2 * F // 3 - 2, 16)
    for sgi in range(S):
        if args.device_clip or sgi > 0:
            # only the first segment is generated on the host (the CPU baseline runs on it); the others come from the same recipe on the device
            dv = make_clip_device(torch, W, H, F, 100 + rank * S + sgi, args.bit_depth, scene_cuts=(F // 3, F // 3 + 47), fade=(2 * F // 3, 10, 0.6, 12), still=still)
            fr = None
        else:
            fr = make_clip(W, H, F, seed=100 + rank * S + sgi, bit_depth=args.bit_depth, scene_cuts=(F // 3, F // 3 + 47), fade=(2 * F // 3, 10, 0.6, 12),
                           pan=(5, 3), still=still)
            dv = torch.from_numpy(fr).cuda(dev_index)
        seg_frames.append(fr); seg_dev.append(dv)
    frames = seg_frames[0]
    torch.cuda.synchronize()
    wl = Workload(torch, lib, shard, cfg, dev_index, rank, S, F, seg_dev, args.paced, dist, backend, world)
    las = wl.las

    # bring the clocks up before anything is timed: a fresh process starts on an idle device and W may be small (half a second of
    # device copies; no lookahead work, and outside the timed region like the warm-up steps themselves)
    if not os.environ.get("X264HIP_BENCH_NO_PREHEAT"):
        pa = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
        pb = torch.empty_like(pa)
        t_end = time.perf_counter() + 0.5
        while time.perf_counter() < t_end:
            for _ in range(8):
                pb.copy_(pa)
            torch.cuda.synchronize()
        del pa, pb
    wl.run_steps(args.warmup)
    wl.barrier()
    for la in las:
        lib.search_profile(la.L, la.ctx_handle(), 1)
    t0 = time.perf_counter()
    outs_all = wl.run_steps(args.steps)
    wl.barrier()
    dt = time.perf_counter() - t0
    outs = outs_all[0]
    prof_ms = prof_launches = prof_searches = 0
    la_stats = np.zeros(8, np.uint64)
    dev_counters = np.zeros(16, np.uint64)
    wspec = np.zeros(4, np.uint64)
    for la in las:
        wb = np.zeros(4, np.uint64)
        lib._ck(la.L.x264hip_weighted_stats(la.ctx_handle(), wb.ctypes.data_as(ctypes.c_void_p)), "weighted_stats")
        wspec += wb
        cbuf = np.zeros(16, np.uint64)
        lib._ck(la.L.x264hip_counters(la.ctx_handle(), cbuf.ctypes.data_as(ctypes.c_void_p), 16), "counters")
        dev_counters += cbuf
        ms_, nl_, ns_ = lib.search_profile(la.L, la.ctx_handle(), 0)
        prof_ms += ms_; prof_launches += nl_; prof_searches += ns_
        la_stats += la.stats()
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    types = "".join("?IiPbB"[o.type] for o in sorted(outs, key=lambda o: o.frame))
    if dist is not None:
        g = wl.gathered
        assert g.shape == (world * S * F, 4) and sorted(g[:, 0].tolist()) == list(range(world * S * F)), "gathered summaries incomplete"

    # ---- what was timed is what the reference would decide: one untimed pass of every segment in the other batching mode
    # (encoder-paced when the timed passes were deep-prefetch batches and vice versa) must give the same frame types and the
    # same cost cells, frame for frame; its wall time is the figure of the other mode.
    # the search kernel alone: one context, nothing else in flight (what the rocprofv3 counter passes in profiles/ measure as well);
    # in the timed region several contexts launch concurrently, which stretches every launch
    solo = None
    solo_lat = None
    kernel_times = None
    if not args.no_check:
        # three such passes, every kernel's least time: now and then something stretches a pass's kernel several-fold (a search launch of
        # 13.8 ms beside 3.0-3.1 in the other passes of the same run, gpurun_out/r07h), and one pass alone would report that
        for _ in range(3):
            torch.cuda.synchronize()
            lib.search_profile(las[0].L, las[0].ctx_handle(), 3)  # | 2: an event pair around every ingest and cell kernel as well
            las[0].reset()
            las[0].run(device_ptrs=wl.seg_ptrs[0], stride=W, paced=args.paced)
            kt_ = lib.kernel_profile(las[0].L, las[0].ctx_handle())
            cms_, cnl_, ncell_ = lib.cell_profile(las[0].L, las[0].ctx_handle())
            sl_ = lib.search_profile_latency(las[0].L, las[0].ctx_handle())
            ms_, nl_, ns_ = lib.search_profile(las[0].L, las[0].ctx_handle(), 0)
            kernel_times = kt_ if kernel_times is None else [a if (a[0] <= b[0] or not b[1]) and a[1] else b for a, b in zip(kernel_times, kt_)]
            if solo_lat is None or (sl_[1] and sl_[0] < solo_lat[0]):
                solo_lat = sl_
            if ms_ > 0 and ns_ and (solo is None or ms_ < solo[0]):
                solo = (ms_, nl_, ns_)
    other_paced = not args.paced
    other_fps = None
    if not args.no_check:
        dt_other, outs_other = wl.timed(1, 0, paced=other_paced)
        for sgi in range(S):
            a, b = outputs_signature(outs_all[sgi], nb), outputs_signature(outs_other[sgi], nb)
            assert a == b, "segment %d: %s and %s passes disagree (first difference at output %d)" % (
                sgi, "paced" if args.paced else "batched", "paced" if other_paced else "batched", next(i for i, (x, y) in enumerate(zip(a, b)) if x != y))
        other_fps = round(S * F / dt_other, 2)
    wl.close()

    window = None
    res = None
    if rank == 0:
        bytes_per_search = algorithmic_bytes_per_search(cfg)
        achieved = (prof_searches * bytes_per_search / 1e9) / (prof_ms / 1e3) if prof_ms > 0 else 0.0
        traffic = tsrc = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                per_search = tj.get("me_rows_kernel_hbm_bytes_per_search")
                # PMC passes are separate runs of this command (profiles/*_pmc_summary.json), reduced to bytes per search; one launch of
                # THIS run carried searches/launches of them
                traffic = round(per_search * prof_searches / max(prof_launches, 1)) if per_search and W == 1920 else None
                tsrc = "%s: FETCH_SIZE + WRITE_SIZE passes of this command under rocprofv3 (not this run), per search x this run's searches per launch" % tj.get("source")
            except Exception:
                traffic = None
        res = {
            "metric": "lookahead frames/sec",
            "value": round(world * S * F * args.steps / dt, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "rccl_ranks": world if (world == 1 or backend == "nccl") else 0,  # ranks of the RCCL process group the summaries were gathered over
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8" if args.bit_depth == 8 else "u16",
            "data": "synthetic",
            "config": {"workload": "%dx%d %d-bit 4:2:0 synthetic, --preset %s --me %s%s: full lookahead "
                                   "(lowres+AQ+intra+ME+cost cells+slicetype decision+MB-tree), %d GOP segment(s) of %d frames in flight per GPU "
                                   "per step, %s" %
                                   (W, H, args.bit_depth, args.preset, args.me,
                                    " (BASELINE configs[1])" if (W, H, args.bit_depth, args.preset, args.me, args.threads) == (1920, 1080, 8, "slow", "dia", 1)
                                    else " --threads %d (%d lookahead bands)" % (args.threads, cfg["lookahead_threads"]) if args.threads > 1 else "",
                                    S, F, "encoder-paced" if args.paced else "deep-prefetch batch"),
                       "frames_per_step": S * F, "segments_in_flight": S, "bframes": cfg["bframes"], "b_adapt": cfg["b_adapt"], "rc_lookahead": cfg["rc_lookahead"],
                       "parallelism": "gop-segments x%d" % world, "slice_types": types[:64]},
            "checked": None if args.no_check else "types + every cost cell of the timed passes == one untimed %s pass of the same segments" % ("encoder-paced" if other_paced else "batched"),
            ("paced_fps" if other_paced else "batched_fps"): other_fps,
            # The kernel's own duration is what the launches of ONE context take with nothing else in flight (`solo`; the counter passes in
            # profiles/ measure the same thing).  In the timed region S contexts launch concurrently: their launches overlap, each is
            # stretched, and the sum of the launch durations exceeds the step time -- kept below as `concurrent`, it is a figure of the
            # contention, not of the kernel.
            "roofline": dict(
                {"bound": "hbm", "kernel": "me_rows_kernel", "peak": HBM_PEAK_GBS, "unit": "GB/s"},
                **({"achieved": round(solo[2] * bytes_per_search / 1e9 / (solo[0] / 1e3), 2),
                    "frac": round(solo[2] * bytes_per_search / 1e9 / (solo[0] / 1e3) / HBM_PEAK_GBS, 5),
                    "what": "one untimed pass of one segment alone on the GPU (no concurrent launches; the fastest of three such passes), HIP events on the library's stream",
                    "launches": solo[1], "searches": solo[2], "avg_launch_ms": round(solo[0] / solo[1], 4), "us_per_search": round(solo[0] * 1e3 / solo[2], 3),
                    "traffic": None if traffic is None else round(traffic * (solo[2] / max(solo[1], 1)) / max(prof_searches / max(prof_launches, 1), 1))}
                   if solo is not None else
                   {"achieved": round(achieved, 2), "frac": round(achieved / HBM_PEAK_GBS, 5), "what": "launches of the timed region (no solo pass: --no-check)",
                    "launches": prof_launches, "searches": prof_searches, "avg_launch_ms": round(prof_ms / max(prof_launches, 1), 4),
                    "us_per_search": round(prof_ms * 1e3 / max(prof_searches, 1), 3), "traffic": traffic}),
                **({"small_launches": {"what": "the search launches of the same pass that cannot fill the chip (fewer waves than wave slots: as long as their dependency chain): "
                                                 "the weighted searches of the fade, one batch per submission (x264hip_prefetch_weighted_fields)",
                                                "launches": solo_lat[1], "searches": solo_lat[2], "ms": round(solo_lat[0], 4)}} if solo_lat and solo_lat[1] else {}),
                **{"traffic_source": tsrc,
                   "concurrent": {"what": "the launches of the timed region: %d contexts launching at once, every launch stretched by the others (sum of launch "
                                          "durations / step time = %.2f)" % (S, prof_ms / max(dt * 1e3, 1e-9)),
                                  "launches": prof_launches, "searches": prof_searches, "avg_launch_ms": round(prof_ms / max(prof_launches, 1), 4),
                                  "us_per_search": round(prof_ms * 1e3 / max(prof_searches, 1), 3), "achieved": round(achieved, 2),
                                  "frac": round(achieved / HBM_PEAK_GBS, 5)}}),
            "roofline_extra_": {
                         "algorithmic_bytes_per_search": bytes_per_search,
                         "note": "not a streaming kernel: with the chip full it is bound by vector instruction issue and the L1 tag rate together (roofline_issue, "
                                 "roofline_l1; instructions and line accesses per block search in profiles/search_issue.json); HBM is the roofline the metric names.  "
                                 "A launch that does not fill the chip is bound by its dependency chain and runs on the latency form of the search out of LDS "
                                 "(me_latency.h; DESIGN.md section 3.1)"},
            "lookahead_stats": {"frame_cost_calls": int(la_stats[0]), "evaluations": int(la_stats[1]),
                                "weights_analysed": int(la_stats[2]), "weights_kept": int(la_stats[3]),
                                "device": {"searches": int(dev_counters[0]), "cell_requests": int(dev_counters[1]), "cell_hits": int(dev_counters[4]),
                                           "cells_on_demand": int(dev_counters[7]), "cells_speculated": int(dev_counters[5]),
                                           "fields_claimed": int(dev_counters[2]), "searches_on_demand": int(dev_counters[13]),
                                           "second_variants_speculated": int(dev_counters[14]), "second_variants_used": int(dev_counters[15]), "weight_sums_from_cache": int(dev_counters[6]),
                                           "unclaimed_field_share": round(1.0 - float(dev_counters[2]) / max(float(dev_counters[0]) - float(dev_counters[13]), 1.0), 4),
                                           "unused_cell_share": round(1.0 - float(dev_counters[4]) / max(float(dev_counters[5]), 1.0), 4),
                                           "note": "counters since the contexts were opened (warm-up, timed steps)"},
                                "weighted_speculation": {"searches_enqueued": int(wspec[0]), "fields_taken_over": int(wspec[1]), "cells_evaluated": int(wspec[2]), "b_cells_used": int(wspec[3]),
                                                         "what": "x264hip_prefetch_weighted_fields: the searches (and P cells) of the pairs whose weight the analysis keeps, ahead of the requests"},
                                "host_ms": {"frame_cost": round(la_stats[4] / 1e6, 2), "weights_analyse": round(la_stats[5] / 1e6, 2),
                                            "prefetch_mbtree": round(la_stats[6] / 1e6, 2), "api_total": round(la_stats[7] / 1e6, 2)}},
        }
        res["roofline"].update(res.pop("roofline_extra_"))
        if kernel_times is not None:
            res["roofline_kernels"] = kernel_rooflines(kernel_times, cfg, W, H, args.bit_depth, solo)
        ipath = os.path.join(ROOT, "profiles", "search_issue.json")
        if os.path.exists(ipath):
            try:
                pm, ps = (solo[0], solo[2]) if solo is not None else (prof_ms, prof_searches)
                res["roofline_issue"] = issue_roofline(json.load(open(ipath)), pm, ps, cfg, wall_s=dt, all_searches=prof_searches)
                res["roofline_issue"]["what"] = ("one untimed pass of one segment alone: " if solo is not None else "") + res["roofline_issue"]["what"]
            except Exception as e:  # pragma: no cover
                res["roofline_issue"] = {"error": str(e)}
        if os.path.exists(ipath):
            try:
                res["roofline_l1"] = l1_roofline(json.load(open(ipath)), solo if solo is not None else (prof_ms, prof_launches, prof_searches), cfg)
            except Exception as e:  # pragma: no cover
                res["roofline_l1"] = {"error": str(e)}
    # ---- N > 1: ONE stream over the ranks (BASELINE configs[3]) beside the GOP-segment figure.  The segment line above is complete at this
    # point; the window shard runs a collective protocol over a communicator of its own, and a rank that dies or a transport that wedges
    # would leave the others inside a collective.  A watchdog therefore bounds it: when it expires rank 0 prints the line it has (with
    # the reason) and every rank leaves -- the driver gets its line whatever happens to the extra measurement.
    if ( world > 1 and not args.no_extra ) or args.shard == "window":
        import threading
        finished = threading.Event()
        limit = float(os.environ.get("X264HIP_BENCH_WINDOW_TIMEOUT", "300"))

        def bail():
            if finished.is_set():
                return
            if rank == 0:
                res["window_shard_error"] = "the window-shard measurement did not finish within %.0f s (watchdog); the line is the GOP-segment measurement" % limit
                print(json.dumps(res), flush=True)
            os._exit(0 if args.shard != "window" else 4)
        timer = threading.Timer(limit, bail)
        timer.daemon = True
        timer.start()
        try:
            window = window_shard_bench(torch, lib, shard, dist, rank, world, dev_index, backend)
        except Exception as e:  # pragma: no cover
            if args.shard == "window":
                raise
            window = {"error": repr(e)} if rank == 0 else None
        finished.set()
        timer.cancel()

    if rank == 0:
        if window is not None:
            res["window_shard"] = window
            if "error" in window and "value" not in window:
                res["window_shard_error"] = window["error"]
            if args.shard == "window" and "value" in window:
                # the one-stream figure as the primary value (strong scaling); the segment figure stays in the line
                res["segments_value"] = res["value"]
                res.update(value=window["value"], scaling="strong", ms_per_step=round(window["seconds"] * 1e3, 3))
                res["config"] = {"workload": window["workload"], "parallelism": "window x%d" % world}
        if world == 1 and not args.no_extra and still is not None:
            # ---- the clip of rounds 1-5 (the same recipe with the fade under the full pan: the weight analysis never keeps a weight there), for
            # comparisons with their figures: what the kept weights of the default clip cost is the difference
            try:
                dev5 = [make_clip_device(torch, W, H, F, 100 + sgi, args.bit_depth, scene_cuts=(F // 3, F // 3 + 47), fade=(2 * F // 3, 10, 0.6, 12)) for sgi in range(S)]
                wl5 = Workload(torch, lib, shard, cfg, dev_index, 0, S, F, dev5, False)
                try:
                    dt5, o5 = wl5.timed(args.steps, args.warmup)
                    st5 = sum((la.stats() for la in wl5.las), np.zeros(8, np.uint64))
                finally:
                    wl5.close()
                res["round5_clip"] = {"what": "the same workload on the clip of rounds 1-5 (fade under a 5 x 3 samples per frame pan: no weight is ever kept, no weighted search runs)",
                                      "value": round(S * F * args.steps / dt5, 2), "unit": "frames/s", "ms_per_step": round(dt5 / args.steps * 1e3, 3),
                                      "weights_analysed": int(st5[2]), "weights_kept": int(st5[3])}
                del dev5
            except Exception as e:  # pragma: no cover
                res["round5_clip"] = {"error": repr(e)}
        if world == 1 and not args.no_extra:
            # ---- what a caller with HOST pictures gets (x264_encoder_encode is handed host buffers, encoder/encoder.c:3368-3454,
            # common/frame.c:445-447): the same segments, the clips in pinned host memory, through the same put / get calls.  The pictures
            # cross PCIe inside the timed region (W x H samples each), on the contexts' DMA streams; `value` above is NOT this figure.
            try:
                frame_bytes = W * H * (1 if args.bit_depth == 8 else 2)
                # the link's own rate: a plain pinned-to-device copy loop, best of three buffers (where a pinned buffer lands among the host's
                # memory nodes moves the figure by a third: 38.8 against 53.8 GB/s on two boxes of the same kind)
                pcie_peak = 0.0
                dp = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
                for _ in range(3):
                    hp = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
                    dp.copy_(hp, non_blocking=True); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(8):
                        dp.copy_(hp, non_blocking=True)
                    torch.cuda.synchronize()
                    pcie_peak = max(pcie_peak, 8 * hp.numel() / (time.perf_counter() - t0) / 1e9)
                    del hp
                del dp
                host_clips = [dv.cpu().pin_memory() for dv in seg_dev]
                def host_fed_run(Sh, steps_h):
                    """Sh of the segments in flight at a time, every picture from pinned host memory; checked against the device-resident passes"""
                    wlh = Workload(torch, lib, shard, cfg, dev_index, 0, Sh, F, host_clips[:Sh], False)
                    try:
                        dth, oh = wlh.timed(steps_h, 1)
                        for sgi in range(Sh):
                            assert outputs_signature(oh[sgi], nb) == outputs_signature(outs_all[sgi], nb), "host-fed segment %d differs from the device-resident pass" % sgi
                        hstat = np.zeros(3, np.uint64)
                        hstat2 = np.zeros(2, np.uint64)
                        for la in wlh.las:
                            b = np.zeros(3, np.uint64)
                            lib._ck(la.L.x264hip_host_transfer_stats(la.ctx_handle(), b.ctypes.data_as(ctypes.c_void_p)), "host_transfer_stats")
                            hstat += b
                            b2 = np.zeros(2, np.uint64)
                            lib._ck(la.L.x264hip_host_transfer_stats2(la.ctx_handle(), b2.ctypes.data_as(ctypes.c_void_p)), "host_transfer_stats2")
                            hstat2 += b2
                    finally:
                        wlh.close()
                    return Sh * F * steps_h / dth, hstat, hstat2
                hf_errors = []

                def host_fed_try(Sh, steps_h):
                    """one more attempt after a failed one (the contexts of an attempt are its own; gpurun_out/r07i: an in-kernel wait timed
                    out once in some sixty such runs, not reproduced since -- reported in the line, not hidden)"""
                    for attempt in range(2):
                        try:
                            return host_fed_run(Sh, steps_h)
                        except lib.X264HipError as e:
                            hf_errors.append("%d segments in flight, attempt %d: %r" % (Sh, attempt, e))
                            if attempt:
                                raise
                steps_h = max(2, args.steps // 2)
                fps_h, hstat, hstat2 = host_fed_try(S, steps_h)
                in_flight_h, by_inflight = S, {str(S): round(fps_h, 2)}
                if S >= 8:
                    # The link takes the segments one after the other (one transfer queue per device): fewer contexts wait less for it and
                    # keep it as busy -- four host-fed segments in flight run at 0.9 of the link, eight at 0.8 (gpurun_out/r07e)
                    f2, h2, h22 = host_fed_try(S // 2, 2 * steps_h)
                    by_inflight[str(S // 2)] = round(f2, 2)
                    if f2 > fps_h:
                        fps_h, hstat, hstat2, in_flight_h = f2, h2, h22, S // 2
                # one stream, encoder-paced, from host buffers: pinned, and plain pageable memory (staged through the library's pinned ring)
                one = {}
                for key_h, clip_h in (("device_resident", seg_dev[0]), ("pinned", host_clips[0]), ("pageable", seg_dev[0].cpu())):
                    w1 = Workload(torch, lib, shard, cfg, dev_index, 0, 1, F, [clip_h], True)
                    try:
                        dt1, o1 = w1.timed(2, 1, paced=True)
                        assert outputs_signature(o1[0], nb) == outputs_signature(outs_all[0], nb), "paced host-fed stream differs"
                        one[key_h] = round(F * 2 / dt1, 2)
                    finally:
                        w1.close()
                pcie_peak = max(pcie_peak, fps_h * frame_bytes / 1e9)  # (the pictures' own rate is a lower bound of what the link can do)
                bound = min(res["value"], pcie_peak * 1e9 / frame_bytes)
                res["host_fed"] = {"what": "the headline's segments with every picture in pinned HOST memory: %d x %d frames per step through x264hip_lookahead_put_frames "
                                           "(host pointers), H2D on the contexts' DMA streams inside the timed region" % (S, F),
                                   "fps": round(fps_h, 2), "segments_in_flight": in_flight_h, "fps_by_segments_in_flight": by_inflight, "pcie_GBps": round(fps_h * frame_bytes / 1e9, 2), "pcie_peak_GBps": round(pcie_peak, 2),
                                   "pcie_peak_what": "a plain pinned-to-device copy loop (8 x 256 MiB, best of three buffers) on this box, or the pictures' own rate if that is higher",
                                   "pcie_bound_fps": round(pcie_peak * 1e9 / frame_bytes, 1), "share_of_min_value_pcie_bound": round(fps_h / bound, 3),
                                   "pictures_direct_from_pinned": int(hstat[1]), "pictures_staged": int(hstat[2]), "transfers_of_a_whole_group": int(hstat2[0]),
                                   "single_stream_paced_fps": one, "checked": "types + every cost cell == the device-resident passes",
                                   **({"failed_attempts": hf_errors} if hf_errors else {})}
                del host_clips
            except Exception as e:  # pragma: no cover
                res["host_fed"] = {"error": repr(e)}
        if world == 1 and not args.no_extra and (W, H) == (1920, 1080):
            # the other single-GPU forms of the BASELINE configurations, clips generated on the device, each with the same check as
            # the headline (the batched passes that were timed == one encoder-paced pass: frame types and every cost cell)
            def other_config(key, what, cfg_x, Fx, Sx, steps_x, depth_x=8, cuts=None, fixture=None):
                """fixture = (name of a tests/golden/fullsize_*.npz, its clip as upscaled_clip() arguments): segment 0 runs the clip the
                real reference decided in the build container, and its coded order, slice types and every cost cell must equal the fixture's"""
                try:
                    devx = [make_clip_device(torch, cfg_x["width"], cfg_x["height"], Fx, 300 + sgi, depth_x, scene_cuts=cuts if cuts is not None else (Fx // 3,)) for sgi in range(Sx)]
                    fix = None
                    if fixture is not None:
                        import zlib
                        from x264_amd.synth import upscaled_clip
                        fix = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_%s.npz" % fixture[0]))
                        clip0 = upscaled_clip(cfg_x["width"], cfg_x["height"], Fx, depth_x, **fixture[1])
                        assert all(zlib.crc32(clip0[k].tobytes()) == int(fix["frame_crc"][k]) for k in (0, Fx // 2, Fx - 1)), "the fixture's clip did not regenerate"
                        devx[0] = torch.from_numpy(clip0.view(np.int16) if depth_x > 8 else clip0).cuda(dev_index)
                        del clip0
                    wlx = Workload(torch, lib, shard, cfg_x, dev_index, 0, Sx, Fx, devx, False)
                    try:
                        dtx, ox = wlx.timed(steps_x, 1)
                        dtp, opx = wlx.timed(2, 1, paced=True)  # one untimed pass first: a form of the search kernel this context has not launched yet loads its code
                        dtp /= 2
                        nbx = cfg_x["bframes"] + 2
                        for sgi in range(Sx):
                            assert outputs_signature(ox[sgi], nbx) == outputs_signature(opx[sgi], nbx), "%s: batched and paced passes disagree" % key
                        if fix is not None:
                            o0 = ox[0]
                            assert [o.frame for o in o0] == [int(v) for v in fix["idx"]] and [o.type for o in o0] == [int(v) for v in fix["type"]], "%s: decisions differ from the reference's fixture" % key
                            for k, o in enumerate(o0):
                                assert all(o.cost_est[i][j] == int(fix["cost"][k][i][j]) for i in range(nbx) for j in range(nbx)), "%s: cost cells of output %d differ from the reference's fixture" % (key, k)
                    finally:
                        wlx.close()
                    del devx
                    res[key] = {"workload": what + ", %d segment(s) of %d frames in flight, clips generated on the device" % (Sx, Fx) +
                                            (" (segment 0: the clip of tests/golden/fullsize_%s.npz)" % fixture[0] if fixture else ""),
                                "value": round(Sx * Fx * steps_x / dtx, 2), "unit": "frames/s", "paced_fps": round(Sx * Fx / dtp, 2),
                                "checked": "batched == paced (types + cost cells)" + ("; segment 0 == the real reference's decisions and cost cells (fixture)" if fixture else "")}
                except Exception as e:  # pragma: no cover
                    res[key] = {"error": str(e)}
            from tests.golden.make_golden import FULL_SIZE_CASES
            # BASELINE configs[2]: 3840x2160, --preset slower --me umh --merange 32 (the lookahead searches with HEX, range 32, b-adapt 2, rc-lookahead 60);
            # 96 frames per segment: the 60-frame window fills and slides under the trellis
            other_config("configs2_4k", "3840x2160 8-bit, --preset slower --me umh --merange 32 (BASELINE configs[2])",
                         lib.la_config(3840, 2160, "slower", bit_depth=8, me="umh", me_range=32), 96, S, 3)
            # BASELINE configs[4] on one GPU: 7680x4320 10-bit, --preset veryslow --me tesa (HEX with SATD full-pel costs, bframes 8, b-adapt 2,
            # rc-lookahead 60); 72 frames per segment: the window fills and slides (288 GB of HBM hold two such segments with room to spare)
            c4 = FULL_SIZE_CASES["configs4_8k_10bit"]
            other_config("configs4_8k_1gpu", "7680x4320 10-bit, --preset veryslow --me tesa (BASELINE configs[4], one GPU; the 60-frame window fills and slides)",
                         lib.la_config(7680, 4320, "veryslow", bit_depth=10, me="tesa"), c4[7], 2, 2, depth_x=10, cuts=(47,), fixture=("configs4_8k_10bit", c4[6]))
            # ONE stream alone on the GPU (one context, one host thread: what a single encoder instance sees), batched and
            # encoder-paced, for configs[1] and configs[2]; same check as above
            res["single_stream"] = {}
            for key_s, what_s, cfg_s, F_s in (("configs1", "1920x1080 8-bit, --preset slow --me dia (BASELINE configs[1])", cfg, F),
                                              ("configs2", "3840x2160 8-bit, --preset slower --me umh --merange 32 (BASELINE configs[2])",
                                               lib.la_config(3840, 2160, "slower", bit_depth=8, me="umh", me_range=32), 64)):
                other_config("_single", what_s, cfg_s, F_s, 1, 4)
                one = res.pop("_single")
                res["single_stream"][key_s] = one if "error" in one else {"workload": one["workload"], "batched_fps": one["value"], "paced_fps": one["paced_fps"],
                                                                          "unit": "frames/s", "checked": one["checked"]}
            # BASELINE configs[3] on one GPU: one 250-frame 4K GOP, --bframes 8 --rc-lookahead 60, ONE stream (the N = 1 point of the window shard)
            try:
                w1 = window_shard_bench(torch, lib, shard, None, 0, 1, dev_index, backend, steps=2, paced_check=True)
                res["configs3_4k_1gpu"] = w1
            except Exception as e:  # pragma: no cover
                res["configs3_4k_1gpu"] = {"error": str(e)}
        if not args.no_primitives:
            try:
                res["primitives"] = primitives_bench(torch, lib, dict(cfg, width=3840, height=2160))
            except Exception as e:  # pragma: no cover
                res["primitives"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline and frames is not None:
            n = min(F, args.cpu_frames)
            opts = "me=%s" % args.me
            if args.me_range:
                opts += ",merange=%d" % args.me_range
            if args.threads > 1:  # same bands on the CPU side (the harness drives the lookahead synchronously)
                opts += ",threads=%d,sync-lookahead=0,lookahead-threads=%d" % (args.threads, cfg["lookahead_threads"])
            res["cpu_baseline"] = cpu_baseline(cfg, frames[:n], args.preset, opts)
            if not args.no_extra and args.threads == 1:
                # the same clip with the reference's own lookahead threading on all host cores (bands of a frame in parallel; at most
                # X264_LOOKAHEAD_THREAD_MAX = 16, and at most mb_h / 4 ... of them)
                nt = max(2, min(16, os.cpu_count() or 2))
                mt = cpu_baseline(cfg, frames[:n], args.preset, opts, lookahead_threads=nt)
                if mt:
                    res["cpu_baseline_threads"] = mt
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if window is not None and "error" in window and "value" not in window and "MISMATCH" in str(window.get("error")):
        raise SystemExit(3)  # a sharded run whose results differ from the single-rank run is a failure, not a figure


def window_shard_bench(torch, lib, shard, dist, rank, world, dev_index, backend, steps=2, frames=250, paced_check=False):
    """BASELINE configs[3]: 3840x2160, one 250-frame GOP, --rc-lookahead 60 --bframes 8, ONE stream over all ranks: rank b % N runs frame
    b's searches and cost cells, the list-0 fields of list-1 references are exchanged, cell summaries are gathered to rank 0 (RCCL over
    xGMI), which decides, runs MB-tree and fetches the per-block maps MB-tree reads.  The one input copy lives on rank 0 and is
    broadcast inside the timed region.  Strong scaling: the same 250 frames at every N.  Returns the result object on rank 0 (None
    elsewhere).  The N > 1 result is checked against a single-rank pass of rank 0: a mismatch is an error (no value is reported)."""
    W, H = 3840, 2160
    cfg = lib.la_config(W, H, "medium", bit_depth=8, bframes=8, rc_lookahead=60, keyint_max=250)
    clip = make_clip_device(torch, W, H, frames, 4242, 8, scene_cuts=(frames // 3,)) if rank == 0 else None  # the other ranks receive the pictures inside every pass
    nb = cfg["bframes"] + 2
    on_dev = backend == "nccl"
    best = None
    outs = None
    tr = destroy = None
    if world > 1:
        # N > 1 goes through the library's own C entry points (x264hip_shard_*: what a C host calls): one RCCL
        # communicator per rank made by the library from an id rank 0 hands round; torch.distributed only carries that id and the timings
        L = lib.load()
        cfg["_frames"] = frames
        if on_dev:
            idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(shard.rccl_unique_id(L)), dtype=torch.uint8))
            dist.broadcast(idt, src=0)
            tr = shard.rccl_transport(L, bytes(idt.cpu().numpy().tobytes()), rank, world, dev_index)
            destroy, tr.destroy = tr.destroy, type(tr.destroy)()  # the communicator outlives the shards of the passes below
        else:
            tr = shard.HostStagedTransport(dist, rank, world)  # test rig: two ranks sharing one GPU over gloo
    for k in range(steps + 1):  # one warm-up pass
        if world > 1:
            outs, dt, stats, rc = shard.run_c_window_shard(torch, lib, rank, world, dev_index, cfg, clip, tr)
            if rc:
                raise SystemExit("window shard: rank %d left x264hip_shard_serve with %d" % (rank, rc))
        else:
            outs, dt, stats = shard.run_window_shard(torch, lib, None, rank, world, dev_index, cfg, clip, on_dev)
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if k > 0:
            best = float(t.item()) if best is None else min(best, float(t.item()))
        dt_last = float(t.item())
    res = None
    ok = torch.ones(1, dtype=torch.int32, device="cuda")
    if rank == 0:
        res = {"workload": "3840x2160 8-bit, one %d-frame GOP, --rc-lookahead 60 --bframes 8 (BASELINE configs[3]); ONE stream: frame b's searches and cost "
                           "cells on rank b %% N, cell summaries gathered to rank 0, which decides" % frames,
               "unit": "frames/s", "n_gpus": world, "rccl_ranks": world if on_dev else 0,
               "exchange": "x264hip_shard_* (C entry points of libx264hip.so), RCCL on the contexts' streams" if on_dev and world > 1 else "RCCL on the contexts' streams" if on_dev else backend,
               "scaling": "strong", "seconds": round(best, 4)}
        if world > 1:
            t1 = None
            ref, t1, st1 = shard.run_window_shard(torch, lib, None, 0, 1, dev_index, cfg, clip, on_dev)
            same = outputs_signature(outs, nb) == outputs_signature(ref, nb)
            if not same:
                ok[0] = 0
                res["error"] = "MISMATCH: decisions or cost cells differ from the single-rank run of the same stream"
            else:
                res["checked"] = "types + every cost cell == single-rank run of the same stream"
                res["value"] = round(frames / best, 2)
            moved = {k: stats.get(k, 0) for k in ("bytes_input_broadcast", "bytes_l0_exchange", "bytes_summaries", "bytes_maps")}
            res["bytes_moved"] = dict(moved, total_without_input=moved["bytes_l0_exchange"] + moved["bytes_summaries"] + moved["bytes_maps"],
                                      all_fields_to_rank0_would_be=8 * ((W + 15) // 16) * ((H + 15) // 16) * (st1["fields_searched"]))
            # rank 0's residual: what it still evaluates itself, against the single-rank run's totals (searches and cells are the two
            # kernels that scale; decisions and MB-tree stay on rank 0)
            own = dict(searches=stats.get("searches_here", 0) + stats.get("remote_fields_searched_here", 0), cells=stats.get("cells_here", 0) + stats.get("cells_on_demand", 0))
            one = dict(searches=st1.get("searches_here", 0), cells=st1.get("cells_here", 0) + st1.get("cells_on_demand", 0))
            frac = (own["searches"] + own["cells"]) / max(one["searches"] + one["cells"], 1)
            res["rank0_share"] = {"searches": own["searches"], "cells": own["cells"], "of_single_rank": one, "fraction_of_search_and_cell_evaluations": round(frac, 4),
                                  "maps_fetched": stats.get("maps_fetched", 0), "maps_recomputed_on_rank0": stats.get("remote_maps_recomputed_here", 0),
                                  "fetch_commands": stats.get("fetch_commands", 0), "chunks": stats.get("chunks", 0)}
            res["single_rank_seconds"] = round(t1, 4)
            res["speedup_vs_single_rank_pass"] = round(t1 / best, 3)
            res["amdahl"] = {"serial_fraction_measured": round(max(0.0, (best - t1 / world) / (t1 * (1 - 1.0 / world))), 4) if world > 1 else None,
                             "what": "the fraction s of the single-rank pass that did not scale, from T(N) = T(1) (s + (1 - s) / N) with both passes measured here"}
        else:
            res["value"] = round(frames / best, 2)
            res["cells_and_searches"] = {"searches": stats.get("searches_here", 0), "cells": stats.get("cells_here", 0), "cells_on_demand": stats.get("cells_on_demand", 0)}
            # What a window shard can reach from here: the searches and the cost cells are what moves to the owner ranks (frame b on rank
            # b % N); everything else of this pass -- ingest, intra, decisions, MB-tree, waiting for the last batch -- stays on rank 0.
            # Both parts measured in ONE more, untimed pass: HIP events around every search and cell launch on the context's stream (the
            # events perturb a pass, so `value` comes from the passes above, without them), wall time around that pass.  The prediction
            # leaves the exchange out (summaries of 8 + 2 mb_h ints per cell, list-0 fields of the list-1 references); the measured N > 1
            # line carries its own serial_fraction_measured.
            _, dt_last, stats = shard.run_window_shard(torch, lib, None, rank, world, dev_index, cfg, clip, on_dev, profile=True)
            par = (stats.get("device_ms_searches", 0.0) + stats.get("device_ms_cells", 0.0)) / 1e3
            if 0 < par < dt_last:
                ser = dt_last - par
                res["amdahl"] = {"pass_seconds": round(dt_last, 4), "shardable_seconds": round(par, 4), "serial_seconds": round(ser, 4),
                                 "device_ms": {"searches": stats.get("device_ms_searches"), "cells": stats.get("device_ms_cells"),
                                               "search_launches": stats.get("search_launches"), "cell_launches": stats.get("cell_launches")},
                                 "predicted_speedup": {str(n): round(dt_last / (ser + par / n), 2) for n in (2, 4, 8)},
                                 "what": "T(N) = serial + shardable / N with both terms measured in one more N = 1 pass after the timed ones (searches + cost cells by HIP "
                                         "events on the context's stream; serial = wall time of that pass minus that); exchange not included"}
            if paced_check:
                # the same stream through the encoder-paced put / get interface: same types, same cost cells
                la = lib.Lookahead(cfg, device=dev_index, max_frames=0)
                try:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    paced = la.run(device_ptrs=[clip[i].data_ptr() for i in range(frames)], stride=W, paced=True)
                    dtp = time.perf_counter() - t0
                finally:
                    la.close()
                if outputs_signature(outs, nb) != outputs_signature(paced, nb):
                    res.pop("value")
                    res["error"] = "MISMATCH: the batched and the encoder-paced pass of the stream disagree"
                else:
                    res["paced_fps"] = round(frames / dtp, 2)
                    res["checked"] = "batched == paced (types + cost cells)"
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # every rank learns the verdict (rank 0's verification pass ends here)
    if destroy is not None and tr is not None:
        destroy(tr.user)
    del clip
    if int(ok.item()) == 0 and rank == 0:
        print("window shard: " + res["error"], file=sys.stderr)
    return res


def kernel_rooflines(kt, cfg, W, H, depth, solo):
    """The kernels of the path beside the search, each against HBM: algorithmic bytes per unit (SURVEY 8d's formulas: what the kernel has
    to read and write once, whatever it re-reads) x units / device time between HIP events around its launches, one segment alone on the
    GPU (x264hip_kernel_profile).  `share_of_device_time`: of the summed device time of the pass's kernels on the main stream."""
    px = 1 if depth == 8 else 2
    mb_w, mb_h = (W + 15) // 16, (H + 15) // 16
    B = mb_w * mb_h
    S = 8 * mb_w * 8 * mb_h * px               # one unpadded lowres plane
    full = 16 * mb_w * 16 * mb_h * px          # the picture at its mod-16 size (frame.c:640-666)
    per_unit = {
        "lowres_tiles_kernel": (full + 4 * S, "per frame: the picture read once + four half-pel planes written (SURVEY 8d); the kernel also writes their strip copy, 8 S more"),
        "aq_kernel": (W * H * px + B * 16, "per frame: the picture read once + AQ factor, two offset maps and the sums of every macroblock written"),
        "intra_kernel": (S + 2 * B, "per frame: plane 0 read once + one cost per 8x8 block"),
        "cell_p_kernel": (10 * B, "per cell: mv cost, intra cost and AQ factor of every block read, lowres_cost written"),
        "cell_b_kernel": (3 * S + 22 * B, "per cell: source and one plane's worth of each reference read once, two vector granules + costs per block, lowres_cost written"),
        "cell_reduce_kernel": (6 * B, "per cell: the per-block costs summed"),
    }
    names = list(per_unit)
    total = sum(kt[k][0] for k in range(len(names))) + (solo[0] if solo else 0.0)
    out = {}
    for k, name in enumerate(names):
        ms, launches, units = kt[k]
        if not launches or ms <= 0:
            continue
        bytes_, what = per_unit[name]
        ach = units * bytes_ / 1e9 / (ms / 1e3)
        out[name] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "ms": round(ms, 4),
                     "launches": int(launches), "units": int(units), "algorithmic_bytes_per_unit": int(bytes_), "what": what,
                     "share_of_device_time": round(ms / total, 4) if total > 0 else None}
    if solo:
        out["me_rows_kernel"] = {"ms": round(solo[0], 4), "share_of_device_time": round(solo[0] / total, 4) if total > 0 else None, "see": "roofline"}
    return out


def l1_roofline(prof, timing, cfg):
    """Wave-wide vector loads of the search kernel against the MEASURED rate of a CU's L1 address unit for exactly this load shape
    (experiments/mem_rates, profiles/r05_mem_rates.json: 8 bytes per lane at the candidate's byte offset, lane = row of the strip, sixteen
    waves per CU reading an L1-resident region: 29.9 CU-cycles per instruction; any dword-aligned shape costs 17.3).  Loads per block
    search: SQ_INSTS_VMEM_RD of the counter passes (profiles/search_issue.json).  timing: (ms, launches, searches) of launches that ran alone.
    (Rounds 2-4 priced TCP_TOTAL_CACHE_ACCESSES against an assumed lookup per clock; both say about 0.6.  What the measurement added: the
    unit is NOT what bounds the kernel -- reading every tap as aligned dwords halves this figure and changes nothing, profiles/r05_search_ab.txt.)"""
    ms, launches, searches = timing
    blocks = ((cfg["width"] + 15) // 16) * ((cfg["height"] + 15) // 16)
    rates = json.load(open(os.path.join(ROOT, "profiles", "r05_mem_rates.json")))
    shape = rates["l1"][0]
    cycles, clock, cus = shape["cu_cycles"], rates["clock_GHz"] * 1e9, rates["cus"]
    loads = prof["vmem_rd_per_block"]
    peak = cus * clock / cycles
    achieved = searches * blocks * loads / (ms / 1e3) if ms > 0 else 0.0
    return {"bound": "l1-address-unit", "kernel": "me_rows_kernel", "achieved": round(achieved / 1e9, 3), "peak": round(peak / 1e9, 3), "unit": "G wave-loads/s",
            "frac": round(achieved / peak, 4), "wave_loads_per_block": loads, "cu_cycles_per_wave_load_measured": cycles, "load_shape": shape["pattern"],
            "searches_per_launch": round(searches / max(launches, 1)), "source": "profiles/r05_mem_rates.json (experiments/mem_rates) + " + str(prof.get("source")),
            "what": "vector loads of the launches that ran alone / their time, against %d CUs x the measured instruction rate of this load shape" % cus}


def issue_roofline(prof, prof_ms, prof_searches, cfg, wall_s=None, all_searches=None):
    """Second roofline of the search kernel, against the ceiling that binds it while the CU is full: vector-ALU issue cycles.
    profiles/search_issue.json holds wave-level VALU instructions per block search (SQ_INSTS_VALU pass of this command) and the
    cycles one of them occupies its SIMD on average (2 for the plain VOP2 kinds, 4 for the rest, measured with
    experiments/gen_valu_rate.py; the kernel's mix from scripts/valu_mix.py); 1024 SIMDs at 2.4 GHz."""
    blocks = ((cfg["width"] + 15) // 16) * ((cfg["height"] + 15) // 16)
    valu = prof["valu_per_block"]
    cpv = prof.get("cycles_per_valu", 4.0)
    peak = 1024 * 2.4e9 / cpv
    achieved = prof_searches * blocks * valu / (prof_ms / 1e3) if prof_ms > 0 else 0.0
    out = {"bound": "valu-issue", "kernel": "me_rows_kernel", "achieved": round(achieved / 1e9, 2), "peak": round(peak / 1e9, 1), "unit": "G wave-instr/s",
           "frac": round(achieved / peak, 4), "valu_per_block": valu, "cycles_per_valu_instruction": cpv, "source": prof.get("source"),
           "what": "achieved = VALU instructions of the launches / summed launch time (launches of several contexts overlap, each is stretched)",
           "note": "the ceiling that binds: %.0f %% of the issue cycles of the resident waves in the counter passes (four waves per SIMD taking turns; a lone "
                   "wave spends 2/3 of a step issuing); a launch that cannot fill the chip is as long as its dependency chain and goes to the latency "
                   "form of the search (DESIGN.md section 3.1, profiles/r04f_search_pmc.json, r04lat_search_pmc.json)"
                   % (100 * prof.get("valu_issue_utilisation_while_resident", 0))}
    if wall_s and all_searches:
        # whole timed region: every search of every context against the wall clock
        agg = all_searches * blocks * valu / wall_s
        out["aggregate"] = {"achieved": round(agg / 1e9, 2), "frac": round(agg / peak, 4),
                            "what": "VALU instructions of all searches of the timed steps / wall time of the timed steps (the GPU also runs every other kernel of the path)"}
    return out


if __name__ == "__main__":
    main()
