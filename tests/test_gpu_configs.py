"""GPU end-to-end for the configurations beyond the BASELINE ones (all verified on hardware since round 1): edge ring
not evaluated (no MB-tree: --no-mbtree, --qp, superfast / ultrafast, qcomp 1), lookahead bands (lookahead_threads > 1),
auto-variance AQ (aq-mode 2 / 3), constant QP, VBV lookahead, the batched main-encode block metrics.  Same golden fixtures and
checks as tests/test_gpu_lookahead.py."""
import os

import numpy as np
import pytest

from tests.golden.make_golden import LOOKAHEAD_CASES_R2
from tests.test_golden import GOLD, check_lookahead_outputs
from x264_amd import lib
from x264_amd.synth import make_chroma, make_clip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(LOOKAHEAD_CASES_R2))
@pytest.mark.parametrize("paced", [True, False])
def test_lookahead_vs_golden_new_configs(name, paced):
    preset, opts, over, depth, W, H, ckw, nf = LOOKAHEAD_CASES_R2[name]
    z = np.load(os.path.join(GOLD, "lookahead_%s.npz" % name))
    frames = make_clip(W, H, nf, bit_depth=depth, **ckw)
    over = dict(over)
    chroma = make_chroma(W, H, nf, seed=ckw.get("seed", 1), bit_depth=depth) if over.pop("_chroma", 0) else None
    cfg = lib.la_config(W, H, preset, bit_depth=depth, **over)
    la = lib.Lookahead(cfg, max_frames=0 if paced else nf + 4)
    try:
        outs = la.run(frames, paced=paced, qp_offsets=True, vbv=bool(cfg["vbv"]), chroma=chroma)
    finally:
        la.close()
    check_lookahead_outputs(outs, z, cfg["bframes"] + 2, check_qp=bool(cfg["aq_mode"]))


def test_aq_modes_against_oracle():
    """x264hip_frame_put with aq-mode 2 / 3: the Q8 inverse qscale map bit for bit against the oracle (sequential FP32 sums,
    correctly rounded roots and divisions, ratecontrol.c:354-398 in the reference build's operation order)."""
    from oracle.oraclelib import Oracle
    for depth in (8, 10):
        o = Oracle(depth)
        W, H = 352, 288
        fr = make_clip(W, H, 2, seed=7, bit_depth=depth, noise=20, texture=0.8)
        for mode, strength in ((2, 1.0), (3, 0.7), (2, 2.5)):
            ctx = lib.Context(W, H, bit_depth=depth, aq_mode=mode, aq_strength=strength, max_frames=8)
            try:
                ctx.frame_put(0, fr[0])
                inv = ctx.inv_qscale(0)
                qp = ctx.qp_offsets(0)
            finally:
                ctx.close()
            want, want_qp = o.aq_frame(fr[0], (W + 15) // 16, (H + 15) // 16, mode, strength)[:2]
            assert np.array_equal(qp, want_qp), (depth, mode, strength, float(np.abs(qp - want_qp).max()))
            assert np.array_equal(inv, want), (depth, mode, strength, int(np.abs(inv.astype(int) - want.astype(int)).max()))


# ---- evaluation level: the same replay as tests/test_gpu_parity.py::_run_sequence, for contexts opened with no_edges /
# lookahead_slices, comparing only what slicetype_slice_cost visits (slicetype.c:823-833) ----------------------------------
def _replay(o, cfg, ctx, frames, visited):
    from tests.test_gpu_parity import SEQ
    nf = len(frames)
    planes, inv, intra = [], [], []
    for i in range(nf):
        ctx.frame_put(i, frames[i])
        pl = o.lowres_init(cfg, frames[i])
        iq = o.aq_frame(frames[i], cfg.mb_w, cfg.mb_h, 1, 1.0)[0]
        ic = o.intra_costs(cfg, pl)   # 0xFFFF where never visited
        assert np.array_equal(ctx.intra_costs(i)[visited], ic[visited]), ("intra", i)
        planes.append(pl); inv.append(iq); intra.append(ic)
    fields, intra_done = {}, set()
    for (p0, p1, b) in SEQ:
        if max(p0, p1, b) >= nf:
            continue
        d0, d1 = b - p0, p1 - b
        with_intra = b not in intra_done
        if p0 == p1:
            out = ctx.frame_cost(p0, p1, b, 0, 0, (0, 0), None, with_intra, False)
            lc, rows, rows_i, oo = o.cell(cfg, planes[b], None, None, 128, None, None, None, None, None, intra[b].copy(), inv[b], with_intra)
            if with_intra:
                assert (out.intra_cost_est, out.intra_cost_est_aq) == (oo.intra_cost_est, oo.intra_cost_est_aq), ("intra sums", b)
                assert np.array_equal(ctx.lowres_costs(b, 0, 0)[1], rows_i), ("intra rows", b)
            intra_done.add(b)
            continue
        do0 = (b, 0, d0 - 1) not in fields
        do1 = d1 > 0 and (b, 1, d1 - 1) not in fields
        if do0:
            fields[(b, 0, d0 - 1)] = o.search_field(cfg, planes[b], planes[p0])
        if do1:
            fields[(b, 1, d1 - 1)] = o.search_field(cfg, planes[b], planes[p1])
        ref1_valid = d1 > 0 and (p1, 0, d0 + d1 - 1) in fields
        out = ctx.frame_cost(p0, p1, b, d0, d1, (do0, do1), None, with_intra, ref1_valid)
        m0, c0 = fields[(b, 0, d0 - 1)]
        gm, gc = ctx.mvs(b, 0, d0 - 1)
        assert np.array_equal(gm, m0), ("L0 mvs (zero where never searched)", p0, p1, b, int((gm != m0).any(1).sum()))
        assert np.array_equal(gc[visited], c0[visited]), ("L0 costs", p0, p1, b)
        dsf = (d0 * 256 + (d0 + d1) // 2) // (d0 + d1)
        if d1 > 0:
            m1, c1 = fields[(b, 1, d1 - 1)]
            gm1, gc1 = ctx.mvs(b, 1, d1 - 1)
            assert np.array_equal(gm1, m1) and np.array_equal(gc1[visited], c1[visited]), ("L1", p0, p1, b)
            r1 = fields[(p1, 0, d0 + d1 - 1)][0] if ref1_valid else None
            lc, rows, rows_i, oo = o.cell(cfg, planes[b], planes[p0], planes[p1], dsf, m0, c0, m1, c1, r1, intra[b], inv[b], with_intra)
        else:
            lc, rows, rows_i, oo = o.cell(cfg, planes[b], planes[p0], None, dsf, m0, c0, None, None, None, intra[b], inv[b], with_intra)
            intra_done.add(b)
        glc, grows = ctx.lowres_costs(b, d0, d1)
        assert np.array_equal(glc[visited], lc[visited]), ("lowres_costs", p0, p1, b, int((glc[visited] != lc[visited]).sum()))
        assert np.array_equal(grows, rows), ("row_satds", p0, p1, b)
        assert (out.cost_est, out.cost_est_aq) == (oo.cost_est, oo.cost_est_aq), ("sums", p0, p1, b)
        if d1 == 0:
            assert out.intra_mbs == oo.intra_mbs
        if with_intra:
            assert (out.intra_cost_est, out.intra_cost_est_aq) == (oo.intra_cost_est, oo.intra_cost_est_aq)


@pytest.mark.parametrize("no_edges,slices", [(1, 1), (0, 2), (0, 4), (1, 3)])
@pytest.mark.parametrize("cfgname", ["hex_r4", "dia_r2_sad", "hex_10bit"])
@pytest.mark.parametrize("clipname", ["fastpan", "noise"])
def test_eval_sequence_ring_and_bands(cfgname, clipname, no_edges, slices):
    from oracle.oraclelib import Oracle
    from tests.common import clip
    from tests.test_gpu_parity import CONFIGS
    depth, me_method, subpel_refine, me_range, subme, mbcmp_satd, fpelcmp_satd, bframes = CONFIGS[cfgname]
    W, H, nf = 352, 288, 4
    frames = clip(clipname, W, H, nf, depth)
    o = Oracle(depth)
    mb_w, mb_h = (W + 15) // 16, (H + 15) // 16
    cfg = o.make_cfg(mb_w, mb_h, me_method=me_method, subpel_refine=subpel_refine, me_range=me_range, mv_range=128, subme=subme,
                     mbcmp_satd=mbcmp_satd, fpelcmp_satd=fpelcmp_satd, n_slices=slices, do_edges=not no_edges)
    ctx = lib.Context(W, H, bit_depth=depth, bframes=bframes, me_method=me_method, subpel_refine=subpel_refine, me_range=me_range,
                      mv_range=128, subme=subme, mbcmp_satd=mbcmp_satd, fpelcmp_satd=fpelcmp_satd, max_frames=8, cost_mv=o._cost_mv,
                      no_edges=no_edges, lookahead_slices=slices)
    visited = np.ones((mb_h, mb_w), bool)
    if no_edges:
        visited[0, :] = visited[-1, :] = visited[:, 0] = visited[:, -1] = False
    try:
        _replay(o, cfg, ctx, frames, visited.reshape(-1))
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("name", ["ssd", "sa8d", "var", "hadamard_ac", "vsad", "asd8"])
def test_pixel_metric_batch(name, depth):
    """x264hip_pixel_metric_batch (ssd / sa8d / var / hadamard_ac / vsad / asd8 over a raster of blocks) against the oracle,
    block by block; the same arithmetic is checked on CPU by tests/test_block_metrics_host.py."""
    import torch
    from oracle.oraclelib import Oracle
    from tests.test_block_metrics_host import METRICS, oracle_metric
    o = Oracle(depth)
    W, H, stride = 160, 96, 192
    rng = np.random.default_rng(11 + depth)
    maxv = (1 << depth) - 1
    a = rng.integers(0, maxv + 1, size=(H, stride)).astype(o.dtype)
    b = np.clip(a.astype(np.int64) + rng.integers(-40, 41, size=(H, stride)), 0, maxv).astype(o.dtype)
    b[:16, :64] = maxv - a[:16, :64]  # large differences too
    tdt = torch.uint8 if depth == 8 else torch.int16
    da = torch.from_numpy(a.view(np.uint8 if depth == 8 else np.int16)).cuda()
    db = torch.from_numpy(b.view(np.uint8 if depth == 8 else np.int16)).cuda()
    assert da.dtype == tdt
    mid, sizes, two = METRICS[name]
    size_idx = {(16, 16): 0, (16, 8): 1, (8, 16): 2, (8, 8): 3, (8, 4): 4, (4, 8): 5, (4, 4): 6}
    ctx = lib.Context(352, 288, bit_depth=depth, max_frames=4)
    try:
        for (w, h) in sizes:
            bw, bh = W // w, H // h
            out = torch.zeros(bw * bh, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            ctx.pixel_metric_batch(mid, size_idx[(w, h)], da.data_ptr(), db.data_ptr() if two else None, stride, bw, bh, out.data_ptr())
            ctx.synchronize()
            got = out.cpu().numpy().view(np.uint64)
            for y in range(bh):
                for x in range(bw):
                    pa, pb = a[y * h:, x * w:], b[y * h:, x * w:]
                    want = oracle_metric(o, name, w, h, pa, stride, pb, stride)
                    want = want & 0xFFFFFFFFFFFFFFFF if name in ("var", "hadamard_ac") else want & 0xFFFFFFFF
                    assert int(got[y * bw + x]) == want, (name, depth, w, h, x, y)
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_frame_dct_quant8x8(depth):
    """sub8x8_dct8 + quant_8x8 of every 8x8 block of a plane pair (x264hip_frame_dct_quant8x8) against the oracle's per-block
    functions; the block arithmetic is also checked on CPU (tests/test_block_metrics_host.py)."""
    import ctypes as C
    import torch
    from oracle.oraclelib import Oracle
    o = Oracle(depth)
    maxv = (1 << depth) - 1
    rng = np.random.default_rng(41 + depth)
    W, H = 1032, 72  # 129 x 9 blocks: a ragged last wave
    fs, ds = W + 12, W + 40
    fenc = rng.integers(0, maxv + 1, size=(H, fs)).astype(o.dtype)
    fdec = rng.integers(0, maxv + 1, size=(H, ds)).astype(o.dtype)
    fenc[:8, :8] = maxv; fdec[:8, :8] = 0
    fdec[8:16, :8] = fenc[8:16, :8]
    mf = rng.integers(300, 14000, size=64).astype(o.ucoef_dtype)
    bias = rng.integers(0, 30000, size=64).astype(o.ucoef_dtype)
    vdt = np.uint8 if depth == 8 else np.int16
    fd, dd = torch.from_numpy(fenc.view(vdt)).cuda(), torch.from_numpy(fdec.view(vdt)).cuda()
    bw, bh = W // 8, H // 8
    coefs = torch.full((bh, bw, 64), 77, dtype=torch.int16 if depth == 8 else torch.int32, device="cuda")
    nz = torch.full((bh, bw), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        ctx.frame_dct_quant8x8(fd.data_ptr(), fs, dd.data_ptr(), ds, W, H, mf, bias, coefs.data_ptr(), nz.data_ptr())
        ctx.synchronize()
    finally:
        ctx.close()
    got, gnz = coefs.cpu().numpy(), nz.cpu().numpy()
    p = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
    dct, quant = o.f("dct"), o.f("quant", C.c_int)
    fe16 = np.zeros((8, 16), o.dtype); fd32 = np.zeros((8, 32), o.dtype)
    for by in range(bh):
        for bx in range(0, bw, 3 if by else 1):
            fe16[:, :8] = fenc[8 * by:8 * by + 8, 8 * bx:8 * bx + 8]
            fd32[:, :8] = fdec[8 * by:8 * by + 8, 8 * bx:8 * bx + 8]
            c = np.zeros(64, o.coef_dtype)
            dct(3, p(c), p(fe16), p(fd32))
            rnz = quant(1, p(c), p(mf), p(bias), 0, 0)
            assert np.array_equal(got[by, bx], c), (by, bx)
            assert int(gnz[by, bx]) == rnz, (by, bx)


@pytest.mark.parametrize("depth", [8, 10])
def test_me_search_batch(depth):
    """x264hip_me_search_batch (x264_me_search_ref with DIA / HEX / UMH / ESA / TESA + refine_subpel, one thread per request) on the
    calls recorded from the reference (tests/golden/me_full_d{8,10}.npz); the same code passes them on CPU
    (tests/test_me_full_host.py)."""
    import torch
    from tests.common import ME_METHODS
    from tests.test_me_full_host import request_geometry
    z = np.load(os.path.join(GOLD, "me_full_d%d.npz" % depth))
    W, H, pw, ph, padh, padv, mv_range = (int(v) for v in z["geom"])
    vdt = np.uint8 if depth == 8 else np.int16
    isz = 1 if depth == 8 else 2
    planes = [torch.from_numpy(np.ascontiguousarray(z["planes"][p]).view(vdt)).cuda() for p in range(4)]
    frame = torch.from_numpy(np.ascontiguousarray(z["fenc_frame"], np.uint8 if depth == 8 else np.uint16).view(vdt)).cuda()
    integral = torch.from_numpy(np.ascontiguousarray(z["integral"]).view(np.int16)).cuda()
    cost_mv = np.ascontiguousarray(z["cost_mv"])
    centre = (cost_mv.size - 1) // 2
    cmv = torch.from_numpy(cost_mv.view(np.int16)).cuda()
    torch.cuda.synchronize()
    org0 = padv * pw + padh
    reqs, want, subs = [], [], []
    for me in ME_METHODS:
        for call in z["calls_%s" % me]:
            i_pixel, mb_x, mb_y, xoff, yoff, subme, me_range, mvpx, mvpy, n_mvc = (int(v) for v in call[:10])
            smin, smax, lim_min, lim_max, sx, sy, org = request_geometry(z["geom"], call)
            q = lib.MeRequest()
            q.i_pixel, q.me_method, q.subpel_refine, q.me_range = i_pixel, ME_METHODS[me], subme, me_range
            q.mbcmp_satd, q.fpelcmp_satd = 1, int(me == "tesa")
            q.x, q.y = sx, sy
            for k in range(2):
                q.mvp[k] = (mvpx, mvpy)[k]
                q.lim_min[k], q.lim_max[k], q.spel_min[k], q.spel_max[k] = lim_min[k], lim_max[k], smin[k], smax[k]
            q.n_mvc = n_mvc
            for i in range(4):
                q.mvc[i][0], q.mvc[i][1] = int(call[10 + 2 * i]), int(call[11 + 2 * i])
            reqs.append(q); want.append(call[18:22]); subs.append(subme)
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        got = ctx.me_search_batch(reqs, frame.data_ptr(), frame.shape[1], [p.data_ptr() + org0 * isz for p in planes], pw,
                                  integral.data_ptr() + org0 * 2, ph * pw, cmv.data_ptr() + 2 * centre)
        # the same requests resident on the device, one method per table (x264hip_me_search_batch_dev): nothing crosses the host link
        import ctypes as C
        got_dev = np.zeros_like(got)
        for me, mid in ME_METHODS.items():
            sel = [k for k, q in enumerate(reqs) if q.me_method == mid]
            arr = ctx.me_requests([reqs[k] for k in sel])
            raw = torch.from_numpy(np.frombuffer(bytes(C.string_at(C.addressof(arr), C.sizeof(arr))), dtype=np.uint8).copy()).cuda()
            res = torch.full((len(sel), 4), -7, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            ctx.me_search_batch_dev(len(sel), raw.data_ptr(), frame.data_ptr(), frame.shape[1], [p.data_ptr() + org0 * isz for p in planes], pw,
                                    integral.data_ptr() + org0 * 2, ph * pw, cmv.data_ptr() + 2 * centre, mid, max(reqs[k].me_range for k in sel), res.data_ptr())
            ctx.synchronize()
            got_dev[sel] = res.cpu().numpy()
    finally:
        ctx.close()
    for k in range(len(reqs)):
        n = 4 if subs[k] >= 2 else 3
        assert np.array_equal(got[k][:n], want[k][:n]), (k, got[k].tolist(), [int(v) for v in want[k]])
        assert np.array_equal(got_dev[k][:n], want[k][:n]), ("device-resident table", k, got_dev[k].tolist(), [int(v) for v in want[k]])
    assert len(reqs) >= 500


@pytest.mark.parametrize("depth", [8, 10])
def test_searches_wave_form_equals_scalar_form(depth):
    """Main-encode search requests run one per WAVE on the device: block costs across the lanes, the candidates of a pattern's set
    requested together, ESA 256 candidates per step (v_qsad_pk_u16_u8), TESA with ordered compaction of the ads survivors and the SAD
    stage's running thresholds as a prefix minimum across the lanes.  The same header compiled for the host runs a request one candidate
    after the other (tests/tools/block_metrics_host.cpp, pinned to the recorded reference calls by tests/test_me_full_host.py).  2 000
    random requests -- all five methods, every partition size, ranges 8 / 16 / 24, predictors all over the window including its edges --
    must give the same vector, cost and cost_mv both ways."""
    import ctypes as C
    import torch
    from tests.test_block_metrics_host import _lib
    from tests.test_me_full_host import MfHostReq, request_geometry
    from tests.common import ME_SIZES
    z = np.load(os.path.join(GOLD, "me_full_d%d.npz" % depth))
    W, H, pw, ph, padh, padv, mv_range = (int(v) for v in z["geom"])
    dt = np.uint8 if depth == 8 else np.uint16
    vdt = np.uint8 if depth == 8 else np.int16
    isz = 1 if depth == 8 else 2
    hplanes = [np.ascontiguousarray(z["planes"][p]) for p in range(4)]
    hframe = np.ascontiguousarray(z["fenc_frame"], dt)
    hint = np.ascontiguousarray(z["integral"])
    cost_mv = np.ascontiguousarray(z["cost_mv"])
    centre = (cost_mv.size - 1) // 2
    L = _lib()
    fn = L.mf_host_u8 if depth == 8 else L.mf_host_u16
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(900 + depth)
    reqs, want = [], []
    for t in range(2000):
        i_pixel = int(rng.integers(0, 7))
        bw, bh = ME_SIZES[i_pixel]
        mb_x, mb_y = int(rng.integers(0, W // 16)), int(rng.integers(0, H // 16))
        xoff, yoff = int(rng.integers(0, 16 // bw)) * bw, int(rng.integers(0, 16 // bh)) * bh
        me = t % 5
        subme = int(rng.choice([1, 2, 5, 7]))
        me_range = int(rng.choice([8, 16, 16, 24]))
        call = [i_pixel, mb_x, mb_y, xoff, yoff]
        smin, smax, lim_min, lim_max, sx, sy, org = request_geometry(z["geom"], call)
        mvp = [int(rng.integers(smin[k], smax[k] + 1)) for k in range(2)]
        n_mvc = int(rng.integers(0, 5))
        mvc = np.zeros((4, 2), np.int16)
        for i in range(n_mvc):
            mvc[i] = [int(rng.integers(smin[k] - 8, smax[k] + 9)) for k in range(2)]
        m = MfHostReq()
        m.i_pixel, m.me_method, m.subpel_refine, m.me_range = i_pixel, me, subme, me_range
        m.mbcmp_satd, m.fpelcmp_satd = 1, int(me == 4)
        m.fenc = hframe.ctypes.data + (sy * hframe.shape[1] + sx) * hframe.itemsize
        m.fenc_stride = hframe.shape[1]
        for p_ in range(4):
            m.ref[p_] = hplanes[p_].ctypes.data + org * hplanes[p_].itemsize
        m.stride = pw
        m.integral = hint.ctypes.data + org * 2
        m.integral_lower = ph * pw
        m.cost_mv = cost_mv.ctypes.data + 2 * centre
        q = lib.MeRequest()
        q.i_pixel, q.me_method, q.subpel_refine, q.me_range = i_pixel, me, subme, me_range
        q.mbcmp_satd, q.fpelcmp_satd = 1, int(me == 4)
        q.x, q.y = sx, sy
        for k in range(2):
            m.mvp[k] = q.mvp[k] = mvp[k]
            m.spel_min[k] = q.spel_min[k] = smin[k]; m.spel_max[k] = q.spel_max[k] = smax[k]
            m.lim_min[k] = q.lim_min[k] = lim_min[k]; m.lim_max[k] = q.lim_max[k] = lim_max[k]
        q.n_mvc = n_mvc
        for i in range(4):
            q.mvc[i][0], q.mvc[i][1] = int(mvc[i][0]), int(mvc[i][1])
        out = np.zeros(4, np.int32)
        fn(C.byref(m), mvc.ctypes.data, n_mvc, out.ctypes.data)
        reqs.append(q); want.append((out.copy(), subme))
    planes = [torch.from_numpy(hplanes[p_].view(vdt)).cuda() for p_ in range(4)]
    frame = torch.from_numpy(hframe.view(vdt)).cuda()
    integral = torch.from_numpy(hint.view(np.int16)).cuda()
    cmv = torch.from_numpy(cost_mv.view(np.int16)).cuda()
    torch.cuda.synchronize()
    org0 = padv * pw + padh
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        got = ctx.me_search_batch(reqs, frame.data_ptr(), frame.shape[1], [p_.data_ptr() + org0 * isz for p_ in planes], pw,
                                  integral.data_ptr() + org0 * 2, ph * pw, cmv.data_ptr() + 2 * centre)
    finally:
        ctx.close()
    for k, (w, subme) in enumerate(want):
        n = 4 if subme >= 2 else 3
        assert np.array_equal(got[k][:n], w[:n]), (k, reqs[k].me_method, reqs[k].i_pixel, reqs[k].me_range, got[k].tolist(), w.tolist())


@pytest.mark.parametrize("depth", [8, 10])
def test_searches_wave_form_equals_oracle(depth):
    """The wave form of the main-encode searches DIRECTLY against the oracle (oracle/x264_oracle.c me_search_full, pinned to the real
    reference by tests/test_me_full_vs_ref.py): 2 500 random requests per depth, half of them the exhaustive methods -- ESA through the
    v_qsad_pk_u16_u8 scan (8-bit) and TESA with its survivor lists -- every partition size, ranges 8 / 16 / 24, predictors anywhere in the
    window, up to four extra candidates.  (test_searches_wave_form_equals_scalar_form compares the wave form with the scalar form of the
    same header; this one has no leg in common with the code under test.)"""
    import torch
    from oracle.oraclelib import Oracle
    from tests.common import ME_SIZES, oracle_me_search
    from tests.test_me_full_host import request_geometry
    z = np.load(os.path.join(GOLD, "me_full_d%d.npz" % depth))
    W, H, pw, ph, padh, padv, mv_range = (int(v) for v in z["geom"])
    o = Oracle(depth)
    dt = np.uint8 if depth == 8 else np.uint16
    vdt = np.uint8 if depth == 8 else np.int16
    isz = 1 if depth == 8 else 2
    hplanes = [np.ascontiguousarray(z["planes"][p_], dt) for p_ in range(4)]
    hframe = np.ascontiguousarray(z["fenc_frame"], dt)
    hint = np.ascontiguousarray(z["integral"])
    cost_mv = np.ascontiguousarray(z["cost_mv"])
    centre = (cost_mv.size - 1) // 2
    names = ["dia", "hex", "umh", "esa", "tesa"]
    rng = np.random.default_rng(4100 + depth)
    reqs, want = [], []
    for t in range(2500):
        me = (3, 4, 3, 4, 0, 1, 2, 3, 4, 2)[t % 10]
        i_pixel = int(rng.integers(0, 7))
        bw, bh = ME_SIZES[i_pixel]
        mb_x, mb_y = int(rng.integers(0, W // 16)), int(rng.integers(0, H // 16))
        xoff, yoff = int(rng.integers(0, 16 // bw)) * bw, int(rng.integers(0, 16 // bh)) * bh
        subme = int(rng.choice([1, 2, 5, 7]))
        me_range = int(rng.choice([8, 16, 16, 24]))
        call = [i_pixel, mb_x, mb_y, xoff, yoff]
        smin, smax, lim_min, lim_max, sx, sy, org = request_geometry(z["geom"], call)
        mvp = [int(rng.integers(smin[k], smax[k] + 1)) for k in range(2)]
        n_mvc = int(rng.integers(0, 5))
        mvc = np.zeros((4, 2), np.int16)
        for i in range(n_mvc):
            mvc[i] = [int(rng.integers(smin[k] - 8, smax[k] + 9)) for k in range(2)]
        fenc = np.zeros((16, 16), dt)
        blk = hframe[sy:sy + bh, sx:sx + bw]
        fenc[:blk.shape[0], :blk.shape[1]] = blk
        full = call + [subme, me_range, mvp[0], mvp[1], n_mvc] + [int(v) for v in mvc.reshape(-1)]
        want.append((oracle_me_search(o, names[me], hplanes, hint, cost_mv, z["geom"], fenc, full), subme))
        q = lib.MeRequest()
        q.i_pixel, q.me_method, q.subpel_refine, q.me_range = i_pixel, me, subme, me_range
        q.mbcmp_satd, q.fpelcmp_satd = 1, int(me == 4)
        q.x, q.y = sx, sy
        for k in range(2):
            q.mvp[k] = mvp[k]
            q.spel_min[k], q.spel_max[k], q.lim_min[k], q.lim_max[k] = smin[k], smax[k], lim_min[k], lim_max[k]
        q.n_mvc = n_mvc
        for i in range(4):
            q.mvc[i][0], q.mvc[i][1] = int(mvc[i][0]), int(mvc[i][1])
        reqs.append(q)
    planes = [torch.from_numpy(hplanes[p_].view(vdt)).cuda() for p_ in range(4)]
    frame = torch.from_numpy(hframe.view(vdt)).cuda()
    integral = torch.from_numpy(hint.view(np.int16)).cuda()
    cmv = torch.from_numpy(cost_mv.view(np.int16)).cuda()
    torch.cuda.synchronize()
    org0 = padv * pw + padh
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        got = ctx.me_search_batch(reqs, frame.data_ptr(), frame.shape[1], [p_.data_ptr() + org0 * isz for p_ in planes], pw,
                                  integral.data_ptr() + org0 * 2, ph * pw, cmv.data_ptr() + 2 * centre)
    finally:
        ctx.close()
    for k, (w, subme) in enumerate(want):
        n = 4 if subme >= 2 else 3
        assert np.array_equal(got[k][:n], w[:n]), (k, names[reqs[k].me_method], reqs[k].i_pixel, reqs[k].me_range, got[k].tolist(), w.tolist())


@pytest.mark.parametrize("depth", [8, 10])
def test_integral_init(depth):
    """x264hip_integral_init against the integral planes the reference built (x264_frame_filter, recorded in the golden file) and
    against plain box sums; then the TESA requests of the recording run on the DEVICE-built planes."""
    import torch
    z = np.load(os.path.join(GOLD, "me_full_d%d.npz" % depth))
    plane = np.ascontiguousarray(z["planes"][0])
    ph, pw = plane.shape
    ref = np.ascontiguousarray(z["integral"])
    vdt = np.uint8 if depth == 8 else np.int16
    dp = torch.from_numpy(plane.view(vdt)).cuda()
    out = torch.zeros((2 * ph, pw), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        ctx.integral_init(dp.data_ptr(), pw, pw, ph, out.data_ptr(), out.data_ptr() + 2 * ph * pw)
    finally:
        ctx.close()
    got = out.cpu().numpy().view(np.uint16)
    assert np.array_equal(got[1:ph - 8, :pw - 8], ref[1:ph - 8, :pw - 8])
    assert np.array_equal(got[ph + 1:2 * ph - 8, :pw - 8], ref[ph + 1:2 * ph - 8, :pw - 8])
    p = plane.astype(np.int64)
    c = np.zeros((ph + 1, pw + 1), np.int64); c[1:, 1:] = p.cumsum(0).cumsum(1)
    for n, base in ((8, 0), (4, ph)):
        box = (c[n:, n:] - c[:-n, n:] - c[n:, :-n] + c[:-n, :-n]) & 0xFFFF
        assert np.array_equal(got[base:base + ph - n + 1, :pw - n + 1], box)


@pytest.mark.parametrize("depth", [8, 10])
def test_frame_filter(depth):
    """x264hip_frame_filter (border expansion + hpel planes + their borders + integral planes of a whole frame) against the four
    padded planes and the integral planes recorded from the reference (x264_frame_expand_border -> x264_frame_filter ->
    x264_frame_expand_border_filtered run row by row like the encoder does, oracle/ref_harness.c:rh_add_ref_frame)."""
    import torch
    z = np.load(os.path.join(GOLD, "me_full_d%d.npz" % depth))
    W, H, pw, ph, padh, padv, _ = (int(v) for v in z["geom"])
    gold = z["planes"]
    vdt = np.uint8 if depth == 8 else np.int16
    isz = 1 if depth == 8 else 2
    pic = np.ascontiguousarray(gold[0][padv:padv + H, padh:padh + W])
    dpic = torch.from_numpy(pic.view(vdt)).cuda()
    planes = torch.full((4, ph, pw), 0x55, dtype=torch.uint8 if depth == 8 else torch.int16, device="cuda")
    integ = torch.zeros((2 * ph, pw), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    org0 = (padv * pw + padh) * isz
    ctx = lib.Context(64, 64, bit_depth=depth, max_frames=2, mv_range=32)
    try:
        ctx.frame_filter(dpic.data_ptr(), W, W, H, [planes[k].data_ptr() + org0 for k in range(4)], pw, padh, padv,
                         integ.data_ptr(), integ.data_ptr() + 2 * ph * pw)
        ctx.synchronize()
    finally:
        ctx.close()
    got = planes.cpu().numpy().view(np.uint8 if depth == 8 else np.uint16)
    for k in range(4):
        assert np.array_equal(got[k], gold[k]), ("plane", k, int((got[k] != gold[k]).sum()))
    gi = integ.cpu().numpy().view(np.uint16)
    ref = z["integral"]
    assert np.array_equal(gi[1:ph - 8, :pw - 8], ref[1:ph - 8, :pw - 8])
    assert np.array_equal(gi[ph + 1:2 * ph - 8, :pw - 8], ref[ph + 1:2 * ph - 8, :pw - 8])


@pytest.mark.parametrize("depth", [8, 10])
def test_frame_put_with_chroma(depth):
    """x264hip_frame_put with the 4:2:0 chroma planes -- from host buffers (staged like the luma) and from device memory -- against
    the oracle's adaptive quantisation of the whole picture (pinned against x264_adaptive_quant_frame in
    tests/test_oracle_vs_ref.py::test_adaptive_quant_with_chroma): aq-mode 1, 2, 3, a non-mod16 size."""
    import ctypes as C
    import torch
    from oracle.oraclelib import Oracle
    o = Oracle(depth)
    vdt = np.uint8 if depth == 8 else np.int16
    maxv = (1 << depth) - 1
    for (W, H, fmt) in ((352, 288, 1), (100, 70, 1), (176, 144, 2), (100, 70, 3)):
        y = make_clip(W, H, 1, seed=5, bit_depth=depth, noise=20)[0]
        if fmt == 1:
            cb, cr = (np.ascontiguousarray(c[0]) for c in make_chroma(W, H, 1, seed=5, bit_depth=depth))
        else:   # 4:2:2: (W+1)//2 x H, 4:4:4: W x H
            rng = np.random.default_rng(W + fmt)
            shape = (H, W if fmt == 3 else (W + 1) // 2)
            cb = rng.integers(0, maxv + 1, size=shape).astype(o.dtype)
            cr = np.clip(rng.normal(maxv / 2, maxv / 6, size=shape), 0, maxv).astype(o.dtype)
        cw = cb.shape[1]
        for mode, strength in ((1, 1.0), (2, 1.0), (3, 0.6)):
            want_inv, want_qp = o.aq_frame(y, (W + 15) // 16, (H + 15) // 16, mode, strength, cb, cr, chroma_format=fmt)[:2]
            ctx = lib.Context(W, H, bit_depth=depth, aq_mode=mode, aq_strength=strength, max_frames=4, chroma_format=fmt)
            try:
                L = ctx.L
                L.x264hip_frame_put.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
                yy = np.ascontiguousarray(y, o.dtype)
                assert L.x264hip_frame_put(ctx.h, 0, yy.ctypes.data, W, 0, cb.ctypes.data, cr.ctypes.data, cw, None) == 0   # host buffers
                dy, dcb, dcr = (torch.from_numpy(a.view(vdt)).cuda() for a in (yy, cb, cr))
                torch.cuda.synchronize()
                assert L.x264hip_frame_put(ctx.h, 1, dy.data_ptr(), W, 1, dcb.data_ptr(), dcr.data_ptr(), cw, None) == 0    # device memory
                for slot in (0, 1):
                    assert np.array_equal(ctx.qp_offsets(slot), want_qp), (W, H, mode, slot)
                    assert np.array_equal(ctx.inv_qscale(slot), want_inv), (W, H, mode, slot)
            finally:
                ctx.close()


def test_put_pictures_device_batch():
    """x264hip_lookahead_put_pictures with device-resident 4:2:0 pictures (one ingest launch for the whole clip, chroma included)
    against the golden run of the same clip fed picture by picture from host buffers."""
    import torch
    name = "chroma_aq"
    preset, opts, over, depth, W, H, ckw, nf = LOOKAHEAD_CASES_R2[name]
    z = np.load(os.path.join(GOLD, "lookahead_%s.npz" % name))
    frames = make_clip(W, H, nf, bit_depth=depth, **ckw)
    cb, cr = make_chroma(W, H, nf, seed=ckw.get("seed", 1), bit_depth=depth)
    over = dict(over); over.pop("_chroma")
    cfg = lib.la_config(W, H, preset, bit_depth=depth, **over)
    dy, dcb, dcr = torch.from_numpy(frames).cuda(), torch.from_numpy(cb).cuda(), torch.from_numpy(cr).cuda()
    torch.cuda.synchronize()
    la = lib.Lookahead(cfg, max_frames=nf + 4)
    try:
        la.put_pictures([dy[i].data_ptr() for i in range(nf)], W, [dcb[i].data_ptr() for i in range(nf)], [dcr[i].data_ptr() for i in range(nf)],
                        cb.shape[2])
        outs = []
        while True:
            o = la.get(True, True)
            if o is None:
                break
            outs.append(o)
    finally:
        la.close()
    check_lookahead_outputs(outs, z, cfg["bframes"] + 2)


@pytest.mark.parametrize("depth", [8, 10])
def test_frame_add_quant_offsets(depth):
    """x264hip_frame_add_quant_offsets (x264_picture_t.prop.quant_offsets) against the oracle's AQ with the same offsets: aq-mode 1
    and 3, and AQ on with strength 0 (what MB-tree forces when AQ is off)."""
    import ctypes as C
    from oracle.oraclelib import Oracle
    o = Oracle(depth)
    W, H = 352, 288
    mb_w, mb_h = (W + 15) // 16, (H + 15) // 16
    y = make_clip(W, H, 1, seed=5, bit_depth=depth, noise=20)[0]
    offs = np.random.default_rng(9).normal(0, 3, size=mb_w * mb_h).astype(np.float32)
    for mode, strength in ((1, 1.0), (3, 0.7), (1, 0.0)):
        want_inv, want_qp = o.aq_frame(y, mb_w, mb_h, mode, strength, quant_offsets=offs)[:2]
        ctx = lib.Context(W, H, bit_depth=depth, aq_mode=mode, aq_strength=strength, max_frames=4)
        try:
            ctx.frame_put(0, y)
            ctx.L.x264hip_frame_add_quant_offsets.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
            assert ctx.L.x264hip_frame_add_quant_offsets(ctx.h, 0, offs.ctypes.data) == 0
            assert np.array_equal(ctx.qp_offsets(0), want_qp), (mode, strength)
            assert np.array_equal(ctx.inv_qscale(0), want_inv), (mode, strength)
        finally:
            ctx.close()
