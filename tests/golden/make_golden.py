"""Generates the committed golden fixtures from the REAL reference (oracle/_ref, built from
/root/reference by oracle/build_ref.sh).  Run in the build container:  python tests/golden/make_golden.py
Fixtures are data only: seeded inputs (or the recipe to regenerate them) and the reference's outputs."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refharness  # noqa: E402
from tests.common import clip, nearest_ref_cells  # noqa: E402
from x264_amd.synth import make_chroma, make_clip, upscaled_clip  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

LOOKAHEAD_CASES = {
    # name: (preset, ref opts, cfg overrides, depth, W, H, clip kwargs, n_frames)
    "medium_cif": ("medium", "", {}, 8, 352, 288, dict(seed=1, scene_cuts=(25,), fade=(40, 8, 0.6, 10)), 60),
    "slow_dia": ("slow", "me=dia", dict(me="dia"), 8, 176, 144, dict(seed=2, scene_cuts=(13, 14, 31)), 64),
    "slower_umh32": ("slower", "me=umh,merange=32", dict(me="umh", me_range=32), 8, 176, 144, dict(seed=3, pan=(9, 5), scene_cuts=(40,)), 72),
    "b8_la60": ("medium", "bframes=8,rc-lookahead=60", dict(bframes=8, rc_lookahead=60), 8, 176, 144, dict(seed=4, fade=(10, 12, 1.5, -20)), 75),
    "veryslow_tesa_10bit": ("veryslow", "me=tesa", dict(me="tesa"), 10, 176, 144, dict(seed=5, scene_cuts=(33,)), 70),
    "keyint24": ("medium", "keyint=24,min-keyint=4", dict(keyint_max=24, keyint_min=4), 8, 176, 144, dict(seed=6, scene_cuts=(7, 50)), 60),
    "veryfast": ("veryfast", "", {}, 8, 176, 144, dict(seed=7, pan=(1, 1), noise=1), 40),
    "nonmod16": ("medium", "", {}, 8, 200, 120, dict(seed=12, pan=(7, 3)), 48),
    "slow_dia_720p": ("slow", "me=dia", dict(me="dia"), 8, 1280, 720, dict(seed=13, scene_cuts=(20,)), 58),
    # weightp=0 + mbtree + psy = WEIGHTP_FAKE: weightdelta != 0 in macroblock_tree_finish (slicetype.c:462-463, 1032-1034)
    "fakeweight_fade": ("fast", "bframes=4,b-adapt=2,b-pyramid=strict,keyint=12,min-keyint=0,rc-lookahead=40,weightp=0",
                        dict(bframes=4, b_adapt=2, b_pyramid=1, keyint_max=12, keyint_min=0, rc_lookahead=40, weightp=0), 8, 176, 144,
                        dict(seed=178, scene_cuts=(28,), pan=(3, 0), fade=(15, 10, 1.5, -4)), 48),
    "fakeweight_opengop": ("fast", "bframes=1,b-adapt=1,keyint=24,min-keyint=0,rc-lookahead=10,weightp=0,open-gop=1",
                           dict(bframes=1, b_adapt=1, keyint_max=24, keyint_min=0, rc_lookahead=10, weightp=0, open_gop=1), 8, 96, 80,
                           dict(seed=840, scene_cuts=(24, 36), pan=(3, 1), fade=(2, 10, 1.5, 5)), 50),
}

# Configurations whose device paths were written after the last GPU session of round 1 (edge ring not evaluated, lookahead
# bands, auto-variance AQ, constant QP, the two fastest presets): same fixtures and checks, but their GPU tests live in
# tests/test_gpu_configs.py.
LOOKAHEAD_CASES_R2 = {
    "no_mbtree": ("medium", "mbtree=0,bframes=5,b-adapt=2,rc-lookahead=30", dict(mb_tree=0, bframes=5, b_adapt=2, rc_lookahead=30),
                  8, 176, 144, dict(seed=14, scene_cuts=(19,), fade=(30, 8, 0.6, 5)), 50),
    "superfast_cif": ("superfast", "", {}, 8, 352, 288, dict(seed=18, scene_cuts=(21,), pan=(6, 2)), 40),
    "ultrafast": ("ultrafast", "", {}, 8, 176, 144, dict(seed=19, scene_cuts=(12,)), 30),
    "cqp": ("slow", "bframes=1,b-adapt=0,keyint=8,scenecut=0,rc-lookahead=5,subme=1,qp=24",
            dict(bframes=1, b_adapt=0, keyint_max=8, scenecut=0, rc_lookahead=5, subme=1, rc_is_cqp=1), 8, 176, 144,
            dict(seed=526, scene_cuts=(21, 40), pan=(3, 3), fade=(9, 10, 0.6, 12)), 46),
    "aq2_nopsy": ("medium", "aq-mode=2,psy=0", dict(aq_mode=2, psy=0), 8, 176, 144, dict(seed=15, scene_cuts=(22,)), 40),
    "aq3_10bit": ("fast", "aq-mode=3,aq-strength=0.5,qcomp=0.4", dict(aq_mode=3, aq_strength=0.5, qcompress=0.4), 10, 176, 144,
                  dict(seed=24, scene_cuts=(30,), pan=(2, 3), fade=(10, 10, 0.6, 15)), 44),
    "bands3_cif": ("medium", "threads=6,sync-lookahead=0,lookahead-threads=3", dict(threads=6, lookahead_threads=3), 8, 352, 288,
                   dict(seed=16, pan=(23, 11), noise=30, texture=0.9, scene_cuts=(20,)), 40),
    "vbv_cif": ("medium", "bitrate=500,vbv-bufsize=300,vbv-maxrate=600", dict(bitrate=500, vbv_bufsize=300, vbv_maxrate=600), 8, 352, 288,
                dict(seed=21, scene_cuts=(23,), pan=(4, 2), fade=(30, 8, 0.6, 10)), 44),
    "vbv_no_mbtree": ("fast", "vbv-bufsize=200,vbv-maxrate=400,mbtree=0,b-adapt=2", dict(vbv_bufsize=200, vbv_maxrate=400, mb_tree=0, b_adapt=2),
                      8, 176, 144, dict(seed=22, scene_cuts=(15,)), 40),
    # whole 4:2:0 pictures: the chroma planes (x264_amd.synth.make_chroma, same seed as the luma clip) enter adaptive quantisation
    "chroma_aq": ("medium", "", dict(_chroma=1), 8, 352, 288, dict(seed=23, scene_cuts=(19,), pan=(4, 1)), 40),
    "bands_auto_720p": ("veryslow", "threads=24,sync-lookahead=0,lookahead-threads=auto", dict(threads=24), 8, 1280, 720, dict(seed=20, pan=(9, 4)), 20),
    # lookahead-less MB-tree (rc-lookahead 0 is only kept with infinite keyint or intra refresh, encoder.c:1128-1133): the propagation of
    # one call carries over to the next through an exchange of accumulators (X264HIP_MBT_SWAP / RESET_QP, slicetype.c:1112-1124,1173-1178)
    "la0_keyint_inf": ("medium", "keyint=infinite,rc-lookahead=0", dict(keyint_max=1 << 30, rc_lookahead=0), 8, 176, 144, dict(seed=31, scene_cuts=(21,), pan=(3, 1)), 40),
    "la0_intra_refresh": ("medium", "intra-refresh=1,rc-lookahead=0,keyint=30", dict(intra_refresh=1, rc_lookahead=0, keyint_max=30), 8, 176, 144,
                          dict(seed=32, scene_cuts=(17,), pan=(2, 2), fade=(25, 8, 0.7, 6)), 44),
}

# BASELINE configs[3] and configs[4] AS WRITTEN (full picture size, a filled 60-frame window, the whole 250-frame GOP): the clip is
# x264_amd.synth.upscaled_clip( W, H, n, depth, **kwargs ), regenerated from the same call on the GPU box; the fixture holds the
# reference's decisions, every cost cell and a CRC-32 of every frame's f_qp_offset / i_propagate_cost (the maps themselves are 32 400 and
# 129 600 entries per frame).  tests/test_gpu_lookahead.py::test_baseline_configs_as_written.
FULL_SIZE_CASES = {
    # name: (preset, ref opts, cfg overrides, depth, W, H, clip kwargs, n_frames)
    "configs3_4k_gop250": ("medium", "bframes=8,rc-lookahead=60", dict(bframes=8, rc_lookahead=60), 8, 3840, 2160,
                           dict(seed=61, pan=(1, 0), fade=(150, 60, 0.8, 8)), 250),
    "configs4_8k_10bit": ("veryslow", "me=tesa", dict(me="tesa"), 10, 7680, 4320, dict(seed=62, pan=(2, 1), scene_cuts=(47,)), 72),
    # BASELINE configs[2]: the 60-frame window fills and slides under the b-adapt-2 trellis, HEX with range 32 (UMH capped, slicetype.c:50-59)
    "configs2_4k_umh32": ("slower", "me=umh,merange=32", dict(me="umh", me_range=32), 8, 3840, 2160,
                          dict(seed=63, pan=(3, 2), scene_cuts=(31,), fade=(50, 10, 0.7, 8)), 72),
}


def gen_full_size(only=None):
    import time
    import zlib
    for name, (preset, opts, over, depth, W, H, ckw, nf) in FULL_SIZE_CASES.items():
        if only and name not in only:
            continue
        t0 = time.time()
        frames = upscaled_clip(W, H, nf, depth, **ckw)
        t1 = time.time()
        r = refharness.Ref(W, H, preset, opts=opts, bit_depth=depth)
        ref = r.lookahead_run(frames, with_qp_offsets=True)
        nb = r.cfg["bframes"] + 2
        qp_crc = np.array([zlib.crc32(np.ascontiguousarray(q).tobytes()) for q in ref["qp_offset"]], np.uint32)
        prop_crc = np.array([zlib.crc32(np.ascontiguousarray(q).tobytes()) for q in ref["propagate"]], np.uint32)
        np.savez_compressed(os.path.join(OUT, "fullsize_%s.npz" % name), idx=ref["idx"], type=ref["type"].astype(np.int8),
                            cost=ref["cost"][:, :nb, :nb], cost_aq=ref["cost_aq"][:, :nb, :nb], intra_mbs=ref["intra_mbs"][:, :nb],
                            qp_crc=qp_crc, prop_crc=prop_crc, frame_crc=np.array([zlib.crc32(f.tobytes()) for f in frames], np.uint32),
                            cfg=np.array([r.cfg[k] for k in sorted(r.cfg)], np.int64), cfg_keys=np.array(sorted(r.cfg)))
        r.close()
        print("fullsize", name, "clip %.0f s, reference %.0f s" % (t1 - t0, time.time() - t1), "types:", "".join("?IiPbB"[t] for t in ref["type"][:60]), flush=True)


EVAL_CONFIGS = [("medium", "", 8), ("slow", "me=dia", 8), ("medium", "subme=1", 8), ("veryslow", "me=tesa", 10)]
EVAL_SEQ = [(0, 0, 0), (0, 1, 1), (0, 2, 2), (0, 2, 1), (1, 1, 1), (0, 3, 3), (0, 3, 1), (0, 3, 2), (1, 3, 2), (2, 3, 3), (3, 3, 3)]


def gen_lookahead(only=None):
    for name, (preset, opts, over, depth, W, H, ckw, nf) in list(LOOKAHEAD_CASES.items()) + list(LOOKAHEAD_CASES_R2.items()):
        if only and name not in only:
            continue
        frames = make_clip(W, H, nf, bit_depth=depth, **ckw)
        r = refharness.Ref(W, H, preset, opts=opts, bit_depth=depth)
        vbv = bool(over.get("vbv_bufsize"))
        cells = None
        if vbv:  # first pass for the coded order, from which the nearest references of every B frame follow
            first = r.lookahead_run(frames)
            cells = np.array(nearest_ref_cells(first["idx"], first["type"]), np.int32)
            r.close()
            r = refharness.Ref(W, H, preset, opts=opts, bit_depth=depth)
        chroma = make_chroma(W, H, nf, seed=ckw.get("seed", 1), bit_depth=depth) if over.get("_chroma") else None
        ref = r.lookahead_run(frames, with_qp_offsets=True, with_vbv=vbv, rc_cells=cells, chroma=chroma)
        nb = r.cfg["bframes"] + 2
        extra = {}
        if vbv:  # what VBV rate control reads: planned types / costs, and per frame the result of the real x264_rc_analyse_slice
            extra = dict(planned_type=ref["planned_type"].astype(np.uint8), planned_satd=ref["planned_satd"], rc=ref["rc"], rc_cells=cells)
        if W * H <= 352 * 288:  # MB-tree outputs (f_qp_offset, i_propagate_cost) of every frame as it leaves the lookahead
            extra.update(qp_offset=ref["qp_offset"], propagate=ref["propagate"])
        np.savez_compressed(os.path.join(OUT, "lookahead_%s.npz" % name), idx=ref["idx"], type=ref["type"].astype(np.int8),
                            cost=ref["cost"][:, :nb, :nb], cost_aq=ref["cost_aq"][:, :nb, :nb],
                            cfg=np.array([r.cfg[k] for k in sorted(r.cfg)], np.int64), cfg_keys=np.array(sorted(r.cfg)), **extra)
        r.close()
        print("lookahead", name, "types:", "".join("?IiPbB"[t] for t in ref["type"][:40]))


def gen_evalseq():
    for preset, opts, depth in EVAL_CONFIGS:
        for clipname in ("fastpan", "noise", "fade"):
            W, H, nf = 176, 144, 4
            frames = clip(clipname, W, H, nf, depth)
            r = refharness.Ref(W, H, preset, opts=opts, bit_depth=depth)
            d = dict(cfg=np.array([r.cfg[k] for k in sorted(r.cfg)], np.int64), cfg_keys=np.array(sorted(r.cfg)))
            for i in range(nf):
                r.add_frame(frames[i])
                iq, _, ss = r.frame_stats(i)
                d["inv_%d" % i] = iq
                d["sums_%d" % i] = np.array(ss, np.uint64)
                d["lowres0_%d" % i] = r.lowres(i, 0)
                d["lowres_crc_%d" % i] = np.array([int(np.uint64(r.lowres(i, p).astype(np.uint64).sum())) for p in range(4)], np.uint64)
            for k, (p0, p1, b) in enumerate(EVAL_SEQ):
                score = r.frame_cost(p0, p1, b)
                lc, rows, summ = r.cell(b, b - p0, p1 - b)
                d["lc_%d" % k] = lc
                d["summ_%d" % k] = np.array(list(summ) + [score], np.int64)
                d["weight_%d" % k] = np.array(r.weight(b), np.int32)
                if b != p0:
                    mv, c = r.mvs(b, 0, b - p0 - 1)
                    d["mv0_%d" % k], d["c0_%d" % k] = mv, c
                if b != p1:
                    mv, c = r.mvs(b, 1, p1 - b - 1)
                    d["mv1_%d" % k], d["c1_%d" % k] = mv, c
            for i in range(nf):
                d["intra_%d" % i] = r.frame_stats(i)[1]
            tag = "%s_%s_%d_%s" % (preset, opts.replace("=", "").replace(",", "_") or "default", depth, clipname)
            np.savez_compressed(os.path.join(OUT, "evalseq_%s.npz" % tag), **d)
            r.close()
            print("evalseq", tag)


def gen_tables():
    for (W, H, depth) in ((176, 144, 8), (1280, 720, 8), (1920, 1080, 8), (176, 144, 10), (3840, 2160, 10)):
        r = refharness.Ref(W, H, "medium", bit_depth=depth)
        np.save(os.path.join(OUT, "cost_mv_r%d_d%d.npy" % (r.cfg["mv_range"], depth)), r.cost_mv())
        print("cost_mv", W, H, depth, r.cfg["mv_range"], r.cfg["lambda"])
        r.close()


def gen_primitives():
    """Known answers of the reference's C vtables on seeded inputs (tools/checkasm.c style)."""
    sizes = [(16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)]
    for depth in (8, 10):
        r = refharness.Ref(64, 64, "medium", bit_depth=depth)
        L = r.lib
        rng = np.random.default_rng(100 + depth)
        dt = np.uint8 if depth == 8 else np.uint16
        maxv = (1 << depth) - 1
        a = rng.integers(0, maxv + 1, size=(48, 64)).astype(dt)
        b = (rng.integers(0, 2, size=(48, 64)) * maxv).astype(dt)
        L.rh_pixel_cmp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long]
        res = np.zeros((4, 7, 3), np.int64)
        offs = [(0, 0), (3, 5), (17, 1)]
        for kind in range(3):
            for si in range(7):
                for oi, (ox, oy) in enumerate(offs):
                    res[kind, si, oi] = L.rh_pixel_cmp(r.ctx, kind, si, C.c_void_p(a.ctypes.data), 64,
                                                       C.c_void_p(b.ctypes.data + (oy * 64 + ox) * a.itemsize), 64)
        for si, s in ((0, 0), (3, 3)):
            for oi, (ox, oy) in enumerate(offs):
                res[3, s, oi] = L.rh_pixel_cmp(r.ctx, 3, si, C.c_void_p(a.ctypes.data), 64,
                                               C.c_void_p(b.ctypes.data + (oy * 64 + ox) * a.itemsize), 64)
        # dct/quant
        fenc = rng.integers(0, maxv + 1, size=(16, 16)).astype(dt)
        fdec = rng.integers(0, maxv + 1, size=(16, 32)).astype(dt)
        cdt = np.int16 if depth == 8 else np.int32
        udt = np.uint16 if depth == 8 else np.uint32
        dcts = {}
        for kind, n in {0: 16, 1: 64, 2: 256, 3: 64, 4: 256, 5: 4, 6: 8}.items():
            o = np.zeros(n, cdt)
            L.rh_dct(r.ctx, kind, C.c_void_p(o.ctypes.data), C.c_void_p(fenc.ctypes.data), C.c_void_p(fdec.ctypes.data))
            dcts["dct%d" % kind] = o
        mf4 = np.zeros(16, udt); b4 = np.zeros(16, udt); mf8 = np.zeros(64, udt); b8 = np.zeros(64, udt)
        L.rh_quant_tables(r.ctx, 0, 0, 26, C.c_void_p(mf4.ctypes.data), C.c_void_p(b4.ctypes.data))
        L.rh_quant_tables(r.ctx, 1, 0, 26, C.c_void_p(mf8.ctypes.data), C.c_void_p(b8.ctypes.data))
        q4 = dcts["dct0"].copy(); q8 = dcts["dct3"].copy()
        L.rh_quant.restype = C.c_int
        nz4 = L.rh_quant(r.ctx, 0, C.c_void_p(q4.ctypes.data), 0, 26, 0)
        nz8 = L.rh_quant(r.ctx, 1, C.c_void_p(q8.ctypes.data), 0, 26, 0)
        # hpel_filter (mc.c:172-196) of a 44x12 area inside a' (rows/columns with the margins the filter reads)
        hw, hh, hs = 44, 12, 64
        hsrc = rng.integers(0, maxv + 1, size=(hh + 8, hs)).astype(dt)
        hsrc[:, 20:26] = maxv; hsrc[2:5, 30:40] = 0
        hout = np.full((3, hh + 8, hs), 7, dt)
        hbuf = np.zeros(hw + 64, np.int16)
        hoff = (3 * hs + 8) * hsrc.itemsize
        L.rh_hpel_filter.argtypes = [C.c_void_p] * 5 + [C.c_long, C.c_int, C.c_int, C.c_void_p]
        L.rh_hpel_filter(r.ctx, hout[0].ctypes.data + hoff, hout[1].ctypes.data + hoff, hout[2].ctypes.data + hoff, hsrc.ctypes.data + hoff, hs, hw, hh,
                         hbuf.ctypes.data)
        np.savez_compressed(os.path.join(OUT, "primitives_d%d.npz" % depth), a=a, b=b, cmp=res, offs=np.array(offs),
                            fenc=fenc, fdec=fdec, mf4=mf4, bias4=b4, mf8=mf8, bias8=b8, q4=q4, q8=q8, nz=np.array([nz4, nz8]),
                            hpel_src=hsrc, hpel_out=hout, hpel_dims=np.array([hw, hh, hs]), **dcts)
        r.close()
        print("primitives", depth)


def gen_me_full():
    """x264_me_search_ref (main-encode form) on a small reference frame: inputs and results of 120 random calls per method."""
    from tests.common import ME_METHODS, ME_SIZES
    W, H = 96, 80
    for depth in (8, 10):
        fr = make_clip(W, H, 2, seed=77 + depth, bit_depth=depth, pan=(5, -3), noise=8, texture=0.6)
        dt = np.uint8 if depth == 8 else np.uint16
        store = {}
        for me, mid in ME_METHODS.items():
            r = refharness.Ref(W, H, "medium", opts="me=%s,partitions=all,merange=24" % me, bit_depth=depth)
            L = r.lib
            L.rh_add_ref_frame.argtypes = [C.c_void_p, C.c_void_p]
            L.rh_me_search.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
            L.rh_get_ref_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            L.rh_get_integral.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
            ref = np.ascontiguousarray(fr[0])
            L.rh_add_ref_frame(r.ctx, ref.ctypes.data)
            geo = (C.c_int * 8)()
            L.rh_ref_geometry(r.ctx, geo)
            w, lines, rstride, padh, padv, has_int, padh_align, sub8 = list(geo)
            pw, ph = w + 2 * padh, lines + 2 * padv
            if "planes" not in store:
                pl = np.zeros((4, ph, pw), dt)
                for p in range(4):
                    L.rh_get_ref_plane(r.ctx, 0, p, pl[p].ctypes.data)
                store["planes"] = pl
                store["cost_mv"] = r.cost_mv()
                store["geom"] = np.array([W, H, pw, ph, padh, padv, r.cfg["mv_range"]])
            if has_int and "integral" not in store:
                raw = np.zeros(2 * ph * rstride, np.uint16)
                L.rh_get_integral(r.ctx, 0, raw.ctypes.data, raw.size)
                x0 = padh_align - padh
                store["integral"] = np.ascontiguousarray(raw.reshape(2 * ph, rstride)[:, x0:x0 + pw])
            rng = np.random.default_rng(1000 + mid)
            rows = []
            for trial in range(120):
                i_pixel = int(rng.integers(0, 7))
                bw, bh = ME_SIZES[i_pixel]
                mb_x, mb_y = int(rng.integers(0, W // 16)), int(rng.integers(0, H // 16))
                xoff, yoff = int(rng.integers(0, 16 // bw)) * bw, int(rng.integers(0, 16 // bh)) * bh
                subme, me_range = int(rng.choice([1, 2, 3, 5, 7, 9])), int(rng.choice([8, 16, 24]))
                mvp = rng.integers(-60, 61, size=2).astype(np.int16) if trial % 3 else np.zeros(2, np.int16)
                n_mvc = int(rng.integers(0, 5))
                mvc = np.ascontiguousarray(rng.integers(-90, 91, size=(4, 2)).astype(np.int16))
                fenc = np.zeros((16, 16), dt)
                sy, sx = 16 * mb_y + yoff, 16 * mb_x + xoff
                fenc[:bh, :bw] = fr[1][sy:sy + bh, sx:sx + bw]
                out = np.zeros(4, np.int32)
                L.rh_me_search(r.ctx, 0, fenc.ctypes.data, mb_x, mb_y, xoff, yoff, i_pixel, subme, me_range, mvp.ctypes.data, mvc.ctypes.data,
                               n_mvc, out.ctypes.data)
                rows.append([i_pixel, mb_x, mb_y, xoff, yoff, subme, me_range, int(mvp[0]), int(mvp[1]), n_mvc] + mvc.reshape(-1).tolist() + out.tolist())
            store["calls_%s" % me] = np.array(rows, np.int32)
            r.close()
        store["fenc_frame"] = fr[1]
        np.savez_compressed(os.path.join(OUT, "me_full_d%d.npz" % depth), **store)
        print("me_full", depth)


if __name__ == "__main__":
    if "--me-full-only" in sys.argv:
        gen_me_full()
        sys.exit(0)
    if "--primitives-only" in sys.argv:
        gen_primitives()
        sys.exit(0)
    if "--full-size" in sys.argv:  # minutes of reference time; the other fixtures stay byte-identical
        i = sys.argv.index("--full-size")
        gen_full_size(sys.argv[i + 1].split(",") if len(sys.argv) > i + 1 else None)
        sys.exit(0)
    if "--lookahead-cases" in sys.argv:  # only the named cases (the other fixtures stay byte-identical)
        gen_lookahead(sys.argv[sys.argv.index("--lookahead-cases") + 1].split(","))
        sys.exit(0)
    if "--lookahead-only" not in sys.argv:
        gen_tables()
        gen_primitives()
        gen_evalseq()
        gen_me_full()
    gen_lookahead()
