"""Deterministic synthetic YUV 4:2:0 clips (luma only matters for the lookahead; chroma is mid-grey).

Recipe follows SURVEY.md section 8(d): a smooth random field (uniform noise, 4 passes of a 5-tap box
blur) is sampled through a window that pans by (3i mod 256, 2i mod 128) per frame, +-3 uniform noise is
added per frame, the luma is inverted at scene changes and an optional linear fade segment exercises
weighted prediction.
"""
import numpy as np


def _box5(a, axis):
    out = np.zeros_like(a)
    for k in range(-2, 3):
        out += np.roll(a, k, axis=axis)
    return out / 5.0


def pan_offsets(n_frames, pan, still=None):
    """Offset of the sampling window per frame: pan[0] / pan[1] samples further every frame; inside still = (start, length) the camera
    all but stops (one sample per frame horizontally) -- a fade over a fast pan is hidden from the lookahead's weight analysis, which
    compares frames without motion compensation (encoder/slicetype.c:191-222), and real fades mostly sit on quiet shots."""
    dx = dy = 0
    out = []
    for i in range(n_frames):
        out.append((dx % 256, dy % 128))
        slow = still is not None and still[0] <= i + 1 < still[0] + still[1]
        dx += 1 if slow else pan[0]
        dy += 0 if slow else pan[1]
    return out


def make_clip(width, height, n_frames, seed=1, bit_depth=8, scene_cuts=(), fade=None, noise=3, pan=(3, 2),
              texture=0.18, still=None):
    """Return uint8/uint16 array [n_frames, height, width] of luma samples.

    scene_cuts: frame indices at which the picture is inverted (a hard cut).
    pan: (dx, dy) full-resolution pixels of camera pan per frame (wraps inside the field).
    texture: share of white noise mixed into the smooth field.
    fade: (start, length, gain_end, offset_end) -- linear fade applied over [start, start+length) and
          held afterwards.
    still: (start, length) -- frames during which the pan slows to one sample per frame (see pan_offsets).
    """
    rng = np.random.default_rng(seed)
    fh, fw = height + 128 + 8, width + 256 + 8
    field = rng.uniform(0.0, 1.0, size=(fh, fw))
    for _ in range(4):
        field = _box5(_box5(field, 0), 1)
    field -= field.min()
    field /= max(field.max(), 1e-9)
    # add a little high-frequency texture so SATD/intra modes are non-trivial
    field = (1.0 - texture) * field + texture * rng.uniform(0.0, 1.0, size=field.shape)
    field = 16.0 + field * 219.0
    cuts = sorted(set(int(c) for c in scene_cuts))
    frames = np.empty((n_frames, height, width), dtype=np.uint8 if bit_depth == 8 else np.uint16)
    scale = 1 << (bit_depth - 8)
    maxv = (1 << bit_depth) - 1
    offs = pan_offsets(n_frames, pan, still)
    for i in range(n_frames):
        dx, dy = offs[i]
        img = field[dy:dy + height, dx:dx + width].copy()
        inverted = sum(1 for c in cuts if c <= i) & 1
        if inverted:
            img = 255.0 - img
        if fade is not None:
            s, ln, g1, o1 = fade
            t = min(max((i - s + 1) / float(ln), 0.0), 1.0)
            img = img * (1.0 + (g1 - 1.0) * t) + o1 * t
        img = img + rng.integers(-noise, noise + 1, size=img.shape)
        img = np.clip(np.rint(img * scale), 0, maxv)
        frames[i] = img.astype(frames.dtype)
    return frames


def make_chroma(width, height, n_frames, seed=0, bit_depth=8):
    """Deterministic Cb / Cr planes [n, (H+1)//2, (W+1)//2] for a 4:2:0 clip: a random texture that drifts frame by frame plus
    per-frame noise of different strength in the two planes, so that the chroma AC energy of a macroblock (which adaptive
    quantisation adds to the luma energy, ratecontrol.c:258-276) varies over the picture and over time."""
    rng = np.random.default_rng(1000 + seed)
    maxv = (1 << bit_depth) - 1
    dt = np.uint8 if bit_depth == 8 else np.uint16
    cw, ch = (width + 1) // 2, (height + 1) // 2
    base = rng.integers(0, maxv + 1, size=(ch, cw))
    smooth = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) // 4
    cb = np.stack([np.clip(np.roll(smooth, i, axis=1) + rng.integers(-9, 10, size=(ch, cw)), 0, maxv) for i in range(n_frames)])
    cr = np.stack([np.clip(np.roll(base, 2 * i, axis=0) // 2 + maxv // 4 + rng.integers(-30, 31, size=(ch, cw)), 0, maxv) for i in range(n_frames)])
    return cb.astype(dt), cr.astype(dt)


def upscaled_clip(W, H, n, depth, factor=4, **kw):
    """A picture sequence of BASELINE configs[3] / configs[4] size without minutes of numpy filtering: a (W/factor x H/factor)
    synthetic clip enlarged by sample repetition plus a little per-sample noise (so that neighbouring blocks differ and sub-pel
    positions matter).  Deterministic in its arguments: the full-size goldens (tests/golden/make_golden.py) are regenerated from
    the same call on the GPU box."""
    small = make_clip(W // factor, H // factor, n, bit_depth=depth, **kw)
    rng = np.random.default_rng(kw.get("seed", 1))
    out = np.empty((n, H, W), small.dtype)
    hi = (1 << depth) - 1
    for i in range(n):
        big = np.repeat(np.repeat(small[i], factor, axis=0), factor, axis=1).astype(np.int32)
        big += rng.integers(-6, 7, big.shape, dtype=np.int32)
        out[i] = np.clip(big, 0, hi).astype(small.dtype)
    return out
