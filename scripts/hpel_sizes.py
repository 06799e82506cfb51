import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x264_amd import lib
ctx = lib.Context(64, 64, bit_depth=8, max_frames=2, mv_range=32)
g = torch.Generator(device="cuda"); g.manual_seed(1)
for W, H in [(1920, 1080), (3840, 2160), (7680, 4320)]:
    hs = W + 64
    src = torch.randint(0, 256, (H + 16, hs), dtype=torch.uint8, device="cuda", generator=g)
    dst = torch.empty((3, H + 16, hs), dtype=torch.uint8, device="cuda")
    ho = 8 * hs + 32
    def run():
        ctx.hpel_filter(dst[0].data_ptr() + ho, dst[1].data_ptr() + ho, dst[2].data_ptr() + ho, src.data_ptr() + ho, hs, W, H)
    run(); ctx.synchronize()
    it = 50
    t0 = time.perf_counter()
    for _ in range(it): run()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / it
    a = torch.empty(2 * W * H, dtype=torch.uint8, device="cuda"); b = torch.empty(2 * W * H, dtype=torch.uint8, device="cuda")
    ctx.device_copy(b.data_ptr(), a.data_ptr(), a.numel()); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(it): ctx.device_copy(b.data_ptr(), a.data_ptr(), a.numel())
    ctx.synchronize()
    dc = (time.perf_counter() - t0) / it
    print("%dx%d hpel %.1f us %.0f GB/s | copy of the same 4*W*H bytes %.1f us %.0f GB/s" % (W, H, dt * 1e6, 4 * W * H / dt / 1e9, dc * 1e6, 4 * W * H / dc / 1e9))
ctx.close()
