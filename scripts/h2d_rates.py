"""What a pinned-host-to-device transfer gets on this box: by size, on one stream and on eight at once; and the same with the library's
copy kernel reading the pinned buffer across PCIe.  (torch's copy_ from pinned memory is a hipMemcpyAsync.)"""
import time, torch, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def rate(nbytes, streams, reps):
    hs = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(streams)]
    ds = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(streams)]
    ss = [torch.cuda.Stream() for _ in range(streams)]
    for i in range(streams):
        with torch.cuda.stream(ss[i]):
            ds[i].copy_(hs[i], non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(streams):
            with torch.cuda.stream(ss[i]):
                ds[i].copy_(hs[i], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return nbytes * streams * reps / dt / 1e9, dt / reps * 1e6
for nbytes in (2 << 20, 8 << 20, 33 << 20, 256 << 20):
    for streams in (1, 2, 8):
        reps = max(4, min(200, (1 << 30) // (nbytes * streams)))
        g, us = rate(nbytes, streams, reps)
        print("hipMemcpyAsync %4d MiB x %d stream(s): %6.1f GB/s, %8.1f us per round" % (nbytes >> 20, streams, g, us))
# a kernel reading pinned memory across PCIe: the library's copy kernel (x264hip_device_copy) with the pinned buffer as its source
import ctypes as C
from x264_amd import lib
cfg = lib.la_config(704, 576, "medium")
la = lib.Lookahead(cfg, max_frames=8)
L, ctx = la.L, la.ctx_handle()
L.x264hip_device_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
for nbytes in (2 << 20, 33 << 20):
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    h.random_(0, 255)
    rc = L.x264hip_device_copy(ctx, d.data_ptr(), h.data_ptr(), nbytes)
    torch.cuda.synchronize()
    ok = bool((d.cpu() == h).all()) if rc == 0 else False
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        L.x264hip_device_copy(ctx, d.data_ptr(), h.data_ptr(), nbytes)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("copy kernel reading pinned memory %4d MiB: rc %d equal %s  %6.1f GB/s, %8.1f us each" % (nbytes >> 20, rc, ok, nbytes * reps / dt / 1e9, dt / reps * 1e6))
la.close()
