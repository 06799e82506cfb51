"""Pictures handed over in HOST memory, the way x264_encoder_encode gets them (encoder/encoder.c:3368-3454, common/frame.c:445-447):
x264hip_lookahead_put_frames / x264hip_frame_put_batch with host pointers -- pinned buffers read by the DMA engines where they are,
pageable ones staged through the library's pinned ring -- and the one-picture road of x264hip_lookahead_put_frame( is_device = 0 ) give
the decisions and cost cells of the device-resident run; nothing waits for the compute stream on the way in."""
import ctypes as C

import numpy as np
import pytest

from x264_amd import lib
from x264_amd.synth import make_clip

pytestmark = pytest.mark.gpu


def _sig(outs, nb):
    return [(o.frame, o.type, [o.cost_est[i][j] for i in range(nb) for j in range(nb)]) for o in outs]


@pytest.mark.parametrize("W,H,depth", [(704, 576, 8), (1280, 720, 10)])
def test_host_pictures_equal_device_pictures(W, H, depth):
    import torch
    nf = 70
    frames = make_clip(W, H, nf, seed=17, bit_depth=depth, scene_cuts=(31,), fade=(45, 8, 0.7, 6), pan=(4, 2))
    cfg = lib.la_config(W, H, "medium", bit_depth=depth)
    nb = cfg["bframes"] + 2
    dev = torch.from_numpy(frames.view(np.int16) if depth > 8 else frames).cuda()
    pinned = torch.from_numpy(frames.view(np.int16) if depth > 8 else frames).pin_memory()
    pageable = np.ascontiguousarray(frames)
    want = None
    for name, ptrs in (("device", [dev[i].data_ptr() for i in range(nf)]), ("pinned", [pinned[i].data_ptr() for i in range(nf)]),
                       ("pageable", [pageable[i].ctypes.data for i in range(nf)])):
        for paced in (False, True):
            la = lib.Lookahead(cfg, max_frames=nf + 4)
            try:
                outs = la.run(device_ptrs=ptrs, stride=W, paced=paced)
                st = np.zeros(3, np.uint64)
                lib._ck(la.L.x264hip_host_transfer_stats(la.ctx_handle(), st.ctypes.data_as(C.c_void_p)), "host_transfer_stats")
                st2 = np.zeros(2, np.uint64)
                lib._ck(la.L.x264hip_host_transfer_stats2(la.ctx_handle(), st2.ctypes.data_as(C.c_void_p)), "host_transfer_stats2")
            finally:
                la.close()
            if want is None:
                want = _sig(outs, nb)
            assert _sig(outs, nb) == want, (name, paced)
            px = 1 if depth == 8 else 2
            if name == "device":
                assert tuple(st) == (0, 0, 0)
            elif name == "pinned":
                assert tuple(int(v) for v in st) == (nf * W * H * px, nf, 0)
                # batched: the clip is one allocation, so every group of sixteen pictures is ONE transfer; paced: a copy kernel per picture
                assert (int(st2[0]) >= nf // 16 and int(st2[1]) == 0) if not paced else int(st2[1]) == nf, (paced, st2)
            else:
                assert tuple(int(v) for v in st) == (nf * W * H * px, 0, nf)
                assert int(st2[0]) == 0 and (int(st2[1]) == nf if paced else True), (paced, st2)


def test_host_pictures_beside_another_context():
    """With a second context open on the device the ingest kernels of a put call are launched once behind its transfers (a launch per
    group beside other contexts' searches made the call several times as long), and a call of more groups than the ring of group buffers
    holds launches in between: 208 pictures = thirteen groups in ONE call, as one pinned allocation (merged transfers), as a pinned
    buffer per picture (straight into the slots) and pageable (staged) -- all equal to the device-resident run."""
    import torch
    W, H, nf = 352, 288, 208
    frames = make_clip(W, H, nf, seed=23, scene_cuts=(90,), fade=(140, 8, 0.7, 6), pan=(3, 1))
    cfg = lib.la_config(W, H, "medium")
    nb = cfg["bframes"] + 2
    dev = torch.from_numpy(frames).cuda()
    pinned = torch.from_numpy(frames).pin_memory()
    singles = [torch.from_numpy(frames[i].copy()).pin_memory() for i in range(nf)]
    pageable = np.ascontiguousarray(frames)
    other = lib.Lookahead(cfg, max_frames=8)  # a second context on the device: nothing is asked of it
    try:
        want = None
        for name, ptrs in (("device", [dev[i].data_ptr() for i in range(nf)]), ("pinned", [pinned[i].data_ptr() for i in range(nf)]),
                           ("pinned, a buffer per picture", [t.data_ptr() for t in singles]), ("pageable", [pageable[i].ctypes.data for i in range(nf)])):
            la = lib.Lookahead(cfg, max_frames=nf + 4)
            try:
                sigs = []
                for _ in range(2):  # the second pass finds every group buffer used
                    la.reset()
                    sigs.append(_sig(la.run(device_ptrs=ptrs, stride=W, paced=False), nb))
                st2 = np.zeros(2, np.uint64)
                lib._ck(la.L.x264hip_host_transfer_stats2(la.ctx_handle(), st2.ctypes.data_as(C.c_void_p)), "host_transfer_stats2")
            finally:
                la.close()
            if want is None:
                want = sigs[0]
            assert sigs[0] == want and sigs[1] == want, name
            if name == "pinned":
                assert int(st2[0]) == 2 * 13, st2  # thirteen whole-group transfers per pass
    finally:
        other.close()
