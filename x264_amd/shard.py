"""GOP-segment sharding of a sequence across the GPUs of one node (one process per GPU) and the only
exchange the path needs: an all-gather of per-frame cost summaries (RCCL over xGMI with backend "nccl",
gloo on CPU for the tests)."""
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
