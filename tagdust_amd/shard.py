"""Static sharding of a batch over ranks + the one collective of the path.

The analogue of run_pHMM()'s thread split (src/barcode_hmm.c:1911-1922): interval = n / world, rank r takes
[r*interval, (r+1)*interval), the last rank takes the remainder.  Per-read results need no exchange; only the
8 outcome + 256 per-barcode counters (barcode_hmm.c:354-384) are summed over ranks."""
import numpy as np

from .lib import NUM_COUNTERS


def shard_bounds(n_reads, world, rank):
    interval = n_reads // world
    lo = rank * interval
    hi = n_reads if rank == world - 1 else (rank + 1) * interval
    return lo, hi


def count_outcomes(read_type, barcode):
    """Host restatement of the device counters (td_counts_get): slot = outcome code, then barcode & 0xFF bins."""
    c = np.zeros(NUM_COUNTERS, np.int64)
    rt = np.asarray(read_type)
    for code in range(8):
        c[code] = int((rt == code).sum())
    bc = np.asarray(barcode)
    ok = (rt == 0) & (bc >= 0)
    c[8:] = np.bincount(bc[ok] & 0xFF, minlength=256)
    return c


def allreduce_counts(counts, dist, device=None):
    """Sum the counter vector over all ranks (RCCL when `device` is a GPU, gloo on CPU); returns numpy int64."""
    import torch
    t = torch.as_tensor(np.asarray(counts, np.int64), device=device)
    dist.all_reduce(t)
    return t.cpu().numpy()
