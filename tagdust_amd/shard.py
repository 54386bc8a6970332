"""Static sharding of a batch over ranks + the one collective of the path.

The analogue of run_pHMM()'s thread split (src/barcode_hmm.c:1911-1922): interval = n / world, rank r takes
[r*interval, (r+1)*interval), the last rank takes the remainder.  Per-read results need no exchange; only the
8 outcome + 256 per-barcode counters (barcode_hmm.c:354-384) are summed over ranks.  Both helpers are the library's own
host-side C functions (include/tagdust_multi.h: td_shard_bounds, td_count_outcomes), the ones td_multi_decode uses."""
import numpy as np

from .lib import shard_bounds, count_outcomes as _count_outcomes, RESULT_DTYPE  # noqa: F401


def count_outcomes(read_type, barcode, lens=None):
    """Host restatement of the device counters: slot = read_type & 7 (an artifact hit is (sequence << 8) | 5), then
    barcode & 0xFF bins of the extracted reads."""
    res = np.zeros(len(read_type), RESULT_DTYPE)
    res["read_type"] = read_type
    res["barcode"] = barcode
    return _count_outcomes(res, lens)


def allreduce_counts(counts, dist, device=None):
    """Sum the counter vector over all ranks (RCCL when `device` is a GPU, gloo on CPU); returns numpy int64."""
    import torch
    t = torch.as_tensor(np.asarray(counts, np.int64), device=device)
    dist.all_reduce(t)
    return t.cpu().numpy()
