"""World-size-2 rehearsal of the multi-GPU path on CPU (gloo).  The split and the counting are the library's own host-side C
functions (td_shard_bounds / td_count_outcomes of include/tagdust_multi.h, the ones td_multi_decode and the device
counters are checked against); only the per-read decode, which needs a GPU, is done by the oracle here.  The reduced
counters must equal single-process counting -- also for a fixture with artifact hits, whose outcome code carries the
sequence index in its upper bits."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden
from tagdust_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    g = load_golden(name)
    n = int(g["n_reads"])
    lo, hi = shard.shard_bounds(n, world, rank)
    offs = g["offs"][lo:hi + 1] - g["offs"][lo]
    seq = g["seq"][g["offs"][lo]:g["offs"][hi]]
    res, _, _ = pyoracle.label_batch(pyoracle.OracleModel(g), seq, offs, float(g["threshold"]), int(g["minlen"]), int(g["dust"]), 1)
    if "art_n" not in g:
        assert np.array_equal(res["read_type"], g["read_type"][lo:hi])  # a shard decodes exactly like the whole batch
    else:
        res = {"read_type": g["read_type"][lo:hi], "barcode": g["barcode"][lo:hi]}   # artifact hits need the filter: take the reference's
    local = shard.count_outcomes(res["read_type"], res["barcode"])
    total = shard.allreduce_counts(local, dist)
    if rank == 0:
        np.save(out, total)
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("name", ["c3_b6_s_r_p", "artifacts_b_r"])
def test_two_rank_shard_and_count_reduce(tmp_path, name):
    out = str(tmp_path / "counts.npy")
    mp.spawn(_worker, args=(2, _free_port(), name, out), nprocs=2, join=True)
    g = load_golden(name)
    want = shard.count_outcomes(g["read_type"], g["barcode"])
    assert np.array_equal(np.load(out), want)
    assert want[:8].sum() == int(g["n_reads"])
    # the same counting in numpy, outcome code = read_type & 7
    rt, bc = g["read_type"], g["barcode"]
    for code in range(8):
        assert want[code] == int(((rt & 7) == code).sum())
    if name == "artifacts_b_r":
        assert want[5] > 0 and (rt[(rt & 7) == 5] >> 8).min() >= 1
    ok = (rt == 0) & (bc >= 0)
    assert np.array_equal(want[8:], np.bincount(bc[ok] & 0xFF, minlength=256))


def test_shard_bounds_cover_batch():
    for n in (0, 1, 7, 64, 1000, 1048576):
        for w in (1, 2, 3, 8):
            b = [shard.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert all(b[i][1] - b[i][0] == n // w for i in range(w - 1))     # interval = n / world, remainder to the last
