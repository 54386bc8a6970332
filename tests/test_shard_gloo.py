"""World-size-2 rehearsal of the multi-GPU path on CPU (gloo): static contiguous shard of the batch over
ranks, per-rank decoding (oracle stands in for the GPU here), all-reduce of the outcome / per-barcode
counters; the reduced counters must equal single-process counting."""
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
    assert np.array_equal(res["read_type"], g["read_type"][lo:hi])      # a shard decodes exactly like the whole batch
    local = shard.count_outcomes(res["read_type"], res["barcode"])
    total = shard.allreduce_counts(local, dist)
    if rank == 0:
        np.save(out, total)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_count_reduce(tmp_path):
    name = "c3_b6_s_r_p"
    out = str(tmp_path / "counts.npy")
    mp.spawn(_worker, args=(2, _free_port(), name, out), nprocs=2, join=True)
    g = load_golden(name)
    want = shard.count_outcomes(g["read_type"], g["barcode"])
    assert np.array_equal(np.load(out), want)
    assert want[:8].sum() == int(g["n_reads"])


def test_shard_bounds_cover_batch():
    for n in (0, 1, 7, 64, 1000, 1048576):
        for w in (1, 2, 3, 8):
            b = [shard.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
