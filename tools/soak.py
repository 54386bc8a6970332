#!/usr/bin/env python3
"""Soak test of the C-ABI on the GPU: many batches of changing size / read length / model through one context (and a
second context on another host thread), results compared with the first decode of the same reads, host and device
memory watched for growth.   usage: tools/soak.py [iterations]"""
import os
import sys
import threading

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_golden
from tagdust_amd import TagdustHip


def rss_mb():
    for l in open("/proc/self/status"):
        if l.startswith("VmRSS"):
            return int(l.split()[1]) / 1024


def worker(tag, iters, seed, errors):
    rng = np.random.default_rng(seed)
    fixtures = [load_golden(n) for n in ("c2_b4_r", "c3_b6_s_r_p", "umi_f_s_r", "scen2_endloss", "c5_b96_f_r_p")]
    c = TagdustHip(0)
    c.set_option("poison_workspace", 1)    # workspace, label-run and keep-bit tables filled with 0xFF before every launch / download
    ref = {}
    try:
        for it in range(iters):
            k = int(rng.integers(len(fixtures)))
            g = fixtures[k]
            if it % 5 == 0 or "cur" not in ref or ref["cur"] != k:
                c.set_option("specialize", int(rng.integers(0, 4) != 0))
                c.upload_model(g)
                c.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
                ref["cur"] = k
            n_all = int(g["n_reads"])
            n = min(n_all, int(rng.choice([0, 1, 63, 64, 65, n_all // 2, n_all])))
            pick = np.sort(rng.choice(n_all, n, replace=False)) if n else np.zeros(0, np.int64)
            reps = int(rng.choice([1, 1, 40, 400])) if n else 1          # sometimes a large batch of repeated reads
            offs_g = g["offs"]
            seqs = [g["seq"][offs_g[i]:offs_g[i + 1]] for i in pick] * reps
            offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
            seq = np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)
            if it % 7 == 3 and n:                   # now and then a probability-only pass over the batch first (MODE_GET_PROB: no
                from tagdust_amd.lib import MODE_GET_PROB   # labels, no runs), or a pipelined batch that asks for the records alone
                if it % 2:
                    c.upload_batch(seq, offs); c.run(MODE_GET_PROB); c.download(labels=False, seq=False)
                else:
                    from tagdust_amd import RESULT_DTYPE
                    r_only = np.zeros(len(offs) - 1, RESULT_DTYPE)
                    c.wait(c.submit(seq, offs, res=r_only))
                    if not np.array_equal(r_only["read_type"][:n], g["read_type"][pick]):
                        errors.append("%s: iteration %d: records-only submission differs" % (tag, it))
                        return
            if it % 2 == 0:                         # the synchronous calls ...
                c.upload_batch(seq, offs)
                c.run()
                res, labels, seq_after = c.download()
            else:                                   # ... and the pipelined ones, two batches in flight
                from tagdust_amd import RESULT_DTYPE
                outs = [(np.zeros(len(offs) - 1, RESULT_DTYPE), np.zeros(int(offs[-1]) + len(offs) - 1, np.int8), np.zeros(int(offs[-1]), np.uint8))
                        for _ in range(2)]
                tk = [c.submit(seq, offs, res=o[0], labels=o[1], seq_out=o[2]) for o in outs]
                for t_ in tk:
                    c.wait(t_)
                if outs[0][0].tobytes() != outs[1][0].tobytes() or not np.array_equal(outs[0][2], outs[1][2]):
                    errors.append("%s: iteration %d: two submissions of one batch differ" % (tag, it))
                    return
                res, labels, seq_after = outs[1]
            if n:   # labels and rewritten sequences of the first and the last repetition, byte for byte
                lab_g = np.concatenate([g["labels"][offs_g[i] + i:offs_g[i + 1] + i + 1] for i in pick])
                seq_g = np.concatenate([g["seq_after"][offs_g[i]:offs_g[i + 1]] for i in pick])
                nb = int(offs[n])
                for r in (0, reps - 1):
                    if not (np.array_equal(labels[r * (nb + n):(r + 1) * (nb + n)], lab_g) and np.array_equal(seq_after[r * nb:(r + 1) * nb], seq_g)):
                        errors.append("%s: iteration %d fixture %d n %d reps %d: labels / sequences differ" % (tag, it, k, n, reps))
                        return
            for r in range(0, reps, max(1, reps // 2)):
                sl = slice(r * n, (r + 1) * n)
                if not (np.array_equal(res["read_type"][sl], g["read_type"][pick]) and np.array_equal(res["barcode"][sl], g["barcode"][pick])
                        and np.array_equal(res["f_score"][sl].view(np.uint32), g["f_score"][pick].astype(np.float32).view(np.uint32))):
                    errors.append("%s: iteration %d fixture %d n %d reps %d differs" % (tag, it, k, n, reps))
                    return
    except Exception as e:                       # noqa: BLE001
        errors.append("%s: %r" % (tag, e))
    finally:
        c.close()


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free0, tot = ctypes.c_size_t(), ctypes.c_size_t()
    errors = []

    def two_threads(n, seed):
        th = [threading.Thread(target=worker, args=("t%d" % t, n, seed + t, errors)) for t in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    # warm-up with the same two threads: the runtime keeps what it allocates per hardware queue (the kernels' scratch backing,
    # ~1 GB per queue that has run the decode kernel) -- that is not growth
    two_threads(12, 0)
    hip.hipMemGetInfo(ctypes.byref(free0), ctypes.byref(tot))
    r0 = rss_mb()
    two_threads(iters, 100)
    free1 = ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(free1), ctypes.byref(tot))
    print("iterations per thread:", iters, "errors:", errors)
    print("device free before %.2f GB after %.2f GB; host RSS before %.0f MB after %.0f MB" % (free0.value / 1e9, free1.value / 1e9, r0, rss_mb()))
    if errors or free0.value - free1.value > 1 << 30:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
