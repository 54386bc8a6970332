#!/usr/bin/env python3
"""Position pruning on unfriendly inputs (config-3 architecture by default; usage: prune_stress.py [reads] [c3|c2|c5]): what do the fall-backs cost when the reads are not what
the architecture expects?  For each data set: kernel ms with pruning off / on, statistics, outputs compared byte for byte."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tagdust_amd import TagdustHip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
bench.select_workload(wl)
model = bench.load_model()
L = bench.READ_LEN
rng = np.random.default_rng(3)
good = np.ascontiguousarray(bench.synth_batch(n, 11)).reshape(n, L)


def ragged(a, lens):
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return np.concatenate([a[i, :lens[i]] for i in range(len(lens))]), offs

sets = {}
sets["bench mix (10 % unrelated)"] = (good.reshape(-1), np.arange(n + 1, dtype=np.int64) * L)
sets["all unrelated (uniform ACGT)"] = (rng.integers(0, 4, n * L, dtype=np.uint8), np.arange(n + 1, dtype=np.int64) * L)
polya = good.copy(); polya[:, 40:] = 0
sets["inserts replaced by poly-A from base 40"] = (polya.reshape(-1), np.arange(n + 1, dtype=np.int64) * L)
nn = good.copy(); nn[rng.random((n, L)) < 0.2] = 4
sets["20 % of all bases N"] = (nn.reshape(-1), np.arange(n + 1, dtype=np.int64) * L)
lens = rng.integers(20, L + 1, n)
sets["ragged 20..150 (sorted into tiles by the library)"] = ragged(good, lens)
short = rng.integers(12, 45, n)
sets["short reads 12..44 (mostly inside the cut)"] = ragged(good, short)
mix = good.copy(); mix[::2] = rng.integers(0, 4, (n + 1) // 2 * L, dtype=np.uint8).reshape(-1, L)
sets["half unrelated, interleaved"] = (mix.reshape(-1), np.arange(n + 1, dtype=np.int64) * L)

for name, (reads, offs) in sets.items():
    out = {}
    for prune in (0, 1):
        os.environ["TD_SPEC_PRUNE"] = str(prune)
        os.environ["TD_SPEC_PRUNE_STATS"] = str(prune)
        c = TagdustHip(0)
        c.upload_model(model); c.set_params(float(model["threshold"]), 16, 100)
        c.upload_batch(np.ascontiguousarray(reads), offs)
        ms = []
        for k in range(3):
            c.counts_reset(); c.run(); c.sync(); ms.append(c.last_kernel_ms())
        cnt = c.diag()
        out[prune] = (min(ms), c.download(), cnt)
        c.close()
    a, b = out[0][1], out[1][1]
    same = a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    cnt = out[1][2]
    tiles = max(int(cnt[236 - 192]), 1)
    print(wl + " %-52s off %7.2f ms  on %7.2f ms (x%.2f)  mean cut %5.1f  mean stop %5.1f  dense tiles %d / %d  restarted %d (bridges failed in %d, mean interval steps %.1f)  identical %s" % (
        name, out[0][0], out[1][0], out[0][0] / out[1][0], cnt[237 - 192] / tiles, cnt[227 - 192] / tiles, cnt[238 - 192], tiles,
        cnt[206 - 192], cnt[207 - 192], cnt[196 - 192] / max(int(cnt[197 - 192]), 1), same), flush=True)
