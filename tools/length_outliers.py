#!/usr/bin/env python3
"""Length classes: host-to-host rate of 2^20-read config-3 batches with 0.1 % reads of 1000 bases, against the uniform batch,
with and without the classes (TD_NO_LENGTH_CLASSES=1).  usage: tools/length_outliers.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tagdust_amd import TagdustHip, RESULT_DTYPE

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bench.select_workload("c3")
g = bench.load_model()
n, L = 1 << 20, bench.READ_LEN
rng = np.random.default_rng(5)
short = bench.synth_batch(n, 99)
lens = np.full(n, L, np.int64)
at = rng.choice(n, n // 1000, replace=False)
lens[at] = 1000
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
seq = rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)
# the short reads keep their architecture; the long ones are the same reads with a 1000-base insert
idx = np.repeat(np.arange(n), lens) ; pos = np.arange(int(offs[-1])) - offs[idx]
m = lens[idx] == L
seq[m] = short[idx[m], pos[m]]
head = pos < 9
seq[~m & head] = short[idx[~m & head], pos[~m & head]]
u_offs = np.arange(n + 1, dtype=np.int64) * L
u_seq = short.reshape(-1)


def rate(seq, offs, env):
    for k, v in env.items():
        os.environ[k] = v
    c = TagdustHip(0)
    c.set_option("pipeline_depth", 3)
    c.upload_model(g); c.set_params(float(g["threshold"]), 16, 100)
    outs = [(np.zeros(n, RESULT_DTYPE), np.zeros(int(offs[-1]), np.uint8)) for _ in range(4)]
    def go(k):
        t = []
        for s in range(k):
            t.append(c.submit(seq, offs, res=outs[s % 4][0], seq_out=outs[s % 4][1]))
            if len(t) >= 3: c.wait(t.pop(0))
        for x in t: c.wait(x)
    go(4); c.sync()
    t0 = time.perf_counter(); go(steps); c.sync(); dt = time.perf_counter() - t0
    info = (c.get_option("length_classes"), c.batch_info()[1] / 2**30, c.get_option("overlap_active"), c.batch_info()[2])
    c.close()
    for k in env: os.environ.pop(k)
    return n * steps / dt, info

for name, a in (("uniform 150", (u_seq, u_offs, {})), ("0.1% of 1000 nt, classes", (seq, offs, {})),
                ("0.1% of 1000 nt, one geometry", (seq, offs, {"TD_NO_LENGTH_CLASSES": "1"})), ("uniform 150", (u_seq, u_offs, {}))):
    r, info = rate(*a)
    print("%-32s %6.2f M reads/s   big slots %d, workspace %.1f GiB, overlap %d, wave slots %d" % (name, r / 1e6, *info), flush=True)
