#!/usr/bin/env python3
"""Config 5 host to host with the pipelined launches on half the wave slots each (two 172 GB workspaces do not fit) against
one stream (TD_OVERLAP=0).  usage: tools/c5_overlap.py [reads per batch] [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tagdust_amd import TagdustHip, RESULT_DTYPE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bench.select_workload("c5")
g = bench.load_model()
seq, offs = bench.synth_host_batch(n, 3)
base = None
for env in ({"TD_OVERLAP": "0"}, {}, {"TD_OVERLAP": "0"}, {}):
    os.environ.update(env)
    c = TagdustHip(0); c.set_option("pipeline_depth", 3)
    c.upload_model(g); c.set_params(float(g["threshold"]), 16, 100)
    outs = [(np.zeros(n, RESULT_DTYPE), np.zeros(len(seq), np.uint8)) for _ in range(4)]
    def go(k):
        t = []
        for s in range(k):
            t.append(c.submit(seq, offs, res=outs[s % 4][0], seq_out=outs[s % 4][1]))
            if len(t) >= 3: c.wait(t.pop(0))
        for x in t: c.wait(x)
    go(3); c.sync()
    t0 = time.perf_counter(); go(steps); c.sync(); dt = time.perf_counter() - t0
    same = True
    if base is None: base = (outs[0][0].copy(), outs[0][1].copy())
    else: same = base[0].tobytes() == outs[0][0].tobytes() and np.array_equal(base[1], outs[0][1])
    print("%-16s %6.2f M reads/s  overlap_active %d  wave slots %d  workspace %.0f GiB  identical %s" % (
        env or "defaults", n * steps / dt / 1e6, c.get_option("overlap_active"), c.batch_info()[2], c.batch_info()[1] / 2**30, same), flush=True)
    c.close()
    for k in env: os.environ.pop(k)
