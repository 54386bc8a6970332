#!/usr/bin/env python3
"""Is the decode kernel's run-to-run spread (two modes ~7 % apart on one box) tied to the process, to the workspace
allocation, or to neither?  Several contexts in one process, each with a fresh 22 GB workspace, the same resident batch."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tagdust_amd import TagdustHip
bench.select_workload("c3")
model = bench.load_model()
n = 1 << 20
reads = bench.synth_batch(n, 5).reshape(-1)
offs = np.arange(n + 1, dtype=np.int64) * 150
ctxs = []
for r in range(5):
    c = TagdustHip(0)
    c.upload_model(model); c.set_params(float(model["threshold"]), 16, 100)
    c.upload_batch(reads, offs)
    c.run(); c.sync()
    ctxs.append(c)
# all five stay alive (five workspaces of 22 GB); the rounds interleave them, so a difference that follows the context is
# a property of its allocation, one that follows the clock is not
for rnd in range(4):
    line = []
    for r, c in enumerate(ctxs):
        ms = []
        for k in range(3):
            c.run(); ms.append(c.last_kernel_ms())
        line.append("ctx%d %.2f" % (r, min(ms)))
    print("round %d: %s" % (rnd, "  ".join(line)), flush=True)
for c in ctxs:
    c.close()
