#!/usr/bin/env python3
"""Per-phase wave-clock shares of td_spec_kernel (TD_SPEC_PROFILE=1 diagnostic build; TD_SPEC_PROFILE=2 in the environment
splits the two sweeps by segment): usage tools/phase_profile.py [c3|c2|c5]"""
import os, sys
import numpy as np
os.environ.setdefault("TD_SPEC_PROFILE", "1")
LEVEL = int(os.environ["TD_SPEC_PROFILE"])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tagdust_amd import TagdustHip
NAMES = ["unpack", "backward", "forward", "bar_prob+random+Q", "label DP", "traceback", "extraction", "artifacts", "DUST", "outputs", "-", "between tiles"]
for wl in sys.argv[1:] or ["c3", "c2"]:
    bench._ACTIVE.clear(); bench._ACTIVE.update(bench.WORKLOADS[wl])
    model = bench.load_model()
    L = bench._ACTIVE["read_len"]; n = int(os.environ.get("PP_READS", 1 << 20))
    reads = bench.synth_batch(n, 1)
    c = TagdustHip(0); c.set_option("specialize", 1); c.upload_model(model)
    c.set_params(float(model["threshold"]), 16, 100)
    c.upload_batch(reads.reshape(-1), np.arange(n + 1, dtype=np.int64) * L)
    c.run(); c.sync(); c.counts_reset(); c.run(); c.sync()
    d = c.diag().astype(np.float64)
    t = d[240 - 192:252 - 192]
    print(wl, "kernel %.2f ms" % c.last_kernel_ms())
    total = t.sum() + (d[208 - 192:224 - 192].sum() if LEVEL == 2 else 0.0)
    for k, nm in enumerate(NAMES):
        if t[k]:
            print("   %-20s %5.1f %%" % (nm, 100 * t[k] / total))
    if LEVEL == 2:   # the sweeps' shares above then hold only what is left outside the segments
        for nm, o in (("backward", 208 - 192), ("forward", 216 - 192)):
            print("   %s by segment: %s" % (nm, "  ".join("%d: %.1f %%" % (j, 100 * d[o + j] / total) for j in range(8) if d[o + j])))
    c.close()
