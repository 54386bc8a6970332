#!/usr/bin/env python3
"""Compact egress A/B/A/B in one process, one context (same workspaces): the bench pipeline with labels, option compact_egress 1 / 0
set per run.  usage: tools/egress_ab.py [steps] [workload]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TD_HOST_THREADS", "8")
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
n = 1 << 20
ctx, model, go, state, kernel_only, close, outs = bench.measure_workload(wl, n, steps, 5, 0, 1, 3, False, check=0, kernel_only_steps=3, labels=True)
for rnd in range(3):
    for ce, lab in (("1", True), ("0", True), ("1", False)):
        ctx.set_option("compact_egress", int(ce))
        ctx.sync(); t0 = time.perf_counter(); go(steps, lab); ctx.sync(); dt = time.perf_counter() - t0
        tl = bench.timeline_summary(state["timeline"])
        print("round %d compact %s labels %d: %.2f M reads/s, %.2f ms/step, start to start %.2f ms" % (rnd, ce, lab, n * steps / dt / 1e6, dt / steps * 1e3, tl["start_to_start_mean_ms"]), flush=True)
print("isolated kernel %.2f ms" % kernel_only()["kernel_ms"])
close()
