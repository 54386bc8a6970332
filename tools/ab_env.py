#!/usr/bin/env python3
"""A/B/A/B of kernel variants on one resident batch in one process (one box, one lease): each variant is a set of
environment settings read when the model is uploaded (TD_SPEC_* knobs, TD_SPEC_EXTRA_OPTS="-DTDS_...").
usage: ab_env.py <c3|c2|c5> "<NAME=VAL NAME=VAL ...>" "<variant 2>" ...   ("" = defaults); prints min kernel ms per round and
whether every variant's outputs equal the first one's."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tagdust_amd import TagdustHip
wl = sys.argv[1]
variants = sys.argv[2:]
n = int(os.environ.get("AB_READS", 1 << 20))
rounds = int(os.environ.get("AB_ROUNDS", 2))
bench.select_workload(wl)
model = bench.load_model()
reads, offs = bench.synth_host_batch(n, 5)
base = None
for rnd in range(rounds):
    for vi, v in enumerate(variants):
        kv = dict(x.split("=", 1) for x in v.split(";") if x)
        old = {k: os.environ.get(k) for k in kv}
        os.environ.update(kv)
        c = TagdustHip(0)
        c.upload_model(model); c.set_params(float(model["threshold"]), 16, 100)
        c.upload_batch(reads, offs)
        ms = []
        for k in range(5):
            c.run(); c.sync(); ms.append(c.last_kernel_ms())
        out = c.download()
        c.close()
        for k, o in old.items():
            if o is None: os.environ.pop(k, None)
            else: os.environ[k] = o
        if base is None:
            base = out
        same = out[0].tobytes() == base[0].tobytes() and np.array_equal(out[1], base[1]) and np.array_equal(out[2], base[2])
        print("%s round %d [%s]: %.2f ms  (%s)%s" % (wl, rnd, v or "defaults", min(ms), " ".join("%.1f" % m for m in ms), "" if same else "  OUTPUTS DIFFER"), flush=True)
