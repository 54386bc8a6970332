import sys, time, numpy as np
sys.path.insert(0, '.')
import bench
from tagdust_amd import TagdustHip
for wl in ("c3", "c2"):
    bench._ACTIVE.clear(); bench._ACTIVE.update(bench.WORKLOADS[wl])
    model = bench.load_model()
    L = bench._ACTIVE["read_len"]; n = 1 << 20
    reads = bench.synth_batch(n, 1)
    offs = np.arange(n + 1, dtype=np.int64) * L
    c = TagdustHip(0); c.set_option("specialize", 1); c.upload_model(model)
    c.set_params(float(model["threshold"]), 16, 100)
    c.upload_batch(reads.reshape(-1), offs)
    for mode, name in ((1, "GET_LABEL"), (4, "GET_PROB"), (5, "ARCH_COMP(backward only)")):
        c.run(mode); c.sync()
        ms = []
        for _ in range(4):
            c.run(mode); c.sync(); ms.append(c.last_kernel_ms())
        print(wl, name, "%.2f ms" % (sum(ms) / len(ms)))
    c.set_params(float(model["threshold"]), 16, 0)
    c.run(1); c.sync(); c.run(1); c.sync(); print(wl, "GET_LABEL dust off %.2f ms" % c.last_kernel_ms())
    c.close()
