import sys, os
import numpy as np
sys.path.insert(0, "/root/repo")
os.environ["TD_SPEC_EXTRA_OPTS"] = "-DTDS_ENDSTATS=1 " + (sys.argv[1] if len(sys.argv) > 1 else "")
import bench
from tagdust_amd import TagdustHip
for wl, n in (("c3", 1 << 20),):
    bench.select_workload(wl)
    model = bench.load_model()
    reads, offs = bench.synth_host_batch(n, 5)
    c = TagdustHip(0)
    c.upload_model(model); c.set_params(float(model["threshold"]), 16, 100)
    c.upload_batch(reads, offs)
    c.run(); c.sync()
    for rep in range(2):
        c.counts_reset()
        # counters are zeroed by reset: min slots need a large start value -> set via a first run? use max/ sum only
        c.run(); c.sync()
        k = c.diag()[200 - 192:206 - 192].astype(np.float64)
        ms = c.last_kernel_ms()
        # min slots are 0 after reset (atomicMin with 0 stays 0): use max start as reference instead
        cc = c.counts()
        print("   mean busy ms by XCD:", np.round(cc[8 + 208:8 + 216] / (k[5] / 8) / 1e5, 2).tolist(), " max by XCD:", np.round(cc[8 + 192:8 + 200] / 1e5, 2).tolist())
        print("   mean busy ms by wave in workgroup:", np.round(cc[8 + 216:8 + 224] / (k[5] / 8) / 1e5, 2).tolist())
        print("%s n=%d kernel %.2f ms: waves %d  last start -> mean end %.2f ms, last start -> last end %.2f ms" % (
            wl, n, ms, k[5], (k[4] / k[5] - k[1]) / 1e5, (k[3] - k[1]) / 1e5), flush=True)
    c.close()
