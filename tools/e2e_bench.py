#!/usr/bin/env python3
"""Host-inclusive throughput through the C-ABI: K contexts on one GPU, each driven by its own host thread through
upload (pack + H2D) -> run -> download (D2H + un-permute), so one context's host stages overlap another's kernel.
Reports reads/s including PCIe and host packing; never the bench `value`."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tagdust_amd import TagdustHip  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    batches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    model = bench.load_model()
    reads = bench.synth_batch(n, 11).reshape(-1)
    offs = np.arange(n + 1, dtype=np.int64) * bench.READ_LEN
    for k in (1, 2, 3):
        ctxs = []
        for _ in range(k):
            c = TagdustHip(0)
            c.upload_model(model)
            c.set_params(float(model["threshold"]), 16, 100)
            c.upload_batch(reads, offs)      # allocate workspaces before timing
            c.run()
            c.download(labels=False)
            ctxs.append(c)

        def worker(c, reps):
            for _ in range(reps):
                c.upload_batch(reads, offs)
                c.run()
                c.download(labels=False, seq=True)

        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(c, batches)) for c in ctxs]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        print("contexts=%d: %d batches of %d reads in %.2f s -> %.2f M reads/s host-inclusive" % (k, k * batches, n, dt, k * batches * n / dt / 1e6))
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
