#!/usr/bin/env python3
"""Host-side throughput of the FASTQ ingest / egress functions (include/tagdust_io.h) on a synthetic 150-bp file."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tagdust_amd import lib as tdlib  # noqa: E402
import bench  # noqa: E402


def main(n=1 << 20):
    reads = bench.synth_batch(n, 3)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    lines = []
    q = b"I" * bench.READ_LEN
    t0 = time.time()
    asc = alpha[reads]
    text = b"".join(b"@READ%d;BARNUM:1\n" % i + asc[i].tobytes() + b"\n+\n" + q + b"\n" for i in range(n))
    print("built %.1f MB of FASTQ in %.1f s" % (len(text) / 1e6, time.time() - t0))
    for th in (1, 4, 16):
        t0 = time.perf_counter()
        pr = tdlib.ParsedReads(text, th)
        dt = time.perf_counter() - t0
        print("td_reads_parse  threads=%2d: %.2f s  %.2f GB/s  %.1f M reads/s" % (th, dt, len(text) / dt / 1e9, n / dt / 1e6))
        if th != 16:
            pr.close()
    res = np.zeros(n, tdlib.RESULT_DTYPE)
    res["barcode"] = np.random.RandomState(1).randint(0, 8, n)
    res["fingerprint"] = -1
    res["mapq"] = 33.3
    seq_out = pr.codes.copy()
    seq_out.reshape(n, -1)[:, :9] = 65
    segs = ["B:" + ",".join(bench.BARCODES), "S:GTA", "R:N", "P:" + bench.ADAPTER]
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        tdlib.write_demultiplexed(os.path.join(d, "out"), segs, pr, res, seq_out)
        dt = time.perf_counter() - t0
        sz = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
        print("td_writer_write (up to 16 threads): %.2f s  %.2f GB/s  %.1f M reads/s" % (dt, sz / dt / 1e9, n / dt / 1e6))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20)
