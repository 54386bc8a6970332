#!/usr/bin/env python3
"""Streaming end to end (td_stream_run): a FASTQ file of COPIES x 2^20 bench reads -> demultiplexed files; stage rates as JSON.
usage: tools/e2e_stream.py [copies] [n_threads]     (default 16 copies = 5.3 GB of FASTQ)
       tools/e2e_stream.py --reference PATH [n_reads]   the reference binary (oracle/_ref/tagdust) on a file of n_reads distinct
           synthetic reads (default 2^19), then the library on the same file -- statistics, calibration (same -seed), model, streaming
           decode in 2^18-read batches -- and a byte comparison of every output file"""
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench


def against_reference(exe, n):
    import numpy as np
    from tagdust_amd import TagdustHip
    from tagdust_amd import lib as tdlib
    bench.select_workload("c3")
    segs = ["B:" + ",".join(bench.BARCODES), "S:" + bench.SPACER, "R:N", "P:" + bench.ADAPTER]
    out = {"reads": n, "arch": " ".join("-%d %s" % (k + 1, s) for k, s in enumerate(segs))}
    with tempfile.TemporaryDirectory() as tmp:
        fq = os.path.join(tmp, "in.fq")
        bench._write_fastq(fq, bench.synth_batch(n, 2024))
        cores = min(len(os.sched_getaffinity(0)), 16)
        cmd = [exe, "-t", str(cores), "-seed", "42"]
        for k, s in enumerate(segs):
            cmd += ["-%d" % (k + 1), s]
        t = time.perf_counter()
        p = subprocess.run(cmd + [fq, "-o", os.path.join(tmp, "ref")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        out["reference"] = {"wall_s": time.perf_counter() - t, "threads": cores, "rc": p.returncode}
        t0 = time.perf_counter()
        text = open(fq, "rb").read((1 << 20) * (14 + 2 * bench.READ_LEN))     # statistics / calibration see the first <= 1M reads
        pr = tdlib.ParsedReads(text, 0)
        c = TagdustHip(0)
        thr = tdlib.estimate_threshold(c, segs, pr.codes, pr.offs, 0.1, seed=42, n_reads=400000, rng=0)
        model, _ = tdlib.build_model(segs, pr.codes, pr.offs, 0.05, 0.1)
        pr.close()
        c.upload_model(model)
        c.set_params(thr, 16, 100)
        t1 = time.perf_counter()
        st = tdlib.stream_run(c, fq, segs, os.path.join(tmp, "own"))
        c.close()
        out["library"] = {"wall_s": time.perf_counter() - t0, "prologue_s (statistics, calibration, model, kernel)": t1 - t0,
                          "stream_s": st["wall_s"], "batches": st["n_batches"], "threshold": thr}
        own = {os.path.basename(f)[3:]: f for f in glob.glob(os.path.join(tmp, "own*.fq"))}
        ref = {os.path.basename(f)[3:]: f for f in glob.glob(os.path.join(tmp, "ref*.fq"))}
        out["output_files"] = sorted(own)
        out["identical_output_files"] = bool(ref) and set(own) == set(ref) and all(open(own[k], "rb").read() == open(ref[k], "rb").read() for k in own)
    return out


if "--reference" in sys.argv:
    i = sys.argv.index("--reference")
    exe = sys.argv[i + 1]
    n = int(sys.argv[i + 2]) if len(sys.argv) > i + 2 else 1 << 19
    print(json.dumps(against_reference(exe, n), indent=1))
else:
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print(json.dumps(bench.e2e_stream(0, copies=copies, n_threads=nt), indent=1))
