#!/usr/bin/env python3
"""End-to-end wall time, FASTQ file in -> demultiplexed FASTQ files out, through the library alone (parse, sequence
statistics, model, threshold calibration with 400 000 simulated reads, decode, write) and -- when a TagDust2 binary is
named with --reference PATH -- through that binary on the same file, same -seed, all host threads; the two runs must
write identical files.   usage: tools/e2e_pipeline.py [n_reads] [--reference PATH]"""
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from tagdust_amd import TagdustHip
from tagdust_amd import lib as tdlib


def write_fastq(path, reads):
    n, L = reads.shape
    rec = np.empty((n, 14 + 2 * L), np.uint8)
    names = np.char.zfill(np.arange(n).astype("U7"), 7)
    rec[:, :10] = np.frombuffer(("".join("@r%s\n" % x for x in names)).encode(), np.uint8).reshape(n, 10)
    rec[:, 10:10 + L] = np.frombuffer(b"ACGTN", np.uint8)[reads]
    rec[:, 10 + L:13 + L] = np.frombuffer(b"\n+\n", np.uint8)
    rec[:, 13 + L:13 + 2 * L] = ord("I")
    rec[:, 13 + 2 * L] = ord("\n")
    rec.tofile(path)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1 << 20
    exe = sys.argv[sys.argv.index("--reference") + 1] if "--reference" in sys.argv else None
    segs = ["B:" + ",".join(bench.BARCODES), "S:" + bench.SPACER, "R:N", "P:" + bench.ADAPTER]
    out = {"reads": n, "read_len": bench.READ_LEN, "arch": " ".join("-%d %s" % (k + 1, s) for k, s in enumerate(segs))}
    with tempfile.TemporaryDirectory() as tmp:
        fq = os.path.join(tmp, "in.fq")
        write_fastq(fq, bench.synth_batch(n, 99))
        out["fastq_bytes"] = os.path.getsize(fq)
        st = {}
        t0 = time.perf_counter()
        text = open(fq, "rb").read()
        st["read_file"] = time.perf_counter() - t0
        t = time.perf_counter(); pr = tdlib.ParsedReads(text, 0); st["parse"] = time.perf_counter() - t
        t = time.perf_counter(); c = TagdustHip(0); st["context (HIP runtime start-up)"] = time.perf_counter() - t
        t = time.perf_counter()
        thr = tdlib.estimate_threshold(c, segs, pr.codes, pr.offs, 0.1, seed=42, n_reads=400000, rng=0)
        st["calibration (emit 400k on host, compile + score on GPU)"] = time.perf_counter() - t
        t = time.perf_counter()
        model, _ = tdlib.build_model(segs, pr.codes, pr.offs, 0.05, 0.1)
        c.upload_model(model)
        st["model + kernel (cached compile)"] = time.perf_counter() - t
        c.set_params(thr, 16, 100)
        t = time.perf_counter(); c.upload_batch(pr.codes, pr.offs); st["pack + H2D"] = time.perf_counter() - t
        t = time.perf_counter(); c.run(); c.sync(); st["decode kernel"] = time.perf_counter() - t
        t = time.perf_counter(); res, _, seq_out = c.download(labels=False); st["D2H + unpack"] = time.perf_counter() - t
        t = time.perf_counter(); tdlib.write_demultiplexed(os.path.join(tmp, "own"), segs, pr, res, seq_out); st["write files"] = time.perf_counter() - t
        t = time.perf_counter(); c.close(); st["close (workspace free)"] = time.perf_counter() - t
        out["library"] = {"wall_s": time.perf_counter() - t0, "threshold": thr, "stages_s": {k: round(v, 3) for k, v in st.items()}}
        if exe and os.path.exists(exe):
            cores = min(len(os.sched_getaffinity(0)), 16)
            cmd = [exe, "-t", str(cores), "-seed", "42"]
            for k, s in enumerate(segs):
                cmd += ["-%d" % (k + 1), s]
            t = time.perf_counter()
            p = subprocess.run(cmd + [fq, "-o", os.path.join(tmp, "ref")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            out["reference"] = {"wall_s": time.perf_counter() - t, "threads": cores, "rc": p.returncode}
            own = {os.path.basename(f)[3:]: f for f in glob.glob(os.path.join(tmp, "own*.fq"))}
            ref = {os.path.basename(f)[3:]: f for f in glob.glob(os.path.join(tmp, "ref*.fq"))}
            same = set(own) == set(ref) and all(open(own[k], "rb").read() == open(ref[k], "rb").read() for k in own)
            out["identical_output_files"] = bool(same)
            out["speedup"] = out["reference"]["wall_s"] / out["library"]["wall_s"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
