#!/usr/bin/env python3
"""bench.py -- throughput of the TagDust2 per-read HMM decoding hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`, or bare --
    `python bench.py --gpus N` with WORLD_SIZE unset starts its own N ranks as child processes before anything touches the
    GPU, relays rank 0's JSON line and exits non-zero if any rank does)

A *step* is one run_pHMM-sized call of the hot path over one batch of synthetic reads per GPU, host to host
(pack -> H2D -> kernels -> D2H -> extraction): the reads of a *fresh host batch* go in as base codes
(td_submit), the device sorts and packs them, runs the fused decode kernel (backward -> forward + posteriors ->
label DP -> Q -> extraction -> artifact filter -> DUST), rewrites the sequences and returns the per-read records and
the rewritten sequences AND the per-base labels (ri->labels, barcode_hmm.c:4503-4514) in input order to host buffers
(td_wait).  Batches are pipelined, so the copies of the neighbouring batches overlap the kernel -- `value` is the host-inclusive rate a caller of the C-ABI gets;
`roofline.kernel_ms` (HIP events around the decode kernel alone) and `extra.kernel_only` (the kernel on a resident
batch, what round 1 reported) stand beside it.  The workload is the configuration BASELINE.json's metric is quoted
on: 150 bp reads, 8-barcode architecture `-1 B:<8 of EDITTAG_6nt_ed_3> -2 S:GTA -3 R:N -4 P:AGATCGGAAGAGC`
(BASELINE.json configs[2]), in batches of --reads (default 2^20, the size of the reference's own batches,
barcode_hmm.c:172).  Reads shard with no data-path collective (weak scaling: every rank decodes its own batches);
the only exchange is one all-reduce of the 264 outcome / per-barcode counters per run (RCCL over xGMI, in place on
the library's device counters).

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# (as the library does when it is loaded -- see td_want_hw_queues in td_api.hip -- but before torch can initialise the runtime)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # ... of which streaming kernels reach about 6.3 TB/s (same guide)
READ_LEN = 150
BARCODES = ["TTGTGT", "AAAAAA", "AAACCC", "AAAGGG", "AAATTT", "AACACG", "AACCAT", "AACGTA"]  # first 8 of EDITTAG_6nt_ed_3
SPACER = "GTA"
ADAPTER = "AGATCGGAAGAGC"
WORKLOAD = ("config3: 150bp reads, arch -1 B:%s -2 S:%s -3 R:N -4 P:%s (BASELINE.json configs[2])"
            % (",".join(BARCODES), SPACER, ADAPTER))
_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


# other BASELINE.json configurations, selectable with --workload for development measurements (the contract's
# bench line is always the default, config 3)
WORKLOADS = {
    "c3": dict(fixture="c3_b6_s_r_p", read_len=150, barcodes=BARCODES, umi=0, spacer=SPACER, adapter=ADAPTER, name=WORKLOAD),
    "c2": dict(fixture="c2_b4_r", read_len=100, barcodes=["TGCT", "AAAA", "AACC", "AAGG", "AATT", "ACAC", "ACCA", "ACGT"],
               umi=0, spacer="", adapter="", simreads=True,
               name="config2: simreads -sim_barnum 8 -sim_readlen 96 -sim_random_frac 0.1 -sim_error_rate 0.02 (4 nt barcode + 96 nt read; the "
                    "random tenth is 96 nt), arch -1 B:<8 of EDITTAG_4nt_ed_2> -2 R:N (BASELINE.json configs[1])"),
    "c5": dict(fixture="c5_b96_f_r_p", read_len=150, barcodes=None, umi=8, spacer="", adapter=ADAPTER,
               name="config5: 150bp reads, arch -1 B:<96 of EDITTAG_6nt_ed_3> -2 F:NNNNNNNN -3 R:N -4 P:AGATCGGAAGAGC (BASELINE.json configs[4])"),
}
_ACTIVE = dict(WORKLOADS["c3"])


def select_workload(name):
    """Make `name` (c2 / c3 / c5) the workload synth_batch() and load_model() work on."""
    global READ_LEN
    _ACTIVE.clear()
    _ACTIVE.update(WORKLOADS[name])
    READ_LEN = _ACTIVE["read_len"]


def synth_host_batch(n, seed):
    """One host batch of the active workload: (base codes uint8, offsets int64).  Configs the reference's simreads can emit
    (config 2) come from the library's restatement of it (td_simreads: byte-identical to `simreads <tags> -seed <seed>
    -sim_barnum 8 -sim_readlen 96 -sim_readlen_mod 0 -sim_numseq <n> -sim_endloss 0 -sim_random_frac 0.1 -sim_error_rate
    0.02`), parsed by the library's FASTQ parser; the others from the generators below."""
    if _ACTIVE.get("simreads"):
        from tagdust_amd import lib as tdlib
        text = tdlib.simreads(_ACTIVE["barcodes"], seed=seed, barnum=len(_ACTIVE["barcodes"]), readlen=READ_LEN - len(_ACTIVE["barcodes"][0]),
                              numseq=n, random_frac=0.1, error_rate=0.02)
        pr = tdlib.ParsedReads(text, 0)
        codes, offs = pr.codes.copy(), pr.offs.copy()
        pr.close()
        return codes, offs
    return synth_batch(n, seed).reshape(-1), np.arange(n + 1, dtype=np.int64) * READ_LEN


def synth_batch(n, seed, read_len=None, random_frac=0.1, sub=0.02):
    if _ACTIVE["fixture"] != "c3_b6_s_r_p":
        return synth_batch_generic(n, seed, _ACTIVE, random_frac, sub)
    return synth_batch_c3(n, seed, READ_LEN, random_frac, sub)


def synth_batch_generic(n, seed, w, random_frac=0.1, sub=0.02):
    """[barcode][UMI][spacer] + uniform insert + 3' adapter prefix, 2 % substitutions, 10 % random reads."""
    rng = np.random.default_rng(seed)
    L = w["read_len"]
    bars = w["barcodes"]
    if bars is None:
        z = np.load(os.path.join(REPO, "tests", "golden", w["fixture"] + ".npz"))
        bars = str(z["seg_seqs"]).split(";")[0].split(",")[:-1]
    x = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    bar = np.array([[_CODE[c] for c in b] for b in bars], np.uint8)
    head = bar[rng.integers(0, len(bars), n)]
    if w["umi"]:
        head = np.concatenate([head, rng.integers(0, 4, size=(n, w["umi"]), dtype=np.uint8)], axis=1)
    if w["spacer"]:
        head = np.concatenate([head, np.tile(np.array([_CODE[c] for c in w["spacer"]], np.uint8), (n, 1))], axis=1)
    y = x.copy()
    y[:, :head.shape[1]] = head
    structured = np.zeros((n, L), bool)
    structured[:, :bar.shape[1]] = True
    if w["adapter"]:
        ad = np.array([_CODE[c] for c in w["adapter"]], np.uint8)
        keep = rng.integers(0, len(ad) + 1, n)
        pos = np.arange(L)[None, :]
        start = (L - keep)[:, None]
        in_ad = pos >= start
        y = np.where(in_ad, ad[np.clip(pos - start, 0, len(ad) - 1)], y)
        structured |= in_ad
    mut = structured & (rng.random((n, L)) < sub)
    y = np.where(mut, rng.integers(0, 4, size=(n, L), dtype=np.uint8), y)
    is_random = rng.random(n) < random_frac
    y[is_random] = x[is_random]
    return np.ascontiguousarray(y)


def synth_batch_c3(n, seed, read_len=READ_LEN, random_frac=0.1, sub=0.02):
    """simulate_reads-style synthetic reads for an architecture simreads cannot emit (S: segment):
    [barcode][GTA] + uniform insert + a prefix of the 3' adapter, 2 % substitutions, 10 % fully random
    reads.  Returns base codes (n, read_len) uint8."""
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 4, size=(n, read_len), dtype=np.uint8)
    bar = np.array([[_CODE[c] for c in b] for b in BARCODES], np.uint8)
    which = rng.integers(0, len(BARCODES), n)
    head = np.concatenate([bar[which], np.tile(np.array([_CODE[c] for c in SPACER], np.uint8), (n, 1))], axis=1)
    ad = np.array([_CODE[c] for c in ADAPTER], np.uint8)
    keep = rng.integers(0, len(ADAPTER) + 1, n)            # how much of the 3' adapter is present
    y = x.copy()
    y[:, :head.shape[1]] = head
    pos = np.arange(read_len)[None, :]
    start = (read_len - keep)[:, None]
    in_ad = pos >= start
    ad_idx = np.clip(pos - start, 0, len(ADAPTER) - 1)
    y = np.where(in_ad, ad[ad_idx], y)
    structured = np.zeros((n, read_len), bool)
    structured[:, :head.shape[1]] = True
    structured |= in_ad
    mut = structured & (rng.random((n, read_len)) < sub)
    y = np.where(mut, rng.integers(0, 4, size=(n, read_len), dtype=np.uint8), y)
    is_random = rng.random(n) < random_frac
    y[is_random] = x[is_random]
    return np.ascontiguousarray(y)


def algorithmic_bytes_per_read(L):
    # packed bases + N mask in, labels + (f_score, r_score, bar_prob) out
    return (L + 3) // 4 + (L + 7) // 8 + (L + 1) + 12


def load_model():
    """Model tables for the workload architecture, as built by the reference's init_model_bag() and
    committed as a fixture (tests/golden/c3_b6_s_r_p.npz); threshold = the reference's calibrated one."""
    z = np.load(os.path.join(REPO, "tests", "golden", _ACTIVE["fixture"] + ".npz"))
    return {k: z[k] for k in z.files}


def _write_fastq(path, reads):
    n, L = reads.shape
    rec = np.empty((n, 2 * L + 16), np.uint8)          # "@r0000000\n" + seq + "\n+\n" + qual + "\n"
    names = np.char.zfill(np.arange(n).astype("U7"), 7)
    rec[:, :10] = np.frombuffer(("".join("@r%s\n" % x for x in names)).encode(), np.uint8).reshape(n, 10)
    rec[:, 10:10 + L] = np.frombuffer(b"ACGTN", np.uint8)[reads]
    rec[:, 10 + L:13 + L] = np.frombuffer(b"\n+\n", np.uint8)
    rec[:, 13 + L:13 + 2 * L] = ord("I")
    rec[:, 13 + 2 * L] = ord("\n")
    rec[:, :14 + 2 * L].tofile(path)


def cpu_baseline(model, n_sample, seed):
    """The CPU path timed on this box's host cores on a bounded sample of the same workload, label phase only, same
    threshold as the GPU run, with all cores of the box's share and with one thread:
      kind "reference": the reference's own run_pHMM(MODE_GET_LABEL) (oracle/_ref/ref_time: ref_time.c linked against the
        unmodified reference sources; reads in memory, no calibration, no file output, and without the controller's
        per-read model rebuild of SURVEY.md quirk Q7, which is not part of the label phase) -- when the snapshot carries it;
      kind "port": the oracle (oracle/td_oracle.c, the pinned restatement), always reported beside it.
    Checker / baseline only: nothing here is on the product path."""
    import subprocess
    import tempfile
    from oracle import pyoracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("TD_CPU_THREADS", "16")))  # a one-GPU box's CPU share is 16 cores
    L = READ_LEN
    thr = float(model["threshold"])
    reads = synth_batch(n_sample, seed).reshape(n_sample, L)
    om = pyoracle.OracleModel(model)

    def port(n, threads):
        offs = np.arange(n + 1, dtype=np.int64) * L
        pyoracle.label_batch(om, reads[:256].reshape(-1), offs[:257], thr, 16, 100, threads)  # warm
        t0 = time.perf_counter()
        pyoracle.label_batch(om, reads[:n].reshape(-1), offs, thr, 16, 100, threads)
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "reads/s", "cores": threads, "reads": n, "seconds": dt}

    n1 = max(min(n_sample // 12, 20000), 256)
    out_port = {"kind": "port", "what": "oracle/td_oracle.c (pinned restatement of the reference's path), pthreads over contiguous ranges",
                "all_cores": port(n_sample, cores), "one_thread": port(n1, 1)}
    exe = os.path.join(REPO, "oracle", "_ref", "ref_time")
    out_ref = None
    if os.path.exists(exe) and _ACTIVE["fixture"] == "c3_b6_s_r_p":
        def ref(n, threads):
            with tempfile.TemporaryDirectory() as tmp:
                fq = os.path.join(tmp, "in.fq")
                _write_fastq(fq, reads[:n])
                cmd = [exe, repr(thr), "-t", str(threads), "-1", "B:" + ",".join(BARCODES), "-2", "S:" + SPACER, "-3", "R:N",
                       "-4", "P:" + ADAPTER, fq, "-o", os.path.join(tmp, "out")]
                p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
                if p.returncode != 0:
                    return None
                d = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
                return {"value": d["reads"] / d["seconds"], "unit": "reads/s", "cores": threads, "reads": d["reads"],
                        "seconds": d["seconds"], "extracted": d["extracted"]}
        try:
            ra, r1 = ref(n_sample, cores), ref(n1, 1)
            if ra and r1:
                out_ref = {"kind": "reference", "what": "the reference's run_pHMM(MODE_GET_LABEL) on reads in memory (oracle/_ref/ref_time), "
                           "calibrated threshold given, no calibration / file I/O / per-read model rebuild (quirk Q7) in the timed region",
                           "all_cores": ra, "one_thread": r1}
        except Exception as e:   # the baseline must not take the bench down
            out_ref = None
            sys.stderr.write("bench.py: reference baseline failed: %s\n" % e)
    head = out_ref or out_port
    out = {"value": head["all_cores"]["value"], "unit": "reads/s", "cores": cores, "kind": head["kind"],
           "sample": "%d reads of the same synthetic workload (label phase only, threshold %.4f as on the GPU), %s, %.1f s wall; "
                     "-t 1 on %d reads: %.0f reads/s" % (head["all_cores"]["reads"], thr, head["what"], head["all_cores"]["seconds"],
                                                          head["one_thread"]["reads"], head["one_thread"]["value"]),
           "one_thread": head["one_thread"], "port": out_port}
    if out_ref:
        out["reference"] = out_ref
    return out


def e2e_stream(dev_index, copies=4, n=1 << 20, n_threads=0, keep_dir=None):
    """FASTQ file in -> demultiplexed FASTQ files out through the library's streaming pipeline (td_stream_run: parse(k+1) ||
    decode(k) || write(k-1)) on the bench workload: a file of `copies` x `n` reads (the same synthetic batch
    written `copies` times), model and threshold given (no calibration in the timed call).  Returns the stage rates."""
    import tempfile
    from tagdust_amd import TagdustHip
    from tagdust_amd import lib as tdlib
    select_workload("c3")
    model = load_model()
    segs = ["B:" + ",".join(BARCODES), "S:" + SPACER, "R:N", "P:" + ADAPTER]
    with tempfile.TemporaryDirectory(dir=keep_dir) as tmp:
        one = os.path.join(tmp, "one.fq")
        _write_fastq(one, synth_batch(n, 4711))
        text = open(one, "rb").read()
        fq = os.path.join(tmp, "in.fq")
        with open(fq, "wb") as fh:
            for _ in range(copies):
                fh.write(text)
        os.remove(one)
        del text
        ctx = TagdustHip(dev_index)
        try:
            ctx.upload_model(model)
            ctx.set_params(float(model["threshold"]), 16, 100)
            # the context as a long-running caller holds it: kernel compiled, workspaces and batch slots of the run's geometry in
            # place (22 GB workspaces are allocated and probed once per context; a first file pays ~1-2 s for that)
            warm = os.path.join(tmp, "warm.fq")
            with open(fq, "rb") as src, open(warm, "wb") as dst:
                dst.write(src.read((1 << 19) * (14 + 2 * READ_LEN) * 2))
            tdlib.stream_run(ctx, warm, segs, os.path.join(tmp, "warm"), n_threads=n_threads)
            os.remove(warm)
            ctx.counts_reset()
            st = tdlib.stream_run(ctx, fq, segs, os.path.join(tmp, "out"), n_threads=n_threads)
            cnt = ctx.counts()
        finally:
            ctx.close()
        out_bytes = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp) if f.startswith("out"))
    r = st["n_reads"]
    rate = lambda sec: (r / sec) if sec > 0 else None
    return {"reads": r, "batches": st["n_batches"], "fastq_bytes_in": st["bytes_in"], "fastq_bytes_out": st["bytes_out"],
            "files_bytes_on_disk": out_bytes, "counters_add_up": bool(int(cnt[:8].sum()) == r),
            "wall_s": st["wall_s"], "value": rate(st["wall_s"]), "unit": "reads/s",
            "parse_stage_reads_per_s": rate(st["parse_s"]), "write_stage_reads_per_s": rate(st["write_s"]),
            "decode_thread_busy_s": st["decode_s"], "parse_busy_s": st["parse_s"], "write_busy_s": st["write_s"], "read_wait_s": st["read_s"],
            "wall_over_slowest_host_stage": (max(st["parse_s"], st["write_s"]) / st["wall_s"]) if st["wall_s"] > 0 else None,
            "what": "td_stream_run on a %d x %d-read FASTQ file in a temporary directory (page cache), %d batches (2^18 reads each: no "
                    "-ref filter, so the batching is free), model + threshold given, context warm (kernel compiled, workspaces and the "
                    "page-locked batch buffers of a previous file in place); wall includes mapping the input and creating the output files"
                    % (copies, n, st["n_batches"])}


def write_casava_files(dirname, n, seed=11):
    """BASELINE configs[3]: the reference's own index reads (dev/casava_read2.fastq.gz, kept as the fixture casava_index: 400
    records with their CASAVA-1.8 names and qualities) tiled to n records with renumbered y coordinates, and the two missing
    76-nt read files synthesised with matching names.  Returns (r1, r2, r3, the index file's segment list)."""
    z = np.load(os.path.join(REPO, "tests", "golden", "casava_index.npz"))
    names = bytes(z["names"]).split(b"\n")
    offs, n0 = z["offs"], int(z["n_reads"])
    alpha = np.frombuffer(b"ACGTN", np.uint8)
    head, ycoord, s2, q2 = [], [], [], []
    for j in range(n0):
        parts = names[j].split(b" ")[0].split(b":")
        head.append(b":".join(parts[:6]) + b":"); ycoord.append(int(parts[6]))
        s2.append(bytes(alpha[z["seq"][offs[j]:offs[j + 1]]])); q2.append(bytes(z["qual"][offs[j]:offs[j + 1]]))
    rng = np.random.default_rng(seed)
    paths = [os.path.join(dirname, "r%d.fq" % k) for k in (1, 2, 3)]
    qual = b"I" * 76
    with open(paths[0], "wb") as f1, open(paths[1], "wb") as f2, open(paths[2], "wb") as f3:
        for lo in range(0, n, 1 << 16):
            hi = min(n, lo + (1 << 16))
            bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (2, hi - lo, 76), dtype=np.uint8)]
            o1, o2, o3 = [], [], []
            for i in range(lo, hi):
                j, t = i % n0, i // n0
                base = head[j] + str(ycoord[j] + 100000 * t).encode()
                o2.append(b"@" + base + b" 2:N:0:\n" + s2[j] + b"\n+\n" + q2[j] + b"\n")
                # (now and then a low-complexity read in one of the plain files: run_rna_dust's DUST then decides the record's outcome)
                b1 = b"A" * 76 if i % 97 == 5 else bases[0, i - lo].tobytes()
                b3 = b"AC" * 38 if i % 101 == 7 else bases[1, i - lo].tobytes()
                o1.append(b"@" + base + b" 1:N:0:\n" + b1 + b"\n+\n" + qual + b"\n")
                o3.append(b"@" + base + b" 3:N:0:\n" + b3 + b"\n+\n" + qual + b"\n")
            f1.write(b"".join(o1)); f2.write(b"".join(o2)); f3.write(b"".join(o3))
    segs = [a for a in str(z["cmdline"]).split() if a.startswith("B:")]
    return paths[0], paths[1], paths[2], segs


def e2e_casava(dev_index, n=1 << 20, n_threads=0):
    """BASELINE configs[3] at the library level: three input files in lock-step through td_stream_run_multi -- the index reads
    decoded on the GPU (12 six-nt indexes + decoy), read 1 and read 3 not decoded (run_rna_dust), outcomes combined per record,
    every read written to the index's file (src/barcode_hmm.c:244-385).  Model and threshold of the fixture; rate in records/s."""
    import tempfile
    from tagdust_amd import TagdustHip
    from tagdust_amd import lib as tdlib
    z = np.load(os.path.join(REPO, "tests", "golden", "casava_index.npz"))
    model = {k: z[k] for k in z.files}
    with tempfile.TemporaryDirectory() as tmp:
        r1, r2, r3, segs = write_casava_files(tmp, n)
        ctx = TagdustHip(dev_index)
        try:
            ctx.upload_model(model)
            ctx.set_params(float(model["threshold"]), int(model["minlen"]), int(model["dust"]))
            files = [(r2, segs, [ctx]), (r1, ["R:N"], None), (r3, ["R:N"], None)]
            tdlib.stream_run_multi(files, os.path.join(tmp, "warm"), batch_reads=1 << 16, n_threads=n_threads)   # context warm, as in e2e_stream
            st, cnt = tdlib.stream_run_multi(files, os.path.join(tmp, "out"), n_threads=n_threads)
        finally:
            ctx.close()
    r = st["n_reads"]
    return {"records": r, "input_files": 3, "batches": st["n_batches"], "fastq_bytes_in": st["bytes_in"], "fastq_bytes_out": st["bytes_out"],
            "counters_add_up": bool(int(cnt[:8].sum()) == r), "extracted": int(cnt[0]), "wall_s": st["wall_s"],
            "value": r / st["wall_s"] if st["wall_s"] > 0 else None, "unit": "records/s (one record = one read of each of the three files)",
            "parse_busy_s": st["parse_s"], "write_busy_s": st["write_s"], "decode_thread_busy_s": st["decode_s"],
            "what": "td_stream_run_multi on the CASAVA three-read shape (BASELINE configs[3]): %d records, index reads decoded on the GPU, "
                    "two 76-nt read files written to the index's READ1 / READ2 files; model + threshold given, context warm" % r}


class _DevCounters:
    """The library's device counters (td_counts_device_ptr) as a CUDA-array-interface object, so that torch can wrap them
    without a copy and RCCL reduces them in place."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def run_pipelined(ctx, host_batches, outs, steps, depth, timeline=None):
    """`steps` host-to-host batches through td_submit / td_wait, `depth` in flight; returns per-batch kernel ms.  outs[k] =
    (records, rewritten sequences, labels or None).  timeline: a list that receives (start_ms, stop_ms, stream) of every
    batch's decode kernel against the context's timeline origin (HIP events on the kernel's own stream)."""
    tickets, k_ms = [], []
    trace = [] if os.environ.get("TD_BENCH_TRACE") else None

    def waited():
        k_ms.append(ctx.last_kernel_ms())
        if timeline is not None:
            timeline.append(ctx.last_kernel_times())

    for k in range(steps):
        o = outs[k % len(outs)]
        t0 = time.perf_counter()
        hb = host_batches[k % len(host_batches)]
        nb = len(hb[0])
        tickets.append(ctx.submit(hb[0], hb[1], res=o[0], seq_out=o[1][:nb], labels=None if o[2] is None else o[2][:nb + len(hb[1]) - 1]))
        t1 = time.perf_counter()
        if len(tickets) >= depth:
            ctx.wait(tickets.pop(0))
            waited()
        if trace is not None:
            trace.append((1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)))
    for t in tickets:
        ctx.wait(t)
        waited()
    if trace:
        sys.stderr.write("submit/wait ms per step: " + " ".join("%.1f/%.1f" % x for x in trace) + "\n")
    return k_ms


def timeline_summary(tl):
    """Start-to-start and overlap of consecutive decode launches from their HIP-event times (ms from one origin)."""
    if len(tl) < 2:
        return None
    tl = sorted(tl)
    starts = [t[0] for t in tl]
    stops = [t[1] for t in tl]
    n = len(tl)
    return {"launches": n, "origin": "td_timeline_origin right before the timed region; HIP events recorded on the launch's own compute stream "
                                      "right before / after the decode kernel (td_last_kernel_times)",
            "start_ms": [round(x, 3) for x in starts], "stop_ms": [round(x, 3) for x in stops], "stream": [t[2] for t in tl],
            "start_to_start_mean_ms": (starts[-1] - starts[0]) / (n - 1),
            "stop_to_stop_mean_ms": (stops[-1] - stops[0]) / (n - 1),
            "first_start_to_last_stop_ms": stops[-1] - starts[0],
            "span_mean_ms": sum(b - a for a, b in zip(starts, stops)) / n,
            "overlap_with_previous_mean_ms": sum(max(0.0, stops[i - 1] - starts[i]) for i in range(1, n)) / (n - 1)}


def measure_workload(name, n, steps, warmup, dev_index, specialize=1, depth=2, pinned=False, check=0, kernel_only_steps=0, labels=True):
    """One workload on this rank's GPU: model upload, oracle spot-check, warm-up, then the caller times `go()`."""
    from tagdust_amd import TagdustHip, RESULT_DTYPE
    from tagdust_amd.lib import PinnedArray
    select_workload(name)
    model = load_model()
    ctx = TagdustHip(dev_index)
    ctx.set_option("specialize", specialize)
    ctx.set_option("pipeline_depth", depth)
    if pinned:
        ctx.set_option("stable_input", 1)     # the input buffers below are never refilled: page-locked input keeps the compact egress
    ctx.upload_model(model)
    ctx.set_params(float(model["threshold"]), 16, 100)
    rank = int(os.environ.get("RANK", "0"))
    n_host = 3            # distinct host batches, handed over in turn: every step uploads reads the device does not hold
    pins = []

    def host_array(shape, dtype):
        if pinned:
            p = PinnedArray(shape, dtype)
            pins.append(p)
            return p.array
        return np.zeros(shape, dtype)     # pageable; zero-filled so that its pages exist before the timed region

    host_batches = []
    for b in range(n_host):
        codes, boffs = synth_host_batch(n, seed=1000 + 17 * rank + b)
        a = host_array((len(codes),), np.uint8)
        a[:] = codes
        host_batches.append((a, boffs))
    max_bases = max(len(h[0]) for h in host_batches)
    outs = [(host_array((n,), RESULT_DTYPE), host_array((max_bases,), np.uint8),
             host_array((max_bases + n,), np.int8) if labels else None) for _ in range(depth + 1)]

    if check and rank == 0:   # correctness spot-check against the oracle (outside the timed region)
        from oracle import pyoracle
        k = min(check, n)
        om = pyoracle.OracleModel(model)
        coffs = np.ascontiguousarray(host_batches[0][1][:k + 1])
        ccodes = np.ascontiguousarray(host_batches[0][0][:int(coffs[-1])])
        ores, olab, oseq = pyoracle.label_batch(om, ccodes, coffs, float(model["threshold"]), 16, 100, min(8, os.cpu_count() or 1))
        res = np.zeros(k, RESULT_DTYPE)
        lab = np.zeros(int(coffs[-1]) + k, np.int8)
        sq = np.zeros(int(coffs[-1]), np.uint8)
        ctx.wait(ctx.submit(ccodes, coffs, res=res, labels=lab, seq_out=sq))
        ok = (np.array_equal(lab, olab) and np.array_equal(res["read_type"], ores["read_type"]) and
              np.array_equal(res["barcode"], ores["barcode"]) and np.array_equal(sq, oseq) and
              np.array_equal(res["f_score"].view(np.uint32), ores["f_score"].view(np.uint32)) and
              np.allclose(res["mapq"], ores["Q"], rtol=0, atol=1e-4))
        if not ok:
            raise SystemExit("bench.py: HIP result differs from the oracle on the %s workload -- refusing to time it" % name)

    run_pipelined(ctx, host_batches, outs, max(warmup, 1), depth)
    ctx.sync()
    ctx.counts_reset()
    ctx.sync()

    state = {"k_ms": [], "timeline": []}

    def go(n_steps=None, with_labels=True):
        state["timeline"] = []
        o = outs if with_labels else [(a, b, None) for a, b, _ in outs]
        ctx.timeline_origin()
        state["k_ms"] = run_pipelined(ctx, host_batches, o, n_steps or steps, depth, state["timeline"])

    def kernel_only(n_steps=None):
        """The decode kernel alone over a resident batch (round 1's figure), outside the timed region."""
        n_steps = n_steps or kernel_only_steps
        if not n_steps:
            return None
        ctx.upload_batch(host_batches[0][0], host_batches[0][1])
        ms = []
        ctx.run(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            ctx.run()
            ms.append(ctx.last_kernel_ms())
        ctx.sync()
        dt = time.perf_counter() - t0
        return {"value": n * n_steps / dt, "unit": "reads/s", "kernel_ms": float(np.mean(ms)), "steps": n_steps,
                "note": "decode kernel alone, batch resident in HBM, no transfers (round 1's headline definition)"}

    def close():
        ctx.close()
        for p in pins:
            p.free()

    return ctx, model, go, state, kernel_only, close, outs


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as child processes -- before this process has
    imported torch or touched HIP, and without ever exec()ing -- with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would, relay rank 0's JSON line, and exit non-zero if any rank does."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TD_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)   # rank 0 prints its line last
    reader.start()
    rc = 0
    try:
        live = list(procs)
        while live:
            for q in list(live):
                c = q.poll()
                if c is None:
                    continue
                live.remove(q)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 1
                    for o in live:             # one rank failed: the others would wait for it in a barrier
                        o.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    reader.join(timeout=10)
    out0 = got[0] if got else b""
    lines = [l for l in out0.decode(errors="replace").splitlines() if l.startswith("{")]
    if rc == 0 and lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    elif rc == 0:
        rc = 1
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
    raise SystemExit(rc)


def bind_rank_to_numa(dev_index, local_rank, local_world):
    """Host side of one rank: its threads (this one, the library's copy pool, torch's) stay on the CPUs of the NUMA node next
    to its GPU -- as far as the process may use them; when the box does not say which node that is, the ranks share the
    allowed CPUs out evenly.  Returns a description for the bench line."""
    from tagdust_amd import lib as tdlib
    before = sorted(os.sched_getaffinity(0))
    node = tdlib.bind_host_to_device(dev_index) if os.environ.get("TD_BENCH_BIND", "1") != "0" else -1
    how = "NUMA node %d of the GPU" % node
    if node < 0:
        how = "unbound"
        if local_world > 1 and os.environ.get("TD_BENCH_BIND", "1") != "0":
            per = max(len(before) // local_world, 1)
            mine = before[(local_rank * per) % len(before):][:per]
            os.sched_setaffinity(0, mine)
            how = "no NUMA information: an even share of the allowed CPUs"
    after = sorted(os.sched_getaffinity(0))
    return {"binding": how, "cpus": len(after), "first_cpu": after[0], "last_cpu": after[-1]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60, help="timed steps (60 x 2^20 reads = 1.2 s per GPU: the pipeline's fill and drain are one step of them)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=1 << 20, help="reads per step per GPU")
    ap.add_argument("--cpu-sample", type=int, default=200000, help="reads in the CPU baseline sample (0 = skip)")
    ap.add_argument("--check", type=int, default=2048, help="reads verified against the oracle before timing (0 = skip)")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS), help="development only; the bench line is c3")
    ap.add_argument("--specialize", type=int, default=1, help="0 = generic ahead-of-time kernel")
    ap.add_argument("--depth", type=int, default=3, help="batches in flight (td_submit pipeline depth)")
    ap.add_argument("--pinned", type=int, default=1, help="1 = the caller's buffers are page-locked (td_host_alloc, under stable_input): no staging copies on the host -- the headline "
                    "since the end of round 4 (a rank's host work and DRAM traffic are what eight ranks share on one node; at one GPU the two modes are equal within "
                    "the box-to-box spread); 0 = pageable numpy arrays, staged through the library's pinned memory (the headline of rounds 3-4, reported beside it)")
    ap.add_argument("--labels", type=int, default=1, help="0 = the timed region does not download the per-base labels (rounds 1-2)")
    ap.add_argument("--sustained", type=int, default=200, help="steps of the extra sustained run (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, help="0 = skip the extra measurements (kernel only, pinned I/O, configs 2 and 5)")
    ap.add_argument("--isolated", type=int, default=0, help="N > 0: nothing but N isolated launches of the decode kernel on a resident batch "
                    "(what roofline.kernel_ms measures) -- the run tools/profile_lease.sh traces to hold rocprofv3's duration against it")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)      # does not return
    # Rank 0 prints ONE JSON line on stdout: everything else that writes to file descriptor 1 while this runs (RCCL's version
    # banner, for one) is sent to stderr, and the line goes to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch
    if os.environ.get("TD_BENCH_DRYRUN") == "1":
        # tests (no GPU): the launch plumbing alone -- rendezvous, barrier, one all-reduce, rank 0's line -- nothing is measured
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            real_stdout.write(json.dumps({"dryrun": True, "n_gpus": world, "rank_sum": int(t.item()),
                                          "local_ranks_seen": os.environ.get("LOCAL_WORLD_SIZE")}) + "\n")
            real_stdout.flush()
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    # one rank per GPU; TD_DIST_BACKEND=gloo lets several ranks rehearse the path on a one-GPU box
    backend = os.environ.get("TD_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    host = bind_rank_to_numa(dev_index, local_rank, local_world)
    # host threads of the library's copy pool (pageable caller memory <-> pinned staging, rebuilding sequences and labels from the
    # compact egress): the rank's share of the CPUs, at most 16 (the decode launch takes 15 ms per 2^20 reads now: the host's
    # 12 ms per step on 8 threads no longer hide behind it with room to spare)
    nthr = max(2, min(16, host["cpus"] // 2))
    if local_world > 1:
        # several ranks on one host share what the job may really use: a cgroup CPU quota below the affinity mask's width (the
        # pool's boxes: 16 of 256) must not be oversubscribed N times over -- every rank takes its share of it
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                nthr = max(2, min(nthr, int(float(q) / float(per)) // local_world))
        except Exception:
            pass
    host["copy_threads"] = nthr
    os.environ.setdefault("TD_HOST_THREADS", str(nthr))
    dist = None
    use_dist = world > 1 or os.environ.get("TD_BENCH_FORCE_DIST") == "1"   # the latter: a 1-rank RCCL group on a one-GPU box
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    from tagdust_amd import NUM_COUNTERS
    n = args.reads
    ctx, model, go, state, kernel_only, close, outs = measure_workload(
        args.workload, n, args.steps, args.warmup, dev_index, args.specialize, args.depth, bool(args.pinned),
        check=args.check, kernel_only_steps=5, labels=bool(args.labels))
    reduce_dev = torch.device("cuda", dev_index) if backend == "nccl" else None
    last_counts = [None]
    if args.isolated > 0:
        iso = kernel_only(args.isolated)
        if rank == 0:
            real_stdout.write(json.dumps({"isolated_launches": iso["steps"], "kernel": "td_spec_kernel" if args.specialize else "td_decode_kernel",
                                          "kernel_ms": iso["kernel_ms"], "reads_per_launch": n, "workload": _ACTIVE["name"],
                                          "kernel_reads_per_s": n / (iso["kernel_ms"] * 1e-3)}) + "\n")
            real_stdout.flush()
        close()
        if use_dist:
            dist.destroy_process_group()
        return

    def reduce_counts():
        # the path's only exchange: the 264 per-outcome / per-barcode counters, summed over ranks once per run, as the
        # reference counts once per run (barcode_hmm.c:354-384).  With RCCL the library's device counters are reduced
        # in place over xGMI; it is inside the timed region
        if not use_dist:
            return
        if backend == "nccl":
            ctx.sync()
            t = torch.as_tensor(_DevCounters(ctx.lib.td_counts_device_ptr(ctx.h), NUM_COUNTERS), device=reduce_dev)
            dist.all_reduce(t)
            last_counts[0] = t.cpu().numpy()
        else:
            t = torch.as_tensor(np.asarray(ctx.counts(), np.int64))
            dist.all_reduce(t)
            last_counts[0] = t.numpy()

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def timed(n_steps, with_labels=True, reduce=True):
        """EXACTLY n_steps steps between two fences; the MAX over ranks."""
        fence()
        t0 = time.perf_counter()
        go(n_steps, with_labels)
        if reduce:
            reduce_counts()
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=reduce_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    elapsed = timed(args.steps, bool(args.labels))
    if use_dist and last_counts[0] is not None and int(last_counts[0][:8].sum()) != n * args.steps * world:
        raise SystemExit("bench.py: reduced outcome counters (%d) do not add up to the reads decoded (%d)"
                         % (int(last_counts[0][:8].sum()), n * args.steps * world))
    ev_ms = state["k_ms"]
    timeline = timeline_summary(state["timeline"])
    overlap_active = ctx.get_option("overlap_active")
    prune_active = ctx.get_option("prune_active")

    # beside the headline, on every rank (these are collective timings): the same pipeline sustained over many more steps, so
    # that an outside clock / a GPU-busy sampler can see it, and the label-free variant rounds 1-2 reported
    extra = {}
    if args.extras and args.sustained > 0:
        dt = timed(args.sustained, bool(args.labels), reduce=False)
        tl = timeline_summary(state["timeline"])
        extra["sustained"] = {"value": n * args.sustained * world / dt, "unit": "reads/s", "steps": args.sustained, "seconds": dt,
                              "ms_per_step": dt / args.sustained * 1e3,
                              "start_to_start_mean_ms": tl["start_to_start_mean_ms"] if tl else None,
                              "overlap_with_previous_mean_ms": tl["overlap_with_previous_mean_ms"] if tl else None,
                              "note": "the headline's pipeline (labels included) over %d steps in one timed region" % args.sustained}
    if args.extras and args.labels:
        st = max(args.steps, 10)
        dt = timed(st, False, reduce=False)
        extra["no_labels"] = {"value": n * st * world / dt, "unit": "reads/s", "steps": st, "ms_per_step": dt / st * 1e3,
                              "note": "records + rewritten sequences only, no label download (the timed region of rounds 1-2)"}

    # The decode kernel's own launch duration: isolated launches over a resident batch, HIP events on the kernel's stream, in
    # this process right after the timed region.  Inside the timed region consecutive launches overlap on purpose (the next
    # batch's workgroups move in as the last one's retire), so the event span of a launch there includes the time it waits
    # for compute units and is reported separately.
    iso = kernel_only() if rank == 0 else None

    if rank == 0:
        total_reads = n * args.steps * world
        value = total_reads / elapsed
        k_ms = float(iso["kernel_ms"])
        bpr = algorithmic_bytes_per_read(READ_LEN)
        achieved = bpr * n / (k_ms * 1e-3) / 1e9
        nreads, ws_bytes, slots = ctx.batch_info()
        rec, traffic_note = load_pmc_record(n, args.workload)
        traffic = rec.get("hbm_bytes_per_launch") if rec else None
        kname = "td_spec_kernel" if args.specialize else "td_decode_kernel"
        out = {
            "metric": "reads/s (150 bp, 8-barcode arch)", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": _ACTIVE["name"], "read_len": READ_LEN, "reads_per_step_per_gpu": n,
                       "timed_region": "host to host: td_submit of a fresh host batch (base codes; 3 distinct host batches in turn) -> H2D -> "
                                       "device sort/pack -> decode kernel -> device un-permute/rewrite -> D2H of records + rewritten "
                                       "sequences%s (into %d rotating sets of host buffers) -> td_wait; "
                                       "%d batches in flight, the decode kernels of consecutive batches on two streams (the next one's "
                                       "workgroups move in as the last one's retire)" % (
                                           " + per-base labels (ri->labels, barcode_hmm.c:4503-4514)" if args.labels else "", len(outs), args.depth),
                       "labels_downloaded": bool(args.labels),
                       "host_buffers": "page-locked (td_host_alloc)" if args.pinned else "pageable numpy arrays (library stages through pinned memory)",
                       "host_threads": int(os.environ["TD_HOST_THREADS"]), "host_binding": host,
                       "parallelism": "static shard of reads over %d GPU(s), one process per GPU, counters all-reduced once per run (%s)" % (
                           world, ("RCCL" if backend == "nccl" else backend) if use_dist else "single rank: no exchange"),
                       "overlap_active": overlap_active, "prune_active": prune_active,
                       "wave_slots": slots, "workspace_bytes": ws_bytes},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kname, "kernel_ms": k_ms,
                         "kernel_ms_how": "average of %d isolated launches on a resident batch (HIP events, same process, after the timed region)" % iso["steps"],
                         "kernel_event_span_ms_in_timed_region": float(np.mean(ev_ms)),
                         "kernel_reads_per_s": n / (k_ms * 1e-3),
                         "host_inclusive_over_kernel": value / world / (n / (k_ms * 1e-3)),
                         "algorithmic_bytes_per_read": bpr, "reads_per_launch": n,
                         "traffic_gbps": (traffic / (k_ms * 1e-3) / 1e9) if traffic else None,
                         "traffic_frac_of_peak": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "traffic_frac_of_achievable": (traffic / (k_ms * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS) if traffic else None,
                         "lds": ({"idx_active_share": rec.get("lds_idx_active_share_of_cu_cycles"),
                                  "bank_conflict_share": rec.get("lds_bank_conflict_share"),
                                  "how": "SQ_LDS_IDX_ACTIVE / CU cycles of the launch, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (PMC passes of the "
                                         "same kernel source, profiles/): the logsum table gather"} if rec else None),
                         "traffic_source": traffic_note,
                         "valu": valu_roofline(rec, n, k_ms, args.workload),
                         "note": "algorithmic bytes are tiny (SURVEY.md 8d: not HBM-bound); the kernel's real HBM traffic is the "
                                 "backward-row spill (traffic_gbps), which no longer binds since the position pruning.  What binds is "
                                 "the compute side: `valu` prices the kernel's VALU wave-instructions (PMC, same kernel source) against "
                                 "the chip's issue peak"},
            "launch_timeline": timeline,
        }
        extra["kernel_only"] = iso
    close()
    other_key = "pageable_io" if args.pinned else "pinned_io"
    other_buffers = "pageable numpy arrays (library stages through pinned memory)" if args.pinned else "page-locked (td_host_alloc)"
    if args.extras and world > 1:
        # every rank again with the other kind of caller buffers (pageable: staged through the library's pinned memory; page-locked:
        # no staging copies at all), so that a host-side limit of the N-GPU run shows up as the difference between the two
        try:
            c2, m2, go2, st2, ko2, close2, _ = measure_workload(args.workload, n, 12, 3, dev_index, args.specialize, args.depth, not args.pinned,
                                                                 check=0, kernel_only_steps=0, labels=bool(args.labels))
            c2.sync(); dist.barrier()
            t1 = time.perf_counter()
            go2(12, bool(args.labels))
            c2.sync(); dist.barrier()
            t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=reduce_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            close2()
            if rank == 0:
                extra[other_key] = {"value": n * 12 * world / float(t.item()), "unit": "reads/s", "steps": 12, "host_buffers": other_buffers,
                                    "note": "all ranks, max over ranks; the headline uses the other kind of caller buffers (config.host_buffers)"}
        except Exception as e:
            if rank == 0:
                extra[other_key] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and args.extras and world == 1:
        # beside the headline: the same pipeline with page-locked caller buffers, and BASELINE configs[1] / configs[4]
        # (parity-test cases per the contract, timed here so that their rates are on the driver's record)
        for key, wl, nn, st, pin in (("c3_" + other_key, args.workload, n, args.steps, not args.pinned), ("config2", "c2", n, 16, False),
                                     ("config5", "c5", n // 4, 8, False)):
            try:
                c2, m2, go2, st2, ko2, close2, _ = measure_workload(wl, nn, st, 3, dev_index, args.specialize, args.depth, pin,
                                                                     check=512 if wl != args.workload else 0, kernel_only_steps=3,
                                                                     labels=bool(args.labels))
                c2.sync()
                t1 = time.perf_counter()
                go2()
                c2.sync()
                dt = time.perf_counter() - t1
                extra[key] = {"value": nn * st / dt, "unit": "reads/s", "steps": st, "reads_per_step": nn,
                              "kernel_ms": float(ko2()["kernel_ms"]), "workload": WORKLOADS[wl]["name"],
                              "overlap_active": c2.get_option("overlap_active"), "prune_active": c2.get_option("prune_active"),
                              "host_buffers": "page-locked" if pin else "pageable", "timed_region": "host to host, as the headline"}
                close2()
            except SystemExit as e:
                extra[key] = {"error": str(e)}
        select_workload(args.workload)
        try:
            extra["e2e"] = e2e_stream(dev_index)
        except Exception as e:      # the end-to-end extra must not take the bench line down
            extra["e2e"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            extra["config4"] = e2e_casava(dev_index)
        except Exception as e:
            extra["config4"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        out["extra"] = extra
        if args.cpu_sample and world == 1:
            out["cpu_baseline"] = cpu_baseline(model, args.cpu_sample, seed=77)
        elif args.cpu_sample:
            out["cpu_baseline"] = None
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


VALU_ISSUE_PEAK = 1.229e12   # wave64 VALU instructions per second: 256 CUs x 4 SIMDs x 2.4 GHz / 2 (= 157.3 TFLOP/s fp32 / 2 / 64)
VALU_NS_FAST, VALU_NS_SLOW = 1.0, 1.75   # tools/ubench/valu_rate.hip: ns per wave64 instruction per SIMD, the two issue classes of gfx950


# logsum evaluations of the reference per read that reach the table (the others have a -inf operand), measured
# there by instrumenting the reference; each costs ~10 lane operations + one LDS look-up in the formulation the survey prices
ALGORITHMIC_TABLE_LOGSUMS_PER_READ = {"c2": 17.3e3, "c3": 63.2e3, "c5": 527e3}
LANE_OPS_PER_LOGSUM = 10


def valu_roofline(rec, n, k_ms, workload="c3"):
    """Compute-side roofline of the decode kernel from the PMC record of the same kernel source (profiles/traffic.json):
    VALU wave-instructions per launch against the chip's issue peak, plain and weighted with the two issue classes the
    micro-benchmark finds (share of the slow class from the static instruction mix of the sweep loops)."""
    if not rec or not rec.get("valu_wave_insts_per_launch"):
        return None
    v = float(rec["valu_wave_insts_per_launch"])
    t = k_ms * 1e-3
    out = {"wave_insts_per_launch": v, "wave_insts_per_read": v / n, "issue_peak_per_s": VALU_ISSUE_PEAK,
           "achieved_per_s": v / t, "frac": v / t / VALU_ISSUE_PEAK,
           "lds_lookups_per_read": (rec["lds_insts_per_launch"] / n) if rec.get("lds_insts_per_launch") else None,
           "wait_any_share": rec.get("wait_any_share_of_wave_cycles"),
           "lds_bank_conflict_share": rec.get("lds_bank_conflict_share"),
           "source": "SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_WAIT_ANY of the PMC passes in profiles/ (same kernel source hash), this run's kernel_ms"}
    sh = rec.get("valu_slow_class_share")
    if sh is not None:
        ns = (1.0 - sh) * VALU_NS_FAST + sh * VALU_NS_SLOW
        out["slow_class_share"] = sh
        out["class_weighted_frac"] = (v / 1024.0) * ns * 1e-9 / t      # 1024 SIMDs
        out["class_weighted_how"] = "per SIMD: instructions x (%.2f ns fast class, %.2f ns slow class; tools/ubench/valu_rate.hip) / kernel time" % (
            VALU_NS_FAST, VALU_NS_SLOW)
    alg = ALGORITHMIC_TABLE_LOGSUMS_PER_READ.get(workload)
    if alg:
        # the reference's own work at this kernel's rate, against the same issue peak in lane operations (x 64 lanes): what the
        # executed count above leaves out is the work the position pruning / restarted sweeps do not do
        lane_ops = alg * LANE_OPS_PER_LOGSUM * n / t
        out["algorithmic"] = {"table_logsums_per_read": alg, "lane_ops_per_logsum": LANE_OPS_PER_LOGSUM,
                              "lane_ops_per_s": lane_ops, "frac_of_issue_peak": lane_ops / (VALU_ISSUE_PEAK * 64.0),
                              "executed_over_algorithmic": (v / n * 64.0) / (alg * LANE_OPS_PER_LOGSUM),
                              "how": "SURVEY.md 8(d): the reference's logsum evaluations that reach the table, per read, x 10 lane operations, "
                                     "x this kernel's reads/s, / (issue peak x 64 lanes)"}
    return out


def load_pmc_record(n, workload):
    """The PMC record of the dominant kernel committed under profiles/ (tools/profile_lease.sh collects it in the same lease
    as a bench run): HBM bytes per launch, VALU / LDS instruction counts, wait shares.  Only a record made for this workload,
    this batch size and this kernel source is used; anything else gives null rather than one build's counters over another
    build's time."""
    tpath = os.path.join(REPO, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, "no profiles/traffic.json"
    try:
        tj = json.load(open(tpath))
    except Exception as e:
        return None, "profiles/traffic.json unreadable: %s" % e
    recs = tj.get("records") or {tj.get("workload", "c3"): tj}
    r = recs.get(workload)
    if not r or r.get("reads_per_launch") != n:
        return None, "profiles/traffic.json has no record for this batch size / workload"
    if r.get("kernel_source_sha16") != kernel_source_sha16():
        return None, "profiles/traffic.json was collected with another kernel source (%s)" % r.get("kernel_source_sha16")
    return r, "profiles/traffic.json (%s, head %s, kernel %.2f ms in that lease)" % (
        r.get("collected", "?"), r.get("head", "?"), r.get("kernel_ms_same_lease", float("nan")))


def load_traffic(n, workload):
    rec, note = load_pmc_record(n, workload)
    return (rec.get("hbm_bytes_per_launch") if rec else None), note


def _code_only(text):
    """C / C++ source without its comments and with runs of white space collapsed (string and character literals kept as they
    are): what the compiler sees, give or take line numbers."""
    out = []
    i, n = 0, len(text)
    while i < n:
        ch = text[i]
        if ch == '"' or ch == "'":
            j = i + 1
            while j < n and text[j] != ch:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        elif ch in " \t\r\n":
            j = i
            while j < n and text[j] in " \t\r\n":
                j += 1
            out.append("\n" if "\n" in text[i:j] else " "); i = j
        else:
            out.append(ch); i += 1
    import re
    t = "".join(out)
    t = re.sub(r"[ \t]*\n[ \t\n]*", "\n", t)     # (a comment line leaves its line break behind: fold those too)
    return re.sub(r"[ \t]+", " ", t).strip()


def kernel_source_sha16():
    """Hash of the files the decode kernel is built from, comments and white space aside: ties a PMC record to the code it
    measured (a comment may be corrected without re-measuring; any token that reaches the compiler may not)."""
    import hashlib
    h = hashlib.sha256()
    for fn in ("td_spec_kernel.inc", "td_artifact.inc", "td_device.h", "td_jit.hip", "td_kernels.hip"):
        h.update(_code_only(open(os.path.join(REPO, "tagdust_amd", "csrc", fn), "r").read()).encode())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    main()
