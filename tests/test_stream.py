"""td_stream_run (include/tagdust_io.h): the reference's batch loop around run_pHMM for one file as a pipeline.  CPU part:
the reader / parser stage alone (ctx NULL) must find the records td_reads_parse finds in the whole text -- the reference's
line state machine, io.c:1697-1799 -- whatever the block size, and cut them into batches of exactly batch_reads records.
GPU part: the files it writes equal those of the whole-text path and of the reference binary."""
import glob
import gzip
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO, load_golden


def _fnv(codes, offs):
    h = 1469598103934665603
    M = (1 << 64) - 1
    for i in range(len(offs) - 1):
        h = ((h ^ int(offs[i + 1] - offs[i])) * 1099511628211) & M
        for c in codes[offs[i]:offs[i + 1]].tolist():
            h = ((h ^ c) * 1099511628211) & M
    return h


def _ugly_fastq(n, seed, crlf=False):
    """Records that exercise the state machine: ragged lengths, N / lower case / IUPAC letters / '.', names with blanks and a
    tab (a control character ends the name), quality lines that start with '@' or '+'."""
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        L = int(rng.randint(1, 90))
        s = "".join(rng.choice(list("ACGTNacgtRY."), L, p=[.22, .22, .22, .22, .03, .01, .01, .01, .01, .02, .02, .01]))
        q = "".join(chr(c) for c in rng.randint(33, 74, L))
        if i % 7 == 0:
            q = "@" + q[1:]
        if i % 11 == 0:
            q = "+" + q[1:]
        name = "r%d len=%d" % (i, L) + ("\textra" if i % 13 == 0 else "")
        out.append("@%s\n%s\n+\n%s\n" % (name, s, q))
    t = "".join(out)
    return (t.replace("\n", "\r\n") if crlf else t).encode()


@pytest.mark.parametrize("crlf", [False, True], ids=["lf", "crlf"])
@pytest.mark.parametrize("block,batch,threads", [(4096, 777, 3), (50000, 1000, 1), (1 << 20, 100000, 4), (4096, 1, 2)])
def test_parse_stage_equals_whole_text_parse(tmp_path, block, batch, threads, crlf):
    from tagdust_amd import lib as tdlib
    text = _ugly_fastq(5000 if batch > 1 else 300, 5, crlf)
    path = str(tmp_path / "in.fq")
    open(path, "wb").write(text)
    pr = tdlib.ParsedReads(text, 1)
    st = tdlib.stream_run(None, path, batch_reads=batch, n_threads=threads, block_bytes=block)
    assert st["n_reads"] == pr.n and st["bytes_in"] == len(text)
    assert st["n_batches"] == (pr.n + batch - 1) // batch
    assert st["codes_fnv"] == _fnv(pr.codes, pr.offs)
    pr.close()


def test_compressed_and_fasta_input(tmp_path):
    """.gz goes through zcat like the reference's io_handler (io.c:382-608), i.e. through the pipe reader with its carry-over
    between blocks; FASTA records have no quality line."""
    from tagdust_amd import lib as tdlib
    text = _ugly_fastq(3000, 9)
    pr = tdlib.ParsedReads(text, 1)
    want = (pr.n, _fnv(pr.codes, pr.offs))
    pr.close()
    gz = str(tmp_path / "in.fq.gz")
    with gzip.open(gz, "wb") as fh:
        fh.write(text)
    st = tdlib.stream_run(None, gz, batch_reads=500, n_threads=2, block_bytes=8192)
    assert (st["n_reads"], st["codes_fnv"]) == want and st["n_batches"] == 6
    fa = b"".join(b">s%d some text\n%s\n" % (i, b"ACGTNNAC"[: 1 + i % 8]) for i in range(1000))
    path = str(tmp_path / "in.fa")
    open(path, "wb").write(fa)
    pr = tdlib.ParsedReads(fa, 1)
    st = tdlib.stream_run(None, path, batch_reads=64, n_threads=2, block_bytes=4096)
    assert st["n_reads"] == 1000 and st["codes_fnv"] == _fnv(pr.codes, pr.offs)
    pr.close()


def test_stream_errors_are_reported(tmp_path):
    from tagdust_amd import lib as tdlib
    from tagdust_amd import TdError
    with pytest.raises(TdError, match="cannot find input file"):
        tdlib.stream_run(None, str(tmp_path / "nope.fq"))
    bad = str(tmp_path / "bad.fq")
    open(bad, "wb").write(b"@a\nACGT\n+\nIIII\n@b\nACGTA\n+\nIII\n")
    with pytest.raises(TdError, match="sequence has 5 characters, base qualities 3"):
        tdlib.stream_run(None, bad)
    sam = str(tmp_path / "x.sam")
    open(sam, "wb").write(b"")
    with pytest.raises(TdError, match="samtools"):      # (this image has none on PATH)
        tdlib.stream_run(None, sam)
    empty = str(tmp_path / "empty.fq")
    open(empty, "wb").write(b"")
    st = tdlib.stream_run(None, empty)
    assert st["n_reads"] == 0 and st["n_batches"] == 0


def test_truncated_gz_is_an_error(tmp_path):
    """A .gz cut short ends zcat with a non-zero status: the run fails instead of returning TD_OK on a partial input."""
    from tagdust_amd import lib as tdlib
    from tagdust_amd import TdError
    gz = str(tmp_path / "cut.fq.gz")
    with gzip.open(gz, "wb") as fh:
        fh.write(_ugly_fastq(4000, 21))
    data = open(gz, "rb").read()
    open(gz, "wb").write(data[: len(data) // 2])
    with pytest.raises(TdError, match="zcat.*failed.*truncated or corrupt input"):
        tdlib.stream_run(None, gz, batch_reads=500, n_threads=2, block_bytes=8192)


def test_pipeline_with_too_few_batch_buffers_does_not_hang(tmp_path):
    """Page-locking can fail part-way (memlock limit, container): the helper thread that adds batch buffers then stops and the
    pipeline runs with what it has.  With fewer batches than the in-flight depth + 2 it used to wait forever (reader for a free
    batch, device thread for a ready one, writer for a finished one).  The streaming code as it is, device calls stubbed
    (tools/tsan_stream/stub.cpp), the allocator cut off after N allocations: every N ends -- with the same files as an
    unlimited run, or with the 'exhausted' error when not even one batch could be had."""
    exe = str(tmp_path / "stream_stub")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(REPO, "include"), "-w", "-o", exe,
                           os.path.join(REPO, "tools", "tsan_stream", "stub.cpp"), os.path.join(REPO, "tagdust_amd", "csrc", "td_stream.cpp"),
                           os.path.join(REPO, "tagdust_amd", "csrc", "td_fastq.cpp")])
    rng = np.random.RandomState(4)
    fq = str(tmp_path / "in.fq")
    with open(fq, "wb") as fh:
        for i in range(20000):
            L = int(rng.randint(20, 90))
            fh.write(("@r%d\n%s\n+\n%s\n" % (i, "".join(rng.choice(list("ACGT"), L)), "I" * L)).encode())
    os.makedirs("/tmp/td_tsan", exist_ok=True)       # (the stub's output prefix)

    def run(limit):
        env = dict(os.environ)
        env.pop("TD_STUB_ALLOC_LIMIT", None)
        if limit is not None:
            env["TD_STUB_ALLOC_LIMIT"] = str(limit)
        r = subprocess.run([exe, fq, "777", "2", "20000", "decode"], env=env, capture_output=True, timeout=120)   # a hang fails here
        files = {os.path.basename(f): open(f, "rb").read() for f in sorted(glob.glob("/tmp/td_tsan/out*.fq"))}
        return r.stdout.decode().splitlines()[0], files

    want_line, want_files = run(None)
    assert want_line.startswith("rc 0 reads 20000 ")
    for limit in (4, 6, 8, 10, 12, 16, 20):
        line, files = run(limit)
        if line.startswith("rc 0"):
            assert line == want_line and files == want_files, limit
        else:
            assert line.startswith("rc 1"), (limit, line)


def test_rq_formatting_equals_printf():
    """The writer formats ";RQ:%0.2f" itself (float x 100 is exact in double, nearbyint rounds half to even like printf rounds
    the exact expansion).  Against C's printf (ctypes) on random floats of every magnitude a Q takes, on every tie k/200 +- 1 ulp,
    and on the values that must fall back to printf (negative, -0, inf, nan, huge)."""
    import ctypes as C
    from tagdust_amd import lib as tdlib
    lib = tdlib.load_library()
    lib.td_format_q.argtypes = [C.c_float, C.c_char_p]
    libc = C.CDLL(None)
    libc.snprintf.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_double]
    rng = np.random.default_rng(3)
    vals = [rng.random(150000, dtype=np.float32) * np.float32(s) for s in (1, 10, 100, 1000, 1e6)]
    ties = (np.arange(0, 40000, dtype=np.float64) / 200.0).astype(np.float32)
    vals += [ties, np.nextafter(ties, np.float32(1e9)), np.nextafter(ties, np.float32(-1))]
    vals.append(np.array([0.0, -0.0, -0.004, -1.5, np.inf, -np.inf, np.nan, 3e38, 1e15, 99999.995, 0.005, 0.015, 0.025, 1.005], np.float32))
    a, b = C.create_string_buffer(64), C.create_string_buffer(400)
    for v in np.concatenate(vals).tolist():
        n = lib.td_format_q(v, a)
        libc.snprintf(b, 400, b"%0.2f", float(np.float32(v)))
        assert a.raw[:n] == b.value, (v, a.raw[:n], b.value)


# ---------------------------------------------------------------------------------------------------------------------
RBIN = os.path.join(REPO, "oracle", "_ref")


def _files(d, prefix):
    return {os.path.basename(p)[len(prefix):]: open(p, "rb").read() for p in sorted(glob.glob(os.path.join(d, prefix + "*.fq")))}


def _segments(g):
    a = str(g["cmdline"]).split()
    return [a[i + 1] for i in range(len(a) - 1) if a[i].startswith("-") and a[i][1:].isdigit()]


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch,block", [("c3_b6_s_r_p", 100, 4096), ("c2_indel_varlen", 64, 20000), ("b_r_s_r", 1000001, 0), ("umi_f_s_r", 37, 4096)])
def test_stream_writes_the_files_of_the_whole_text_path(tmp_path, name, batch, block):
    """Several small batches through the pipeline (reader, device, writer all busy at once) write byte for byte what
    td_reads_parse -> td_batch_upload / td_run / td_batch_download -> td_writer_write write for the text at once -- which
    tests/test_io.py and the drop-in tests tie to the reference binary's files."""
    from test_dropin_gpu import _write_fastq
    from tagdust_amd import TagdustHip
    from tagdust_amd import lib as tdlib
    g = load_golden(name)
    fq = str(tmp_path / "in.fq")
    _write_fastq(g, fq)
    segs = _segments(g)
    text = open(fq, "rb").read()
    c = TagdustHip(0)
    try:
        c.upload_model(g)
        c.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
        pr = tdlib.ParsedReads(text, 1)
        c.upload_batch(pr.codes, pr.offs)
        c.run()
        res, _, seq_out = c.download(labels=False)
        tdlib.write_demultiplexed(str(tmp_path / "whole"), segs, pr, res, seq_out)
        pr.close()
        c.counts_reset()
        st = tdlib.stream_run(c, fq, segs, str(tmp_path / "stream"), batch_reads=batch, n_threads=3, block_bytes=block)
        cnt = c.counts()
    finally:
        c.close()
    n = int(g["n_reads"])
    assert st["n_reads"] == n and st["n_batches"] == (n + batch - 1) // batch
    assert int(cnt[:8].sum()) == int((g["lens"] > 0).sum())
    a, b = _files(str(tmp_path), "whole"), _files(str(tmp_path), "stream")
    assert a and set(a) == set(b)
    for k in a:
        assert a[k] == b[k], "output file *%s differs" % k
    assert st["bytes_out"] == sum(len(v) for v in b.values())


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(RBIN, "tagdust_rtest")), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("threads", [1, 3])
def test_stream_equals_the_reference_binary_with_artifact_filter(tmp_path, threads):
    """The -DRTEST reference reads batches of 1000 records (barcode_hmm.c:165-175); with -ref the artifact filter's thread
    ranges -- and with them which routine scores a read -- are taken per batch.  The fixture's reads, five times over (1015
    reads: one full batch and a tail), through td_stream_run with batch_reads = 1000 must give the reference binary's files."""
    from test_dropin_gpu import _write_fastq
    from tagdust_amd import TagdustHip
    from tagdust_amd import lib as tdlib
    g = load_golden("artifacts_b_r")
    one = str(tmp_path / "one.fq")
    _write_fastq(g, one)
    fq, fa = str(tmp_path / "in.fq"), str(tmp_path / "art.fa")
    open(fq, "wb").write(open(one, "rb").read() * 5)
    open(fa, "wb").write(bytes(g["art_fasta_text"]))
    args = str(g["cmdline"]).split()
    args[args.index("-ref") + 1] = fa
    p = subprocess.run([os.path.join(RBIN, "tagdust_rtest")] + args + ["-t", str(threads), fq, "-o", "cpu"], cwd=str(tmp_path),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-1500:]
    art = tdlib.parse_fasta(open(fa, "rb").read())
    segs = _segments(g)
    # sequence statistics, model and calibrated threshold like the reference's own prologue: get_sequence_stats() reads batches
    # until it has seen more than 1 000 000 reads (io.c:125-188) -- this file is shorter, so all of it
    head = tdlib.ParsedReads(open(fq, "rb").read(), 1)
    hc, ho = head.codes.copy(), head.offs.copy()
    head.close()
    c = TagdustHip(0)
    try:
        thr = tdlib.estimate_threshold(c, segs, hc, ho, float(g["d"]), seed=42, n_reads=4000, rng=1)
        model, _ = tdlib.build_model(segs, hc, ho, 0.05, float(g["d"]))
        c.set_artifacts(art[0], art[1], int(g["art_filter_error"]), threads)
        c.upload_model(model)
        c.set_params(thr, int(g["minlen"]), int(g["dust"]))
        st = tdlib.stream_run(c, fq, _segments(g), str(tmp_path / "gpu"), batch_reads=1000, n_threads=2, block_bytes=30000)
    finally:
        c.close()
    assert st["n_reads"] == 5 * int(g["n_reads"])
    a, b = _files(str(tmp_path), "cpu"), _files(str(tmp_path), "gpu")
    assert a and set(a) == set(b)
    for k in a:
        assert a[k] == b[k], "output file *%s differs" % k


@pytest.mark.gpu
def test_stream_at_full_batch_size(tmp_path):
    """Three batches' worth of the bench workload (2.5 x 2^20 reads, 0.8 GB of FASTQ) with the reference's batch size: the
    output of the concatenated file is the concatenation of the outputs of its parts (whole-text path, one part per call),
    and the device counters add up."""
    import bench
    from tagdust_amd import TagdustHip
    from tagdust_amd import lib as tdlib
    bench.select_workload("c3")
    model = bench.load_model()
    segs = ["B:" + ",".join(bench.BARCODES), "S:" + bench.SPACER, "R:N", "P:" + bench.ADAPTER]
    parts = [(1 << 20, 11), (1 << 20, 12), (1 << 19, 13)]
    fq = str(tmp_path / "in.fq")
    c = TagdustHip(0)
    try:
        c.upload_model(model)
        c.set_params(float(model["threshold"]), 16, 100)
        want = {}
        with open(fq, "wb") as fh:
            for k, (n, seed) in enumerate(parts):
                part = str(tmp_path / ("part%d.fq" % k))
                bench._write_fastq(part, bench.synth_batch(n, seed))
                text = open(part, "rb").read()
                fh.write(text)
                pr = tdlib.ParsedReads(text, 8)
                c.upload_batch(pr.codes, pr.offs)
                c.run()
                res, _, seq_out = c.download(labels=False)
                tdlib.write_demultiplexed(str(tmp_path / ("p%d" % k)), segs, pr, res, seq_out)
                pr.close()
                for name, data in _files(str(tmp_path), "p%d" % k).items():
                    want[name] = want.get(name, b"") + data
                for f in glob.glob(str(tmp_path / ("p%d*" % k))) + [part]:
                    os.remove(f)
                del text
        c.counts_reset()
        st = tdlib.stream_run(c, fq, segs, str(tmp_path / "out"), batch_reads=1000001)
        cnt = c.counts()
    finally:
        c.close()
    total = sum(n for n, _ in parts)
    assert st["n_reads"] == total and st["n_batches"] == 3 and int(cnt[:8].sum()) == total
    got = _files(str(tmp_path), "out")
    assert set(got) == set(want)
    for k in want:
        assert got[k] == want[k], "output file *%s differs" % k


def _thr_lines(path):
    out = {}
    for l in open(path).read().splitlines():
        f = l.split("\t")
        if len(f) >= 3 and f[1].strip().replace(".", "").replace("%", "").isdigit():
            out[f[2].strip()] = f[1].strip()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_ctx", [(1 << 20, 1), (70001, 2)], ids=["2^20-records", "two-contexts"])
def test_multi_file_stream_equals_the_reference_binary(tmp_path, n, n_ctx):
    """BASELINE configs[3] at the library level: the CASAVA three-read shape -- the reference's own index reads (fixture
    casava_index) tiled to n records with the two 76-nt read files synthesised around them -- through td_stream_run_multi: three
    readers in lock-step, the index file decoded (over n_ctx contexts: run_pHMM's contiguous ranges), reads 1 and 3 through
    run_rna_dust, outcomes combined per record (barcode_hmm.c:329-351), print_all's READ1 / READ2 files.  Everything from the
    library alone (statistics over the reference's first batch, model, threshold calibration, decode, files) == the files and the
    counts of the unmodified reference binary on the same three files."""
    if not os.path.exists(os.path.join(RBIN, "tagdust_rtest")):
        pytest.skip("oracle/_ref binaries not built")
    import bench
    from tagdust_amd import TagdustHip
    from tagdust_amd import lib as tdlib
    d = str(tmp_path)
    r1, r2, r3, segs = bench.write_casava_files(d, n)
    g = load_golden("casava_index")
    args = str(g["cmdline"]).split()
    subprocess.run([os.path.join(RBIN, "tagdust_rtest")] + args + ["-t", "16", r2, r1, r3, "-o", os.path.join(d, "cpu")], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    # the reference takes its sequence statistics from the first batch it reads: 1000 records in the -DRTEST build (barcode_hmm.c:159-166)
    head = b"".join(open(r2, "rb").readlines()[:4000])
    pr = tdlib.ParsedReads(head, 1)
    ctxs = [TagdustHip(0) for _ in range(n_ctx)]
    try:
        thr = tdlib.estimate_threshold(ctxs[0], segs, pr.codes, pr.offs, float(g["d"]), seed=42, n_reads=4000, rng=1)
        model, _ = tdlib.build_model(segs, pr.codes, pr.offs, 0.05, float(g["d"]))
        for c in ctxs:
            c.set_option("poison_workspace", 1)
            c.upload_model(model)
            c.set_params(thr, 16, 100)
        st, cnt = tdlib.stream_run_multi([(r2, segs, ctxs), (r1, ["R:N"], None), (r3, ["R:N"], None)], os.path.join(d, "gpu"),
                                         n_devices=n_ctx, dust=100, batch_reads=1 << 18 if n_ctx == 1 else 9999, n_threads=4)
    finally:
        pr.close()
        for c in ctxs:
            c.close()
    assert st["n_reads"] == n
    a, b = _files(d, "cpu"), _files(d, "gpu")
    assert len(a) == 26 and set(a) == set(b), (sorted(a), sorted(b))
    for k in a:
        assert a[k] == b[k], "output file *%s differs" % k
    log = _thr_lines(os.path.join(d, "cpu_logfile.txt"))
    assert int(log["successfully extracted"]) == cnt[0] and int(log["problems with architecture"]) == cnt[1]
    assert int(log["low complexity"]) == cnt[6] and int(log["total input reads"]) == n == int(cnt[:8].sum())


# ---------------------------------------------------------------------------------------------------------------------
# SAM / BAM input: io_handler() pipes the file through `samtools view` (io.c:467-575) and read_sam_chunk() takes QNAME, SEQ and
# QUAL of every line (io.c:1498-1660).  The image has no samtools; a stand-in on PATH that does what `view [-S] -F <mask> <file|->`
# does with alignment TEXT (drop header lines and records with a masked flag bit) serves both td_stream_run and the reference
# binary, which start it the same way.
_SAMTOOLS = '''#!/usr/bin/env python3
import sys
a = sys.argv[1:]
assert a[0] == "view", a
mask, path = 0, None
i = 1
while i < len(a):
    if a[i] in ("-SF", "-F"):
        mask = int(a[i + 1]); i += 2
    else:
        path = a[i]; i += 1
src = sys.stdin.buffer if path == "-" else open(path, "rb")
out = sys.stdout.buffer
for line in src:
    if line.startswith(b"@"):
        continue
    f = line.split(b"\\t")
    if int(f[1]) & mask:
        continue
    out.write(line)
'''


def _sam_env(tmp_path, monkeypatch):
    d = tmp_path / "bin"
    d.mkdir()
    p = d / "samtools"
    p.write_text(_SAMTOOLS)
    p.chmod(0o755)
    monkeypatch.setenv("PATH", str(d) + os.pathsep + os.environ["PATH"])
    return str(d)


def _sam_text(names, seqs, quals, seed=0, drop_every=0):
    """Alignment text for the given reads: a header, every record with 11 mandatory fields and some optional ones; with
    drop_every, an extra secondary (256) or QC-failed (512) record in front of every drop_every-th read -- `view -F 768` drops them."""
    rng = np.random.RandomState(seed)
    out = ["@HD\tVN:1.6\tSO:unsorted\n", "@SQ\tSN:chr1\tLN:1000000\n", "@PG\tID:x\tCL:a b c\n"]
    for i, (n, s, q) in enumerate(zip(names, seqs, quals)):
        if drop_every and i % drop_every == 0:
            out.append("%s\t%d\tchr1\t%d\t0\t%dM\t*\t0\t0\t%s\t%s\n" % (n + "_x", 256 if i % 2 else 512, 5 + i, len(s), s[::-1], q))
        flag = int(rng.choice([0, 4, 16]))
        opt = "\tNM:i:%d\tXX:Z:some text" % (i % 3) if i % 4 else ""
        out.append("%s\t%d\t%s\t%d\t%d\t%dM\t*\t0\t0\t%s\t%s%s\n" % (n, flag, "*" if flag == 4 else "chr1", 1 + i, int(rng.randint(0, 60)), len(s), s, q, opt))
    return "".join(out).encode()


def _plain_reads(n, seed):
    rng = np.random.RandomState(seed)
    names, seqs, quals = [], [], []
    for i in range(n):
        L = int(rng.randint(1, 120))
        names.append("read%d/1" % i)
        seqs.append("".join(rng.choice(list("ACGTNacgt"), L, p=[.23, .23, .23, .23, .04, .01, .01, .01, .01])))
        quals.append("".join(chr(c) for c in rng.randint(33, 74, L)))
    return names, seqs, quals


@pytest.mark.parametrize("kind", ["sam", "bam", "sam.gz"])
def test_sam_input_goes_through_samtools_view(tmp_path, monkeypatch, kind):
    """Parse stage alone (no GPU): the records of alignment text are those of the FASTQ text with the same names, bases and
    qualities; header lines, secondary and QC-failed records do not count; blocks cut lines anywhere."""
    from tagdust_amd import lib as tdlib
    _sam_env(tmp_path, monkeypatch)
    names, seqs, quals = _plain_reads(3000, 21)
    fq = "".join("@%s\n%s\n+\n%s\n" % t for t in zip(names, seqs, quals)).encode()
    pr = tdlib.ParsedReads(fq, 1)
    want = (pr.n, _fnv(pr.codes, pr.offs))
    pr.close()
    sam = _sam_text(names, seqs, quals, 3, drop_every=17)
    path = str(tmp_path / ("in." + kind))
    if kind.endswith(".gz"):
        with gzip.open(path, "wb") as fh:
            fh.write(sam)
    else:
        open(path, "wb").write(sam)      # (a .bam here is alignment text as well: the stand-in does not decode BGZF)
    for block in (4096, 1 << 20):
        st = tdlib.stream_run(None, path, batch_reads=700, n_threads=2, block_bytes=block)
        assert (st["n_reads"], st["codes_fnv"]) == want and st["n_batches"] == 5


def test_sam_input_without_samtools_is_an_error(tmp_path, monkeypatch):
    """No samtools on PATH: the shell's status comes back through pclose and ends the run with a message that names the
    command (the reference exits on the empty stream it then reads)."""
    from tagdust_amd import lib as tdlib
    names, seqs, quals = _plain_reads(10, 2)
    path = str(tmp_path / "in.sam")
    open(path, "wb").write(_sam_text(names, seqs, quals))
    empty = tmp_path / "nobin"
    empty.mkdir()
    for tool in ("sh", "zcat", "cat"):
        for d in os.environ["PATH"].split(os.pathsep):
            if os.path.exists(os.path.join(d, tool)):
                os.symlink(os.path.join(d, tool), str(empty / tool))
                break
    monkeypatch.setenv("PATH", str(empty))
    with pytest.raises(Exception, match="samtools"):
        tdlib.stream_run(None, path, batch_reads=100, n_threads=1, block_bytes=4096)


def test_malformed_sam_lines_are_errors(tmp_path, monkeypatch):
    from tagdust_amd import lib as tdlib
    _sam_env(tmp_path, monkeypatch)
    path = str(tmp_path / "in.sam")
    open(path, "wb").write(b"r1\t0\tchr1\t1\t0\t4M\t*\t0\t0\tACGT\n")
    with pytest.raises(Exception, match="fewer than 11 fields"):
        tdlib.stream_run(None, path, batch_reads=100, n_threads=1, block_bytes=4096)
    open(path, "wb").write(b"r1\t0\tchr1\t1\t0\t4M\t*\t0\t0\tACGT\t*\n")
    with pytest.raises(Exception, match="one quality per base"):
        tdlib.stream_run(None, path, batch_reads=100, n_threads=1, block_bytes=4096)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(RBIN, "tagdust_rtest")), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("name", ["c3_b6_s_r_p", "c2_b4_r"])
def test_sam_stream_equals_the_reference_binary(tmp_path, monkeypatch, name):
    """The fixture's reads as alignment text (with header lines and records `view -F 768` drops), three times over, through the
    reference binary and through td_stream_run with the reference's batch size: the same files.  At most 900 records: the
    reference's read_sam_chunk returns the record count when a batch fills (io.c:1652-1655) and its callers take any non-zero
    status for an error, so the -DRTEST binary (batches of 1000) exits with status 1000 & 255 on anything longer; td_stream_run
    has no such limit (the parse-stage test above runs five batches)."""
    from tagdust_amd import TagdustHip
    from tagdust_amd import lib as tdlib
    from test_dropin_gpu import _write_fastq
    _sam_env(tmp_path, monkeypatch)
    g = load_golden(name)
    fq = str(tmp_path / "one.fq")
    _write_fastq(g, fq)
    lines = open(fq, "rb").read().decode().split("\n")
    names = [l[1:].split()[0] for l in lines[0::4] if l]
    seqs = [l for l in lines[1::4]][:len(names)]
    quals = [l for l in lines[3::4]][:len(names)]
    keep = [i for i in range(len(names)) if len(seqs[i]) > 0]
    names, seqs, quals = [([v[i] for i in keep] * 3)[:900] for v in (names, seqs, quals)]
    names = ["%s_%d" % (n, i) for i, n in enumerate(names)]
    sam = str(tmp_path / "in.sam")
    open(sam, "wb").write(_sam_text(names, seqs, quals, 5, drop_every=9))
    args = str(g["cmdline"]).split()
    p = subprocess.run([os.path.join(RBIN, "tagdust_rtest")] + args + ["-t", "2", sam, "-o", "cpu"], cwd=str(tmp_path),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-1500:]
    segs = _segments(g)
    text = "".join("@%s\n%s\n+\n%s\n" % t for t in zip(names, seqs, quals)).encode()
    head = tdlib.ParsedReads(text, 1)
    hc, ho = head.codes.copy(), head.offs.copy()
    head.close()
    c = TagdustHip(0)
    try:
        thr = tdlib.estimate_threshold(c, segs, hc, ho, float(g["d"]), seed=42, n_reads=4000, rng=1)
        model, _ = tdlib.build_model(segs, hc, ho, 0.05, float(g["d"]))
        c.upload_model(model)
        c.set_params(thr, int(g["minlen"]), int(g["dust"]))
        st = tdlib.stream_run(c, sam, segs, str(tmp_path / "gpu"), batch_reads=1000, n_threads=2, block_bytes=30000)
    finally:
        c.close()
    assert st["n_reads"] == len(names)
    a, b = _files(str(tmp_path), "cpu"), _files(str(tmp_path), "gpu")
    assert a and set(a) == set(b)
    for k in a:
        assert a[k] == b[k], "output file *%s differs" % k


@pytest.mark.gpu
def test_a_full_disk_fails_the_run(tmp_path):
    """Every output file a link to /dev/full: the appender thread's pwrite comes back with ENOSPC while the next batch is being
    formatted; the run must end with an error that says so (not hang, not report success)."""
    from test_dropin_gpu import _write_fastq
    from tagdust_amd import TagdustHip, TdError
    from tagdust_amd import lib as tdlib
    g = load_golden("c2_b4_r")
    fq = str(tmp_path / "in.fq")
    _write_fastq(g, fq)
    segs = _segments(g)
    c = TagdustHip(0)
    try:
        c.upload_model(g)
        c.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
        st = tdlib.stream_run(c, fq, segs, str(tmp_path / "out"), batch_reads=50, n_threads=2, block_bytes=4096)
        assert st["n_reads"] == int(g["n_reads"])
        made = sorted(glob.glob(str(tmp_path / "out*.fq")))
        assert made
        for p in made:
            os.remove(p)
            os.symlink("/dev/full", p)
        with pytest.raises(TdError, match="write failed"):
            tdlib.stream_run(c, fq, segs, str(tmp_path / "out"), batch_reads=50, n_threads=2, block_bytes=4096)
    finally:
        c.close()
