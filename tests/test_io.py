"""FASTQ ingest / demultiplexed FASTQ egress of the library (include/tagdust_io.h) against the reference binary:
own parser + decode + own writer must produce the files the unmodified reference CLI writes, byte for byte.
The decode step is the oracle here (CPU); tests/test_dropin_gpu.py and test_io_gpu below use the HIP path."""
import glob
import os
import subprocess

import numpy as np
import pytest

from conftest import load_golden, REPO
from tagdust_amd import lib as tdlib

RBIN = os.path.join(REPO, "oracle", "_ref")
HAVE_REF = os.path.exists(os.path.join(RBIN, "tagdust_rtest"))
NAMES = ["c2_b4_r", "c3_b6_s_r_p", "scen2_endloss", "umi_f_s_r", "short_q_given", "casava_index", "b_r_s_r", "dust_b_r"]


def fastq_text(g):
    names = bytes(g["names"]).split(b"\n")
    offs = g["offs"]
    out = []
    for i in range(int(g["n_reads"])):
        s = bytes(np.frombuffer(b"ACGTN", np.uint8)[g["seq"][offs[i]:offs[i + 1]]])
        out.append(b"@" + names[i] + b"\n" + s + b"\n+\n" + bytes(g["qual"][offs[i]:offs[i + 1]]) + b"\n")
    return b"".join(out)


def segments_of(g):
    segs = []
    for t, grp in zip(g["seg_type"], str(g["seg_seqs"]).split(";")):
        t = chr(int(t))
        seqs = grp.split(",")
        if t in "BS":
            seqs = seqs[:-1]
        segs.append("%s:%s" % (t, ",".join(seqs)))
    return segs


def reference_outputs(g, tmp):
    fq = os.path.join(tmp, "in.fq")
    open(fq, "wb").write(fastq_text(g))
    p = subprocess.run([os.path.join(RBIN, "tagdust_rtest")] + str(g["cmdline"]).split() + [fq, "-o", "ref"], cwd=tmp,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-1500:]
    return {os.path.basename(f)[3:]: open(f, "rb").read() for f in glob.glob(os.path.join(tmp, "ref*.fq"))}


@pytest.mark.parametrize("name", NAMES)
def test_parser_matches_reference_loader(name):
    g = load_golden(name)
    for threads in (1, 4):
        pr = tdlib.ParsedReads(fastq_text(g), threads)
        assert pr.n == int(g["n_reads"])
        assert np.array_equal(pr.offs, g["offs"]) and np.array_equal(pr.codes, g["seq"])
        assert pr.names() == bytes(g["names"]).split(b"\n")
        pr.close()


def test_parser_chunked_large_input_and_fasta():
    rng = np.random.RandomState(5)
    recs, want = [], []
    for i in range(60000):
        s = bytes(np.frombuffer(b"ACGTNacgtu.", np.uint8)[rng.randint(0, 11, rng.randint(30, 120))])
        q = bytes(rng.randint(33, 74, len(s)).astype(np.uint8))   # quality lines may start with '@' or '+'
        recs.append(b"@r%d some text\tafter tab\n" % i + s + b"\n+\n" + q + b"\n")
        want.append(s)
    text = b"".join(recs)
    a, b = tdlib.ParsedReads(text, 1), tdlib.ParsedReads(text, 8)
    assert a.n == b.n == 60000
    assert np.array_equal(a.offs, b.offs) and np.array_equal(a.codes, b.codes) and np.array_equal(a.qual_off, b.qual_off)
    assert a.names()[7] == b"r7 some text"
    lut = np.full(256, 4, np.uint8)
    for ch, c in zip(b"ACGTUacgtu.", [0, 1, 2, 3, 3, 0, 1, 2, 3, 3, 5]):
        lut[ch] = c
    assert np.array_equal(a.codes[a.offs[123]:a.offs[124]], lut[np.frombuffer(want[123], np.uint8)])
    fa = tdlib.ParsedReads(b">x1\nACGT\n>x2 y\nGGNA\n", 1)
    assert fa.n == 2 and list(fa.codes) == [0, 1, 2, 3, 2, 2, 4, 0] and list(fa.qual_off) == [-1, -1]


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/tagdust_rtest not built")
@pytest.mark.parametrize("name", NAMES)
def test_writer_reproduces_reference_files(tmp_path, name):
    from oracle import pyoracle
    g = load_golden(name)
    want = reference_outputs(g, str(tmp_path))
    pr = tdlib.ParsedReads(fastq_text(g), 2)
    ores, _, oseq = pyoracle.label_batch(pyoracle.OracleModel(g), pr.codes, pr.offs, float(g["threshold"]), int(g["minlen"]), int(g["dust"]), 2)
    res = np.zeros(pr.n, tdlib.RESULT_DTYPE)
    for k in ("f_score", "b_score", "r_score", "bar_prob", "read_type", "barcode", "fingerprint"):
        res[k] = ores[k]
    res["mapq"] = ores["Q"]
    tdlib.write_demultiplexed(str(tmp_path / "own"), segments_of(g), pr, res, oseq)
    got = {os.path.basename(f)[3:]: open(f, "rb").read() for f in glob.glob(str(tmp_path / "own*.fq"))}
    # the reference only creates... every file of the set, possibly empty; so does the writer
    assert set(got) == set(want), (sorted(got), sorted(want))
    for k in want:
        assert got[k] == want[k], k


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/tagdust_rtest not built")
@pytest.mark.parametrize("name", ["c3_b6_s_r_p", "scen2_endloss"])
def test_io_gpu(tmp_path, name):
    """parse -> HIP decode -> write == the reference CLI's files."""
    from tagdust_amd import TagdustHip
    g = load_golden(name)
    want = reference_outputs(g, str(tmp_path))
    pr = tdlib.ParsedReads(fastq_text(g), 2)
    c = TagdustHip(0)
    try:
        c.upload_model(g)
        c.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
        c.upload_batch(pr.codes, pr.offs)
        c.run()
        res, _, seq_out = c.download(labels=False)
    finally:
        c.close()
    tdlib.write_demultiplexed(str(tmp_path / "own"), segments_of(g), pr, res, seq_out)
    got = {os.path.basename(f)[3:]: open(f, "rb").read() for f in glob.glob(str(tmp_path / "own*.fq"))}
    assert set(got) == set(want)
    for k in want:
        assert got[k] == want[k], k


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/tagdust_rtest not built")
@pytest.mark.parametrize("name", ["c2_b4_r", "c3_b6_s_r_p", "scen2_p_b_r_p"])
def test_whole_pipeline_without_the_reference(tmp_path, name):
    """Everything from the library alone -- parse FASTQ, sequence statistics, model construction, threshold calibration
    (emission on the host, scoring on the GPU), decoding, demultiplexed output -- must write the files the unmodified
    reference CLI writes for the same command line."""
    from tagdust_amd import TagdustHip
    g = load_golden(name)
    want = reference_outputs(g, str(tmp_path))
    segs = segments_of(g)
    pr = tdlib.ParsedReads(fastq_text(g), 2)
    c = TagdustHip(0)
    try:
        thr = tdlib.estimate_threshold(c, segs, pr.codes, pr.offs, float(g["d"]), seed=42, n_reads=4000, rng=1)
        model, _ = tdlib.build_model(segs, pr.codes, pr.offs, 0.05, float(g["d"]))   # calibration leaves e = 0.05 (calibrateQ.c:117)
        c.upload_model(model)
        c.set_params(thr, 16, 100)
        c.upload_batch(pr.codes, pr.offs)
        c.run()
        res, _, seq_out = c.download(labels=False)
    finally:
        c.close()
    tdlib.write_demultiplexed(str(tmp_path / "own"), segs, pr, res, seq_out)
    got = {os.path.basename(f)[3:]: open(f, "rb").read() for f in glob.glob(str(tmp_path / "own*.fq"))}
    assert set(got) == set(want)
    for k in want:
        assert got[k] == want[k], k


def test_quality_line_must_match_the_sequence_length():
    """A FASTQ record whose quality line is shorter than its sequence (e.g. a file cut off mid-record) is an error, as in the
    reference (io.c:1776-1781, "Length of sequence and base qualities differ") -- never a writer reading past the line."""
    good = b"@r1\nACGTACGT\n+\nIIIIIIII\n@r2\nACGT\n+\nIIII\n"
    pr = tdlib.ParsedReads(good, 1)
    assert pr.n == 2
    pr.close()
    for bad in (b"@r1\nACGTACGT\n+\nIIIIIIII\n@r2\nACGTACGT\n+\nIII",      # cut off inside the last quality line
                b"@r1\nACGTACGT\n+\nIIII\n@r2\nACGT\n+\nIIII\n",          # short quality in the middle
                b"@r1\nACGT\n+\nIIIIII\n"):                                  # quality longer than the sequence
        with pytest.raises(tdlib.TdError, match="base qualities"):
            tdlib.ParsedReads(bad, 1)
    # a large multi-threaded parse with one bad record somewhere
    rec = b"@x\n" + b"ACGT" * 25 + b"\n+\n" + b"I" * 100 + b"\n"
    text = rec * 30000 + b"@bad\nACGTAC\n+\nIII\n" + rec * 30000
    with pytest.raises(tdlib.TdError, match="bad"):
        tdlib.ParsedReads(text, 4)


def test_fasta_parse_matches_reference_read_fasta():
    """td_fasta_parse against what the reference's get_fasta()/read_fasta() made of the same file (fixture
    artifacts_b_r: mixed case, CRLF line ends, a blank in the header)."""
    from tagdust_amd import lib as tdlib
    g = load_golden("artifacts_b_r")
    st, ix, names = tdlib.parse_fasta(bytes(g["art_fasta_text"]))
    assert np.array_equal(ix, g["art_index"])
    assert np.array_equal(st, g["art_string"])
    assert names == [b"artifact_1", b"artifact_2", b"artifact_3"]   # white space -> '_' (io.c:1973-1977)
    st2, ix2, names2 = tdlib.parse_fasta(b"")
    assert len(ix2) == 1 and ix2[0] == 0 and names2 == []
    st3, ix3, _ = tdlib.parse_fasta(b">only_a_header\n")
    assert ix3.tolist() == [0, 1] and st3.tolist() == [ord("X")]


def test_base_coding_of_every_byte_value_and_length():
    """td_encode_bases (sixteen letters at a time, an overlapping last block) against init_nuc_code's table (nuc_code.c:46-74) spelled
    out here: every byte value a sequence line can hold (32..255 without 127), at every length 1..70 and every phase within a block."""
    from tagdust_amd import lib as tdlib
    table = np.full(256, 4, np.uint8)
    for ch, v in (("A", 0), ("a", 0), ("C", 1), ("c", 1), ("G", 2), ("g", 2), ("T", 3), ("t", 3), ("U", 3), ("u", 3), (".", 5)):
        table[ord(ch)] = v
    allowed = np.array([b for b in range(32, 256) if b != 127], np.uint8)
    rng = np.random.RandomState(4)
    recs, want = [], []
    for L in range(1, 71):
        for rep in range(6):
            s = allowed[(np.arange(L) * (rep + 1) + rng.randint(0, len(allowed))) % len(allowed)] if rep < 3 else rng.choice(allowed, L)
            if s[0] in (ord("@"), ord("+"), ord(">")):
                s[0] = ord("A")                     # (a sequence line does not start a record or a separator)
            recs.append(b"@r%d_%d\n" % (L, rep) + s.tobytes() + b"\n+\n" + b"I" * L + b"\n")
            want.append(table[s])
    text = b"".join(recs)
    for threads in (1, 3):
        pr = tdlib.ParsedReads(text, threads)
        assert pr.n == len(want)
        for i, w in enumerate(want):
            got = pr.codes[pr.offs[i]:pr.offs[i + 1]]
            assert np.array_equal(got, w), (i, got[:20], w[:20])
        pr.close()
