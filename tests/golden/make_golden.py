#!/usr/bin/env python3
"""Generate the committed parity fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference and oracle/_ref built by
`make -C oracle ref`).  For every scenario it
  1. produces an input FASTQ (reference `simreads`, the build's own seeded generator for
     architectures simreads cannot emit, or a slice of a reference test data file),
  2. runs oracle/_ref/ref_dump_rtest (own harness linked against the unmodified reference
     sources; -DRTEST makes the threshold calibration small and libc-independent) on it,
  3. converts the binary dump into tests/golden/<name>.npz.

A fixture is data: inputs, the model tables init_model_bag() built, and the per-read outputs
of backward()/forward_max_posterior_decoding()/run_pHMM(MODE_GET_LABEL).  No reference source
text is stored.

usage: python tests/golden/make_golden.py [scenario ...]
"""
import gzip
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
RBIN = os.path.join(REPO, "oracle", "_ref")

ADAPTER = "AGATCGGAAGAGC"


def read_tags(path, n):
    tags = []
    for line in open(path):
        if ":" in line and not line.startswith("["):
            tags.append(line.strip().split(":")[1])
    return tags[:n]


# ---------------------------------------------------------------------------------------------
# binary dump reader (format written by oracle/ref_dump.c, version 2)
# ---------------------------------------------------------------------------------------------
class _Rd:
    def __init__(self, buf):
        self.b, self.o = buf, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def arr(self, dtype, n):
        a = np.frombuffer(self.b, dtype=dtype, count=n, offset=self.o).copy()
        self.o += a.nbytes
        return a


def parse_dump(path):
    r = _Rd(open(path, "rb").read())
    assert r.arr("S4", 1)[0] == b"TDRF"
    assert r.take("i") == 2
    d = {}
    d["e"], d["d"], d["threshold"] = r.arr("<f4", 3)
    d["q_given"], d["minlen"], d["dust"], d["matchstart"], d["matchend"] = r.arr("<i4", 5)
    d["ssi_background"] = r.arr("<f8", 5)
    (d["ssi_expected_5_len"], d["ssi_expected_3_len"], d["ssi_mean_5_len"], d["ssi_stdev_5_len"],
     d["ssi_mean_3_len"], d["ssi_stdev_3_len"], d["ssi_average_length"]) = r.arr("<f8", 7)
    d["ssi_max_seq_len"] = r.take("i")
    nseg = r.take("i")
    seg_type, seg_nseq, seg_len, seg_seqs = [], [], [], []
    for _ in range(nseg):
        t, n, sl = r.arr("<i4", 3)
        seg_type.append(t); seg_nseq.append(n); seg_len.append(sl)
        seg_seqs.append([r.arr("S%d" % sl, 1)[0].decode() for _ in range(n)])
    d["seg_type"] = np.array(seg_type, np.int32)
    d["seg_nseq"] = np.array(seg_nseq, np.int32)
    d["seg_len"] = np.array(seg_len, np.int32)
    d["seg_seqs"] = np.array(";".join(",".join(s) for s in seg_seqs))
    S, H, C, avg = r.arr("<i4", 4)
    d["S"], d["H"], d["C"], d["avg_len"] = S, H, C, avg
    d["bg"] = r.arr("<f4", 5)
    n_hmm, n_col, skip = [], [], []
    for _ in range(S):
        a, b = r.arr("<i4", 2)
        n_hmm.append(a); n_col.append(b); skip.append(r.arr("<f4", 1)[0])
    d["n_hmm"] = np.array(n_hmm, np.int32)
    d["n_col"] = np.array(n_col, np.int32)
    d["skip"] = np.array(skip, np.float32)
    cols = r.arr("<f4", C * 21).reshape(C, 21)
    d["trans"], d["eM"], d["eI"] = cols[:, 0:9].copy(), cols[:, 9:14].copy(), cols[:, 14:19].copy()
    d["sM"], d["sI"] = cols[:, 19].copy(), cols[:, 20].copy()
    d["label"] = r.arr("<i4", H)
    d["A"] = r.arr("<f4", H * H).reshape(H, H)
    n = r.take("i")
    lens = np.zeros(n, np.int32)
    names, seq, qual, labels, seq_after = [], [], [], [], []
    sc = np.zeros((n, 3), np.float32)
    bar = np.zeros(n, np.float64)
    mapq = np.zeros(n, np.float32)
    ints = np.zeros((n, 3), np.int32)
    for i in range(n):
        ln, nl = r.arr("<i4", 2)
        lens[i] = ln
        names.append(r.arr("S%d" % nl, 1)[0] if nl else b"")
        seq.append(r.arr("u1", ln)); qual.append(r.arr("u1", ln))
        sc[i] = r.arr("<f4", 3)
        bar[i] = r.arr("<f8", 1)[0]
        labels.append(r.arr("i1", ln + 1))
        mapq[i] = r.arr("<f4", 1)[0]
        ints[i] = r.arr("<i4", 3)
        seq_after.append(r.arr("u1", ln))
    if r.o < len(r.b):  # optional trailer: -ref artifact sequences as read_fasta() left them
        assert r.arr("S4", 1)[0] == b"ARTF"
        n_art, fe, nt = r.arr("<i4", 3)
        d["art_n"], d["art_filter_error"], d["art_threads"] = n_art, fe, nt
        d["art_index"] = r.arr("<i4", n_art + 1)
        d["art_string"] = r.arr("u1", int(d["art_index"][-1]))
    assert r.o == len(r.b), (r.o, len(r.b))
    d["n_reads"] = n
    d["lens"] = lens
    d["offs"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    d["names"] = np.array(b"\n".join(names))
    d["seq"] = np.concatenate(seq) if n else np.zeros(0, np.uint8)
    d["qual"] = np.concatenate(qual) if n else np.zeros(0, np.uint8)
    d["labels"] = np.concatenate(labels) if n else np.zeros(0, np.int8)   # read i at offs[i]+i, len+1 entries
    d["seq_after"] = np.concatenate(seq_after) if n else np.zeros(0, np.uint8)
    d["b_score"], d["f_score"], d["r_score"] = sc[:, 0].copy(), sc[:, 1].copy(), sc[:, 2].copy()
    d["bar_prob"] = bar
    d["mapq"] = mapq
    d["read_type"], d["barcode"], d["fingerprint"] = ints[:, 0].copy(), ints[:, 1].copy(), ints[:, 2].copy()
    return d


# ---------------------------------------------------------------------------------------------
# input generators
# ---------------------------------------------------------------------------------------------
def run(cmd, **kw):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, **kw)
    if p.returncode != 0:
        sys.stderr.write(p.stdout.decode(errors="replace"))
        raise SystemExit("command failed: %s" % " ".join(cmd))
    return p.stdout.decode(errors="replace")


def simreads(out, tagfile, extra, rtest=False):
    exe = os.path.join(RBIN, "simreads_rtest" if rtest else "simreads")
    run([exe, tagfile] + extra + ["-o", out])
    return open(out + "_tagdust_arch.txt").read().split()


def mutate(rng, s, sub, indel):
    out = []
    for ch in s:
        u = rng.random_sample()
        if u < sub:
            out.append("ACGT"[rng.randint(4)])
        elif u < sub + indel / 2:
            continue
        elif u < sub + indel:
            out.append(ch)
            out.append("ACGT"[rng.randint(4)])
        else:
            out.append(ch)
    return "".join(out)


def synth_reads(path, n, total_len, barcodes, umi_len, spacer, adapter, seed,
                random_frac=0.1, sub=0.02, indel=0.0, n_frac=0.0, var_len=None):
    """Own generator for architectures simreads cannot emit (S: / F: segments):
    [barcode][UMI][spacer] mutated + uniform insert + mutated adapter, fixed total length."""
    rng = np.random.RandomState(seed)
    with open(path, "w") as fh:
        for i in range(n):
            L = total_len if var_len is None else int(rng.randint(var_len[0], var_len[1] + 1))
            if rng.random_sample() < random_frac:
                s = "".join("ACGT"[k] for k in rng.randint(0, 4, L))
                name = "READ%d;RANDOM" % i
            else:
                b = rng.randint(len(barcodes)) if barcodes else -1
                head = (barcodes[b] if barcodes else "") + \
                    "".join("ACGT"[k] for k in rng.randint(0, 4, umi_len)) + spacer
                head = mutate(rng, head, sub, indel)
                tail = mutate(rng, adapter, sub, indel)
                keep = len(tail) if not adapter else int(rng.randint(0, len(tail) + 1))
                tail = tail[:keep]
                ins = max(L - len(head) - len(tail), 0)
                s = (head + "".join("ACGT"[k] for k in rng.randint(0, 4, ins)) + tail)[:L]
                name = "READ%d;BARNUM:%d" % (i, b + 1)
            if n_frac:
                s = "".join("N" if rng.random_sample() < n_frac else ch for ch in s)
            fh.write("@%s\n%s\n+\n%s\n" % (name, s, "I" * len(s)))


# ---------------------------------------------------------------------------------------------
# scenarios
# ---------------------------------------------------------------------------------------------
EXTRA = {}   # per-scenario input data that is not part of the dump


def scenarios(tmp):
    dev = os.path.join(REF, "dev")
    sc = {}

    def c2():  # BASELINE config-2 shape
        fq = os.path.join(tmp, "c2.fq")
        arch = simreads(fq, os.path.join(dev, "EDITTAG_4nt_ed_2.txt"),
                        "-seed 42 -sim_barnum 8 -sim_readlen 96 -sim_readlen_mod 0 -sim_numseq 360 -sim_endloss 0 "
                        "-sim_random_frac 0.1 -sim_error_rate 0.02".split())
        return fq, ["-seed", "42"] + arch[1:arch.index("in.fq")]
    sc["c2_b4_r"] = c2

    def c2_indel():  # same arch, indels + end loss + variable read length
        fq = os.path.join(tmp, "c2i.fq")
        arch = simreads(fq, os.path.join(dev, "EDITTAG_4nt_ed_2.txt"),
                        "-seed 7 -sim_barnum 8 -sim_readlen 60 -sim_readlen_mod 20 -sim_numseq 300 -sim_endloss 2 "
                        "-sim_random_frac 0.15 -sim_error_rate 0.05 -sim_InDel_frac 0.3".split())
        return fq, ["-seed", "42"] + arch[1:arch.index("in.fq")]
    sc["c2_indel_varlen"] = c2_indel

    def c3():  # BASELINE config-3 shape (own generator: simreads cannot emit S:)
        fq = os.path.join(tmp, "c3.fq")
        bars = read_tags(os.path.join(dev, "EDITTAG_6nt_ed_3.txt"), 8)
        synth_reads(fq, 260, 150, bars, 0, "GTA", ADAPTER, seed=3, indel=0.01, n_frac=0.002)
        return fq, ["-seed", "42", "-1", "B:" + ",".join(bars), "-2", "S:GTA", "-3", "R:N", "-4", "P:" + ADAPTER]
    sc["c3_b6_s_r_p"] = c3

    def c5():  # BASELINE config-5 shape, H = 100 (the total_prob[100] limit)
        fq = os.path.join(tmp, "c5.fq")
        bars = read_tags(os.path.join(dev, "EDITTAG_6nt_ed_3.txt"), 96)
        synth_reads(fq, 48, 150, bars, 8, "", ADAPTER, seed=5)
        return fq, ["-seed", "42", "-1", "B:" + ",".join(bars), "-2", "F:NNNNNNNN", "-3", "R:N", "-4", "P:" + ADAPTER]
    sc["c5_b96_f_r_p"] = c5


    def c5_big():  # config-5 architecture (H = 100) on >= 1000 reads covering every outcome of the label path
        fq = os.path.join(tmp, "c5big.fq")
        bars = read_tags(os.path.join(dev, "EDITTAG_6nt_ed_3.txt"), 96)
        rng = np.random.RandomState(55)

        def dna(n):
            return "".join("ACGT"[k] for k in rng.randint(0, 4, n))
        with open(fq, "w") as fh:
            for i in range(1100):
                kind = rng.choice(8, p=[0.52, 0.08, 0.08, 0.08, 0.08, 0.08, 0.04, 0.04])
                L = 150 if rng.random_sample() < 0.7 else int(rng.randint(40, 150))
                b = bars[rng.randint(96)]
                umi = dna(8)
                ad = ADAPTER[:int(rng.randint(0, len(ADAPTER) + 1))]
                if kind == 1:      # UMI of the wrong length: fingerprint not found
                    umi = dna(int(rng.choice([5, 6, 7, 9, 10, 11])))
                elif kind == 2:    # a 6-mer that is no barcode: decoy HMM
                    b = dna(6)
                if kind == 3:      # insert shorter than -minlen in front of the whole adapter
                    s_ = b + umi + dna(int(rng.randint(2, 15))) + ADAPTER
                elif kind == 4:    # low-complexity insert: DUST
                    u = dna(int(rng.randint(1, 4)))
                    s_ = b + umi + (u * 150)[:max(L - 14 - len(ad), 20)] + ad
                elif kind == 5:    # unrelated sequence
                    s_ = dna(L)
                elif kind == 6:    # barcode with two substitutions / an indel
                    bb = list(b)
                    bb[rng.randint(6)] = "ACGT"[rng.randint(4)]
                    bb[rng.randint(6)] = "ACGT"[rng.randint(4)]
                    if rng.random_sample() < 0.5:
                        del bb[rng.randint(len(bb))]
                    s_ = "".join(bb) + umi + dna(max(L - 14 - len(ad), 20)) + ad
                else:
                    s_ = mutate(rng, b + umi, 0.02, 0.01) + dna(max(L - 14 - len(ad), 20)) + mutate(rng, ad, 0.02, 0.01)
                if kind == 7:      # N bases anywhere
                    s_ = "".join("N" if rng.random_sample() < 0.03 else ch for ch in s_)
                fh.write("@READ%d;KIND:%d\n%s\n+\n%s\n" % (i, kind, s_, "I" * len(s_)))
        return fq, ["-seed", "42", "-1", "B:" + ",".join(bars), "-2", "F:NNNNNNNN", "-3", "R:N", "-4", "P:" + ADAPTER]
    sc["c5_big_b96_f_r_p"] = c5_big

    def scen1():  # dev/bar_read_test.sh scenario 1 (first 400 reads + the random tail)
        fq = os.path.join(tmp, "s1.fq")
        arch = simreads(fq, os.path.join(dev, "EDITTAG_6nt_ed_4.txt"),
                        "-seed 42 -sim_barnum 4 -sim_readlen 20 -sim_readlen_mod 0 -sim_numseq 400 -sim_endloss 0 "
                        "-sim_random_frac 0.1 -sim_error_rate 0.02".split(), rtest=True)
        return fq, ["-seed", "42"] + arch[1:arch.index("in.fq")]
    sc["scen1_b_r"] = scen1

    def scen2():  # dev/bar_read_test.sh scenario 2: 5' and 3' partial linkers
        fq = os.path.join(tmp, "s2.fq")
        arch = simreads(fq, os.path.join(dev, "EDITTAG_6nt_ed_4.txt"),
                        "-seed 42 -sim_barnum 4 -sim_5seq GGGGGGG -sim_3seq TTTTTTT -sim_readlen 20 -sim_readlen_mod 0 "
                        "-sim_numseq 400 -sim_endloss 0 -sim_random_frac 0.1 -sim_error_rate 0.02".split(), rtest=True)
        return fq, ["-seed", "42"] + arch[1:arch.index("in.fq")]
    sc["scen2_p_b_r_p"] = scen2

    def scen2_endloss():  # partial linkers actually partial (end loss), exercises the 5'/3' Gaussians
        fq = os.path.join(tmp, "s2e.fq")
        arch = simreads(fq, os.path.join(dev, "EDITTAG_6nt_ed_4.txt"),
                        "-seed 11 -sim_barnum 4 -sim_5seq GGGGGGGA -sim_3seq TTTTTTTC -sim_readlen 30 -sim_readlen_mod 4 "
                        "-sim_numseq 300 -sim_endloss 4 -sim_random_frac 0.1 -sim_error_rate 0.03 -sim_InDel_frac 0.1".split())
        return fq, ["-seed", "42"] + arch[1:arch.index("in.fq")]
    sc["scen2_endloss"] = scen2_endloss

    def casava():  # BASELINE config-4, file 2: 6-nt index reads from the reference's own test data
        fq = os.path.join(tmp, "casava2.fq")
        with gzip.open(os.path.join(dev, "casava_read2.fastq.gz"), "rt") as fh, open(fq, "w") as out:
            lines = fh.readlines()
            out.writelines(lines[4 * 2000:4 * 2400])  # a slice past the leading all-N reads
        arch = open(os.path.join(REF, "casava_demo", "casava_6nt_arch.txt")).readline().split()
        return fq, ["-seed", "42"] + arch[1:]
    sc["casava_index"] = casava

    def short():  # very short / too-short reads, -Q given (quirk Q1: threshold 0), dust off
        fq = os.path.join(tmp, "short.fq")
        bars = read_tags(os.path.join(dev, "EDITTAG_4nt_ed_2.txt"), 4)
        synth_reads(fq, 200, 0, bars, 0, "", "", seed=9, random_frac=0.2, var_len=(3, 40), n_frac=0.01)
        return fq, ["-Q", "5", "-dust", "0", "-minlen", "12", "-1", "B:" + ",".join(bars), "-2", "R:N"]
    sc["short_q_given"] = short

    def umi_only():  # F + R without barcode, custom -e / -i
        fq = os.path.join(tmp, "umi.fq")
        synth_reads(fq, 200, 60, [], 6, "GG", "", seed=13)
        return fq, ["-seed", "42", "-e", "0.02", "-i", "0.2", "-1", "F:NNNNNN", "-2", "S:GG", "-3", "R:N"]
    sc["umi_f_s_r"] = umi_only

    def opt_g():  # O: and G: segment types (the manual's CAGE example), internal P
        fq = os.path.join(tmp, "og.fq")
        bars = ["TTTAGG", "ATTCCA", "GCTCAA", "CATCCC"]
        synth_reads(fq, 200, 70, bars, 0, "GGG", "", seed=17, indel=0.01)
        return fq, ["-seed", "42", "-1", "O:N", "-2", "B:" + ",".join(bars), "-3", "S:GGG", "-4", "R:N"]
    sc["o_b_s_r"] = opt_g

    def int_p():  # internal P segment + G segment
        fq = os.path.join(tmp, "ip.fq")
        bars = ["ACAGTG", "ACTTGA", "TTAGGC"]
        synth_reads(fq, 160, 80, bars, 0, "CTGCA", "", seed=19)
        return fq, ["-seed", "42", "-1", "B:" + ",".join(bars), "-2", "P:CTGCA", "-3", "G:G", "-4", "R:N"]
    sc["b_intp_g_r"] = int_p

    def two_reads():  # two read segments around a spacer: one input file, READ1 / READ2 output files
        fq = os.path.join(tmp, "rr.fq")
        rng = np.random.RandomState(31)
        bars = ["ACAGTG", "CTTGTA", "GGCTAC"]
        sp = "GATCGGAAGAGC"
        with open(fq, "w") as fh:
            for i in range(220):
                a = "".join("ACGT"[k] for k in rng.randint(0, 4, rng.randint(10, 40)))     # some first reads shorter than -minlen
                b2 = "".join("ACGT"[k] for k in rng.randint(0, 4, rng.randint(14, 45)))
                s_ = bars[rng.randint(3)] + a + mutate(rng, sp, 0.03, 0.02) + b2
                if rng.random_sample() < 0.1:
                    s_ = "".join("ACGT"[k] for k in rng.randint(0, 4, len(s_)))
                fh.write("@READ%d\n%s\n+\n%s\n" % (i, s_, "I" * len(s_)))
        return fq, ["-seed", "42", "-1", "B:" + ",".join(bars), "-2", "R:N", "-3", "S:" + sp, "-4", "R:N"]
    sc["b_r_s_r"] = two_reads

    def dust():  # low-complexity inserts: DUST flags some reads (outcome 6), -dust 30
        fq = os.path.join(tmp, "dust.fq")
        rng = np.random.RandomState(29)
        bars = read_tags(os.path.join(dev, "EDITTAG_4nt_ed_2.txt"), 4)
        with open(fq, "w") as fh:
            for i in range(240):
                L = int(rng.randint(24, 90))
                kind = i % 6
                if kind == 0:
                    ins = "ACGT"[rng.randint(4)] * L
                elif kind == 1:
                    u = "".join("ACGT"[k] for k in rng.randint(0, 4, 2)); ins = (u * L)[:L]
                elif kind == 2:
                    u = "".join("ACGT"[k] for k in rng.randint(0, 4, 3)); ins = (u * L)[:L]
                elif kind == 3:   # random head, repetitive tail beyond the 64 characters DUST looks at
                    ins = "".join("ACGT"[k] for k in rng.randint(0, 4, 66)) + "A" * 20
                elif kind == 4:   # repetitive with a few substitutions
                    ins = mutate(rng, "ACGT"[rng.randint(4)] * L, 0.1, 0.0)
                else:
                    ins = "".join("ACGT"[k] for k in rng.randint(0, 4, L))
                s_ = bars[rng.randint(len(bars))] + ins
                fh.write("@READ%d\n%s\n+\n%s\n" % (i, s_, "I" * len(s_)))
        return fq, ["-seed", "42", "-dust", "30", "-1", "B:" + ",".join(bars), "-2", "R:N"]
    sc["dust_b_r"] = dust

    def window():  # -start / -end: the architecture sits in a window of the read; -Q given (calibration with a window is undefined)
        fq = os.path.join(tmp, "win.fq")
        rng = np.random.RandomState(37)
        bars = read_tags(os.path.join(dev, "EDITTAG_4nt_ed_2.txt"), 6)
        with open(fq, "w") as fh:
            for i in range(260):
                pre = "".join("ACGT"[k] for k in rng.randint(0, 4, 5))
                ins = "".join("ACGT"[k] for k in rng.randint(0, 4, int(rng.randint(30, 52))))
                post = "".join("ACGT"[k] for k in rng.randint(0, 4, int(rng.randint(14, 40))))
                body = mutate(rng, bars[rng.randint(len(bars))], 0.03, 0.02) + ins
                if i % 7 == 0:
                    body = "".join("ACGT"[k] for k in rng.randint(0, 4, len(body)))
                if i % 11 == 0:
                    body = bars[rng.randint(len(bars))] + "A" * 50          # low-complexity insert
                s_ = (pre + body + post)
                s_ = s_ + "".join("ACGT"[k] for k in rng.randint(0, 4, max(0, 62 - len(s_))))
                fh.write("@READ%d\n%s\n+\n%s\n" % (i, s_, "I" * len(s_)))
        return fq, ["-Q", "5", "-start", "5", "-end", "61", "-dust", "30", "-1", "B:" + ",".join(bars), "-2", "R:N"]
    sc["window_b_r"] = window

    def artifacts():  # -ref: artifact matching between extraction and DUST, 3 threads (4-groups + left-over reads)
        fq = os.path.join(tmp, "art.fq")
        fa = os.path.join(tmp, "art.fa")
        rng = np.random.RandomState(23)
        arts = ["".join("ACGT"[k] for k in rng.randint(0, 4, L)) for L in (70, 48, 95)]
        with open(fa, "w", newline="") as fh:
            for k, a in enumerate(arts):
                body = "\n".join(a[x:x + 40] for x in range(0, len(a), 40))
                if k == 1:
                    body = body.lower().replace("\n", "\r\n")      # lower case, CRLF line ends
                fh.write(">artifact %d\n%s\n" % (k + 1, body))
        bars = read_tags(os.path.join(dev, "EDITTAG_4nt_ed_2.txt"), 4)
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        with open(fq, "w") as fh:
            for i in range(203):  # 203 reads over 3 threads: ranges of 67, 67, 69 -> 3 + 3 + 1 left-over reads
                b = bars[rng.randint(len(bars))]
                L = int(rng.randint(36, 60))
                u = rng.random_sample()
                if u < 0.45:      # a window of an artifact, either strand, a few substitutions
                    a = arts[rng.randint(len(arts))]
                    st = int(rng.randint(0, max(len(a) - L, 1)))
                    ins = a[st:st + L]
                    if rng.random_sample() < 0.5:
                        ins = "".join(comp[ch] for ch in reversed(ins))
                    ins = mutate(rng, ins, 0.03 if u < 0.3 else 0.12, 0.02)
                else:
                    ins = "".join("ACGT"[k] for k in rng.randint(0, 4, L))
                s_ = b + ins
                fh.write("@READ%d\n%s\n+\n%s\n" % (i, s_, "I" * len(s_)))
        os.environ["REF_DUMP_THREADS"] = "3"
        EXTRA["artifacts_b_r"] = {"art_fasta_text": np.frombuffer(open(fa, "rb").read(), np.uint8)}   # input data of -ref
        return fq, ["-seed", "42", "-ref", fa, "-fe", "9", "-1", "B:" + ",".join(bars), "-2", "R:N"]
    sc["artifacts_b_r"] = artifacts
    return sc


def main():
    want = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        for name, fn in scenarios(tmp).items():
            if want and name not in want:
                continue
            fq, args = fn()
            dump = os.path.join(tmp, name + ".bin")
            cmd = [os.path.join(RBIN, "ref_dump_rtest"), dump] + args + [fq, "-o", os.path.join(tmp, name + "_out")]
            log = run(cmd, cwd=tmp)
            os.environ.pop("REF_DUMP_THREADS", None)
            d = parse_dump(dump)
            d.update(EXTRA.get(name, {}))
            d["cmdline"] = np.array(" ".join(a if not a.startswith(tmp) else os.path.basename(a) for a in args))
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
            kb = os.path.getsize(os.path.join(HERE, name + ".npz")) / 1024
            vals, cnt = np.unique(d["read_type"], return_counts=True)
            print("%-18s reads=%4d S=%d H=%3d C=%3d thr=%.4f outcomes=%s  %.0f KB" % (
                name, d["n_reads"], d["S"], d["H"], d["C"], d["threshold"],
                dict(zip(vals.tolist(), cnt.tolist())), kb))
            tail = [l for l in log.splitlines() if "WARNING" in l]
            for l in tail:
                print("   ", l)


if __name__ == "__main__":
    main()
