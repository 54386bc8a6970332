"""End-to-end drop-in check on the GPU box: the reference's own CLI / FASTQ I/O / controller / threshold
calibration with run_pHMM() bound to libtagdust_hip.so through integration/run_phmm_shim.c
(oracle/_ref/tagdust_hip_rtest) must write byte-identical output files to the unmodified reference binary
(oracle/_ref/tagdust_rtest), and reproduce the reference's own regression gold line for
dev/bar_read_test.sh scenario 1.  Both binaries are built by `make -C oracle ref` in the build container."""
import glob
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import load_golden, REPO

pytestmark = pytest.mark.gpu

RBIN = os.path.join(REPO, "oracle", "_ref")
NEED = ["tagdust_rtest", "tagdust_hip_rtest", "simreads_rtest", "evalres_rtest"]
# dev/barread1_tagdust_results_gold.txt:2 (the reference's regression value for scenario 1)
SCEN1_GOLD = "tagdust\t1.0000\t0.9542\t0.9947\t0.9739\t0.0000\t8951.00\t48.00\t0.00\t1001.00"


def _have():
    return all(os.path.exists(os.path.join(RBIN, b)) for b in NEED)


def _write_fastq(g, path):
    names = bytes(g["names"]).split(b"\n")
    offs = g["offs"]
    with open(path, "wb") as fh:
        for i in range(int(g["n_reads"])):
            s = bytes(np.frombuffer(b"ACGTN", np.uint8)[g["seq"][offs[i]:offs[i + 1]]])
            q = bytes(g["qual"][offs[i]:offs[i + 1]])
            fh.write(b"@" + names[i] + b"\n" + s + b"\n+\n" + q + b"\n")


def _run(binary, args, cwd, env=None, strict=True, check_rc=True):
    """Run a binary of oracle/_ref.  The GPU-bound one runs with TAGDUST_HIP_STRICT=1 (the shim fails instead of handing a batch
    to the reference's CPU code) and must report at exit that it decoded at least one batch on the GPU and delegated none:
    a byte-identical file then really comes from the HIP path.  check_rc=False: runs that are meant to fail -- when run_pHMM
    returns kslFAIL the reference's own error path crashes more often than it exits (main.c:209-215)."""
    e = dict(os.environ)
    if binary == "tagdust_hip_rtest" and strict:
        e["TAGDUST_HIP_STRICT"] = "1"
    e.update(env or {})
    p = subprocess.run([os.path.join(RBIN, binary)] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=e)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 or not check_rc, out[-2000:]
    if binary == "tagdust_hip_rtest" and strict:
        assert "TAGDUST_HIP_STRICT: refusing" not in out, out[-2000:]
        rep = re.findall(r"tagdust_hip: batches gpu=(\d+) delegated=(\d+)", out)
        assert rep and int(rep[-1][0]) > 0 and int(rep[-1][1]) == 0, out[-2000:]
    return out


def _outputs(d, prefix):
    out = {}
    for p in sorted(glob.glob(os.path.join(d, prefix + "*"))):
        if p.endswith("_logfile.txt"):
            continue
        out[os.path.basename(p)[len(prefix):]] = open(p, "rb").read()
    return out


@pytest.mark.skipif(not _have(), reason="oracle/_ref binaries not built (need the build container's `make -C oracle ref`)")
@pytest.mark.parametrize("name", ["c2_b4_r", "c3_b6_s_r_p", "scen2_endloss", "umi_f_s_r", "b_r_s_r", "dust_b_r", "window_b_r"])
def test_reference_cli_with_gpu_run_phmm_writes_identical_files(tmp_path, name):
    g = load_golden(name)
    fq = str(tmp_path / "in.fq")
    _write_fastq(g, fq)
    args = str(g["cmdline"]).split()
    _run("tagdust_rtest", args + [fq, "-o", "cpu"], str(tmp_path))
    log = _run("tagdust_hip_rtest", args + [fq, "-o", "gpu"], str(tmp_path))
    cpu, gpu = _outputs(str(tmp_path), "cpu"), _outputs(str(tmp_path), "gpu")
    assert cpu and set(cpu) == set(gpu), (sorted(cpu), sorted(gpu), log[-1500:])
    for k in cpu:
        assert cpu[k] == gpu[k], "output file *%s differs" % k


@pytest.mark.skipif(not _have(), reason="oracle/_ref binaries not built")
def test_bar_read_test_scenario1_gold(tmp_path):
    """dev/bar_read_test.sh scenario 1 end to end with the GPU-bound binary: simreads_rtest -> tagdust -> evalres."""
    d = str(tmp_path)
    tags = os.path.join(REPO, "tests", "golden", "EDITTAG_6nt_ed_4_first4.txt")
    _run("simreads_rtest", [tags, "-seed", "42", "-sim_barnum", "4", "-sim_readlen", "20", "-sim_readlen_mod", "0",
                            "-sim_numseq", "10000", "-sim_endloss", "0", "-sim_random_frac", "0.1", "-o", "barread1.fq",
                            "-sim_error_rate", "0.02"], d)
    _run("tagdust_hip_rtest", ["-seed", "42", "barread1.fq", "-arch", "barread1.fq_tagdust_arch.txt", "-o", "barread1_tagdust"], d)
    fqs = sorted(glob.glob(os.path.join(d, "barread1_tagdust*.fq")))
    _run("evalres_rtest", ["-name", "tagdust"] + [os.path.basename(f) for f in fqs] + ["-o", "barread1_tagdust"], d)
    lines = open(os.path.join(d, "barread1_tagdust_results.txt")).read().splitlines()
    assert SCEN1_GOLD in [l.strip() for l in lines], lines


@pytest.mark.skipif(not _have(), reason="oracle/_ref binaries not built")
def test_architecture_selection_through_gpu(tmp_path):
    """-arch file with several candidate architectures: test_architectures() -> run_pHMM(MODE_ARCH_COMP) on the GPU must pick
    the same architecture (same posteriors in the log) and write the same files as the CPU reference."""
    g = load_golden("c2_b4_r")
    fq = str(tmp_path / "in.fq")
    _write_fastq(g, fq)
    arch = str(tmp_path / "arch.txt")
    real = " ".join(str(g["cmdline"]).split()[2:])
    open(arch, "w").write("tagdust -1 R:N\n" + "tagdust " + real + "\n" + "tagdust -1 B:GGGG,CCCC -2 R:N\n")
    cpu_log = _run("tagdust_rtest", ["-seed", "42", "-t", "3", "-arch", arch, fq, "-o", "cpu"], str(tmp_path))
    gpu_log = _run("tagdust_hip_rtest", ["-seed", "42", "-t", "3", "-arch", arch, fq, "-o", "gpu"], str(tmp_path))
    pick = lambda log: [l.split("]", 1)[-1].strip() for l in log.splitlines() if "Confidence" in l or "Using:" in l or "-1 " in l]
    assert pick(cpu_log) == pick(gpu_log) and pick(cpu_log)
    cpu, gpu = _outputs(str(tmp_path), "cpu"), _outputs(str(tmp_path), "gpu")
    assert cpu and set(cpu) == set(gpu)
    for k in cpu:
        assert cpu[k] == gpu[k], "output file *%s differs" % k


# the other three scenarios of dev/bar_read_test.sh with their gold lines (dev/barread2_..., read_paired_..., barread_paired_...)
GOLD2 = "tagdust\t1.0000\t0.9775\t0.9974\t0.9874\t0.0013\t8976.00\t23.00\t0.00\t1001.00"
GOLD3 = "tagdust\t1.0000\t1.0000\t1.0000\t1.0000\t0.0017\t8999.00\t0.00\t0.00\t1001.00"
SIM_COMMON = ["-sim_readlen", "20", "-sim_readlen_mod", "0", "-sim_numseq", "10000", "-sim_endloss", "0", "-sim_error_rate", "0.02"]


def _sim(d, out, extra):
    tags = os.path.join(REPO, "tests", "golden", "EDITTAG_6nt_ed_4_first4.txt")
    _run("simreads_rtest", [tags, "-seed", "42"] + extra + SIM_COMMON + ["-o", out], d)


def _eval(d, pattern, out):
    fqs = sorted(glob.glob(os.path.join(d, pattern)))
    assert fqs, pattern
    _run("evalres_rtest", ["-name", "tagdust"] + [os.path.basename(f) for f in fqs] + ["-o", out], d)
    return [l.strip() for l in open(os.path.join(d, out + "_results.txt")).read().splitlines()]


@pytest.mark.skipif(not _have(), reason="oracle/_ref binaries not built")
def test_bar_read_test_scenario2_gold(tmp_path):
    """single-end, 5' and 3' partial linkers + barcode + read"""
    d = str(tmp_path)
    _sim(d, "barread2.fq", ["-sim_barnum", "4", "-sim_5seq", "GGGGGGG", "-sim_3seq", "TTTTTTT", "-sim_random_frac", "0.1"])
    _run("tagdust_hip_rtest", ["-seed", "42", "barread2.fq", "-arch", "barread2.fq_tagdust_arch.txt", "-o", "barread2_tagdust"], d)
    assert GOLD2 in _eval(d, "barread2_tagdust*.fq", "barread2_tagdust")


@pytest.mark.skipif(not _have(), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("barnum,gold", [("0", GOLD3), ("4", GOLD2)], ids=["scenario3_linkers", "scenario4_barcode"])
def test_bar_read_test_paired_scenarios_gold(tmp_path, barnum, gold):
    """paired-end: read 1 carries the architecture (through the GPU), read 2 is 'R:N'; outcomes and barcodes are
    combined per record index by the reference's controller (barcode_hmm.c:329-351)"""
    d = str(tmp_path)
    _sim(d, "r1.fq", ["-sim_barnum", barnum, "-sim_5seq", "GGGGGGG", "-sim_3seq", "TTTTTTT", "-sim_random_frac", "0.1"])
    _sim(d, "r2.fq", ["-sim_barnum", "0", "-sim_random_frac", "0.00"])
    arch = open(os.path.join(d, "r1.fq_tagdust_arch.txt")).read() + open(os.path.join(d, "r2.fq_tagdust_arch.txt")).read()
    open(os.path.join(d, "combo_arch.txt"), "w").write(arch)
    _run("tagdust_hip_rtest", ["-seed", "42", "-sim_numseq", "1", "r1.fq", "r2.fq", "-arch", "combo_arch.txt", "-o", "paired_tagdust"], d)
    assert gold in _eval(d, "paired_tagdust_*READ1.fq", "paired_tagdust")


@pytest.mark.skipif(not _have(), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("threads", [1, 3])
def test_reference_cli_with_artifact_filter(tmp_path, threads):
    """-ref: match_to_reference runs on the device (td_set_artifacts); the thread count decides which reads are the
    left-over reads of a range, so both binaries get the same -t."""
    g = load_golden("artifacts_b_r")
    fq, fa = str(tmp_path / "in.fq"), str(tmp_path / "art.fa")
    _write_fastq(g, fq)
    open(fa, "wb").write(bytes(g["art_fasta_text"]))
    args = str(g["cmdline"]).split()
    args[args.index("-ref") + 1] = fa
    args += ["-t", str(threads)]
    _run("tagdust_rtest", args + [fq, "-o", "cpu"], str(tmp_path))
    log = _run("tagdust_hip_rtest", args + [fq, "-o", "gpu"], str(tmp_path))
    cpu, gpu = _outputs(str(tmp_path), "cpu"), _outputs(str(tmp_path), "gpu")
    assert cpu and set(cpu) == set(gpu), (sorted(cpu), sorted(gpu), log[-1500:])
    for k in cpu:
        assert cpu[k] == gpu[k], "output file *%s differs" % k
    # the per-artifact hit counts the controller logs (barcode_hmm.c:423-428) come from read_type >> 8
    def hits(prefix):
        lines = open(os.path.join(str(tmp_path), prefix + "_logfile.txt")).read().splitlines()
        return sorted(l.split("\t", 1)[1] for l in lines if "artifact_" in l)   # drop the time stamp
    assert hits("cpu") == hits("gpu") and hits("cpu")


@pytest.mark.skipif(not _have(), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("name,extra", [("artifacts_b_r", ["-t", "3"]), ("c3_b6_s_r_p", []), ("c2_indel_varlen", ["-t", "5"]), ("window_b_r", [])])
def test_reference_cli_over_several_contexts(tmp_path, name, extra):
    """TAGDUST_HIP_DEVICES: the shim hands every batch to td_multi_decode, which splits it like run_pHMM splits it over
    threads and merges in input order.  With one physical GPU the list names device 0 three times (three contexts, host sum
    of the counters); the files must still equal the CPU reference's, artifact thread ranges included."""
    g = load_golden(name)
    fq = str(tmp_path / "in.fq")
    _write_fastq(g, fq)
    args = str(g["cmdline"]).split() + extra
    if "-ref" in args:
        fa = str(tmp_path / "art.fa")
        open(fa, "wb").write(bytes(g["art_fasta_text"]))
        args[args.index("-ref") + 1] = fa
    _run("tagdust_rtest", args + [fq, "-o", "cpu"], str(tmp_path))
    log = _run("tagdust_hip_rtest", args + [fq, "-o", "gpu"], str(tmp_path), env={"TAGDUST_HIP_DEVICES": "0,0,0", "TAGDUST_HIP_ALLOW_DUPLICATE_DEVICES": "1"})
    cpu, gpu = _outputs(str(tmp_path), "cpu"), _outputs(str(tmp_path), "gpu")
    assert cpu and set(cpu) == set(gpu), (sorted(cpu), sorted(gpu), log[-1500:])
    for k in cpu:
        assert cpu[k] == gpu[k], "output file *%s differs" % k


@pytest.mark.skipif(not _have(), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("name", ["c3_b6_s_r_p", "dust_b_r"])
def test_reference_cli_through_the_synchronous_calls(tmp_path, name):
    """TAGDUST_HIP_SYNC=1: the shim's single-context path (td_batch_upload / td_run / td_batch_download per run_pHMM call) -- the
    default goes through td_multi_decode's pipelined pieces."""
    g = load_golden(name)
    fq = str(tmp_path / "in.fq")
    _write_fastq(g, fq)
    args = str(g["cmdline"]).split()
    _run("tagdust_rtest", args + [fq, "-o", "cpu"], str(tmp_path))
    _run("tagdust_hip_rtest", args + [fq, "-o", "gpu"], str(tmp_path), env={"TAGDUST_HIP_SYNC": "1"})
    cpu, gpu = _outputs(str(tmp_path), "cpu"), _outputs(str(tmp_path), "gpu")
    assert cpu and set(cpu) == set(gpu)
    for k in cpu:
        assert cpu[k] == gpu[k], "output file *%s differs" % k


@pytest.mark.skipif(not _have(), reason="oracle/_ref binaries not built")
def test_config0_read_only_architecture(tmp_path):
    """BASELINE.json configs[0]: '-1 R:N' (no barcode).  The label phase of such a run is run_rna_dust() (no HMM), but
    the threshold calibration still pushes its simulated reads through run_pHMM(MODE_GET_PROB) with the one-HMM model:
    the GPU-bound binary must select the same threshold and write the same files as the reference."""
    d = str(tmp_path)
    tags = os.path.join(REPO, "tests", "golden", "EDITTAG_6nt_ed_4_first4.txt")
    _run("simreads_rtest", [tags, "-seed", "42", "-sim_barnum", "0", "-sim_readlen", "50", "-sim_readlen_mod", "0",
                            "-sim_numseq", "3000", "-sim_endloss", "0", "-sim_random_frac", "0.1", "-sim_error_rate", "0.02",
                            "-o", "c0.fq"], d)
    _run("tagdust_rtest", ["-seed", "42", "-1", "R:N", "c0.fq", "-o", "cpu"], d)
    log = _run("tagdust_hip_rtest", ["-seed", "42", "-1", "R:N", "c0.fq", "-o", "gpu"], d)
    cpu, gpu = _outputs(d, "cpu"), _outputs(d, "gpu")
    assert cpu and set(cpu) == set(gpu), (sorted(cpu), sorted(gpu), log[-1500:])
    for k in cpu:
        assert cpu[k] == gpu[k], "output file *%s differs" % k
    thr = lambda p: [l.split("\t", 1)[1] for l in open(os.path.join(d, p + "_logfile.txt")).read().splitlines() if "selected threshold" in l]
    assert thr("cpu") == thr("gpu") and thr("cpu")


@pytest.mark.skipif(not _have(), reason="oracle/_ref binaries not built")
def test_casava_three_file_run_with_arch_file(tmp_path):
    """BASELINE.json configs[3] shape (dev/casava_test.sh; its read-1/read-3 files are not in the checkout, so they are
    synthesised with matching CASAVA-1.8 names): three input files, '-arch' file with the index-read and the plain-read
    architecture.  Per file the controller picks an architecture (MODE_ARCH_COMP on the GPU), the index file goes through
    calibration and labelling on the GPU, the other two through run_rna_dust, and print_all combines them -- the
    GPU-bound binary must write the same files as the reference."""
    g = load_golden("casava_index")
    d = str(tmp_path)
    rng = np.random.RandomState(11)
    names = bytes(g["names"]).split(b"\n")
    offs = g["offs"]
    with open(os.path.join(d, "r1.fq"), "wb") as f1, open(os.path.join(d, "r2.fq"), "wb") as f2, open(os.path.join(d, "r3.fq"), "wb") as f3:
        for i in range(int(g["n_reads"])):
            base = names[i].split(b" ")[0]
            s2 = bytes(np.frombuffer(b"ACGTN", np.uint8)[g["seq"][offs[i]:offs[i + 1]]])
            f2.write(b"@" + names[i] + b"\n" + s2 + b"\n+\n" + bytes(g["qual"][offs[i]:offs[i + 1]]) + b"\n")
            for fh, k in ((f1, b"1"), (f3, b"3")):
                s_ = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, 76)])
                fh.write(b"@" + base + b" " + k + b":N:0:\n" + s_ + b"\n+\n" + b"I" * 76 + b"\n")
    with open(os.path.join(d, "arch.txt"), "w") as fh:
        fh.write("tagdust " + " ".join(str(g["cmdline"]).split()[2:]) + "\n")
        fh.write("tagdust -1 R:N\n")
    args = ["-seed", "42", "-arch", "arch.txt", "r1.fq", "r2.fq", "r3.fq"]
    _run("tagdust_rtest", args + ["-o", "cpu"], d)
    log = _run("tagdust_hip_rtest", args + ["-o", "gpu"], d)
    cpu, gpu = _outputs(d, "cpu"), _outputs(d, "gpu")
    assert len(cpu) > 4 and set(cpu) == set(gpu), (sorted(cpu), sorted(gpu), log[-1500:])
    for k in cpu:
        assert cpu[k] == gpu[k], "output file *%s differs" % k
    assert any("_READ1" in k for k in cpu) and any("_READ2" in k for k in cpu)


@pytest.mark.skipif(not _have(), reason="oracle/_ref binaries not built")
def test_inputs_the_reference_crashes_on_are_not_delegated(tmp_path):
    """A -start/-end batch with a read shorter than -end makes the reference read past that read and abort ("munmap_chunk():
    invalid pointer", exit 134 -- checked here against the unmodified binary), and a window without an end makes it decode a
    negative length (segfault).  Neither is handed to the reference's code: the first is decoded on the GPU on what the read has
    inside the window (strict run, delegated=0, exit 0), the second is refused with a message."""
    g = load_golden("window_b_r")
    fq = str(tmp_path / "in.fq")
    _write_fastq(g, fq)
    lines = open(fq, "rb").read().split(b"\n")
    lines[1], lines[3] = lines[1][:40], lines[3][:40]          # the first read now ends before -end 61
    open(fq, "wb").write(b"\n".join(lines))
    args = str(g["cmdline"]).split()
    p = subprocess.run([os.path.join(RBIN, "tagdust_rtest")] + args + [fq, "-o", "cpu"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode != 0                                   # the reference's own behaviour on this input: a crash
    _run("tagdust_hip_rtest", args + [fq, "-o", "gpu"], str(tmp_path))     # strict: gpu > 0, delegated = 0, exit 0
    assert _outputs(str(tmp_path), "gpu")
    bad = [a for a in args if a not in ("-end", "61")]
    out = _run("tagdust_hip_rtest", bad + [fq, "-o", "bad"], str(tmp_path), strict=False, check_rc=False)
    assert "is not a window" in out and not re.findall(r"batches gpu=[1-9]", out)


@pytest.mark.skipif(not _have(), reason="oracle/_ref binaries not built")
def test_delegation_is_counted_and_strict_refuses_it(tmp_path):
    """TAGDUST_HIP_DELEGATE=4 hands the calibration batches (MODE_GET_PROB) to the reference's own code and keeps the labelling on
    the GPU: same files, and the exit report counts the hand-over.  With TAGDUST_HIP_STRICT=1 on top the shim refuses instead."""
    g = load_golden("c2_b4_r")
    fq = str(tmp_path / "in.fq")
    _write_fastq(g, fq)
    args = str(g["cmdline"]).split()
    assert "-Q" not in args                                    # this run calibrates its threshold
    _run("tagdust_rtest", args + [fq, "-o", "cpu"], str(tmp_path))
    out = _run("tagdust_hip_rtest", args + [fq, "-o", "gpu"], str(tmp_path), env={"TAGDUST_HIP_DELEGATE": "4"}, strict=False)
    rep = re.findall(r"tagdust_hip: batches gpu=(\d+) delegated=(\d+)", out)
    assert rep and int(rep[-1][0]) >= 1 and int(rep[-1][1]) >= 1, out[-1500:]
    cpu, gpu = _outputs(str(tmp_path), "cpu"), _outputs(str(tmp_path), "gpu")
    assert cpu and set(cpu) == set(gpu)
    for k in cpu:
        assert cpu[k] == gpu[k], "output file *%s differs" % k
    out = _run("tagdust_hip_rtest", args + [fq, "-o", "strict"], str(tmp_path), env={"TAGDUST_HIP_DELEGATE": "4", "TAGDUST_HIP_STRICT": "1"}, strict=False, check_rc=False)
    assert "TAGDUST_HIP_STRICT: refusing" in out and "TAGDUST_HIP_DELEGATE" in out


@pytest.mark.skipif(not _have(), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("devs", ["gpu0", "0;1", "0,0", "-1", ""], ids=["text", "semicolon", "twice", "negative", "empty-uses-default"])
def test_device_list_parse_errors_are_reported(tmp_path, devs):
    """TAGDUST_HIP_DEVICES must be a list of distinct non-negative indices; anything else ends the run with a message instead
    of 64 contexts on device 0."""
    g = load_golden("c2_b4_r")
    fq = str(tmp_path / "in.fq")
    _write_fastq(g, fq)
    args = str(g["cmdline"]).split()
    out = _run("tagdust_hip_rtest", args + [fq, "-o", "x"], str(tmp_path), env={"TAGDUST_HIP_DEVICES": devs}, strict=(devs == ""), check_rc=(devs == ""))
    if devs:
        assert "TAGDUST_HIP_DEVICES" in out and not re.findall(r"batches gpu=[1-9]", out), out[-1500:]
