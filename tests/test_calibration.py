"""Threshold calibration (include/tagdust_model.h: td_calibration_emit / td_calibration_select /
td_estimate_threshold) against the reference's estimateQthreshold() (src/calibrateQ.c:17-235).  The fixtures were
produced by the reference's -DRTEST build with -seed 42 (4000 simulated reads, private LCG), so emitting with the same
generator, scoring the emitted reads and running the sweep must give the fixture's threshold bit for bit."""
import numpy as np
import pytest

from conftest import load_golden, GOLDEN_NAMES
from tagdust_amd import lib as tdlib
from test_model_builder import _segments

CALIBRATED = [n for n in GOLDEN_NAMES if n not in ("short_q_given", "window_b_r")]   # those were run with -Q (no calibration)


@pytest.mark.parametrize("name", CALIBRATED)
def test_calibration_with_oracle_scoring(name):
    from oracle import pyoracle
    g = load_golden(name)
    assert int(g["q_given"]) == 0
    segs = _segments(g)
    codes, offs, is_random = tdlib.calibration_emit(segs, g["seq"], g["offs"], float(g["d"]), seed=42, n_reads=4000, rng=1)
    assert len(is_random) == 4000 and is_random[:2000].sum() == 0 and is_random[2000:].sum() == 2000
    scoring, _ = tdlib.build_model(segs, g["seq"], g["offs"], 0.05, float(g["d"]))
    res, _, _ = pyoracle.label_batch(pyoracle.OracleModel(scoring), codes, offs, 0.0, int(g["minlen"]), 0, 4)
    thr = tdlib.calibration_select(res["Q"], is_random)
    assert np.float32(thr).view(np.uint32) == np.float32(g["threshold"]).view(np.uint32), (thr, float(g["threshold"]))


def test_libc_rng_variant_is_deterministic():
    g = load_golden("c2_b4_r")
    a = tdlib.calibration_emit(_segments(g), g["seq"], g["offs"], 0.1, seed=7, n_reads=400, rng=0)
    b = tdlib.calibration_emit(_segments(g), g["seq"], g["offs"], 0.1, seed=7, n_reads=400, rng=0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert (np.diff(a[1]) >= int(g["avg_len"])).all()          # every emitted read reaches the average length


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2_b4_r", "c3_b6_s_r_p", "scen2_endloss", "o_b_s_r", "c5_b96_f_r_p"])
def test_estimate_threshold_on_gpu(name):
    from tagdust_amd import TagdustHip
    g = load_golden(name)
    c = TagdustHip(0)
    try:
        thr = tdlib.estimate_threshold(c, _segments(g), g["seq"], g["offs"], float(g["d"]), seed=42, n_reads=4000, rng=1)
    finally:
        c.close()
    assert np.float32(thr).view(np.uint32) == np.float32(g["threshold"]).view(np.uint32), (thr, float(g["threshold"]))


@pytest.mark.gpu
def test_compare_architectures_on_gpu():
    """td_compare_architectures (test_architectures(), backward-only scoring on the GPU) against the same quantity formed
    from the oracle's b_scores with the reference's summation order; the winner is the architecture the reads were
    simulated with (and what the reference CLI picks, tests/test_dropin_gpu.py)."""
    from oracle import pyoracle
    from tagdust_amd import TagdustHip
    g = load_golden("c2_b4_r")
    real = _segments(g)
    cands = [["R:N"], real, ["B:GGGG,CCCC", "R:N"]]
    T = 3
    c = TagdustHip(0)
    try:
        post, best = tdlib.compare_architectures(c, cands, g["seq"], g["offs"], 0.05, 0.1, n_threads=T)
    finally:
        c.close()
    assert best == 1 and abs(float(post.sum()) - 1.0) < 1e-3
    n = int(g["n_reads"])
    tot = []
    for segs in cands:
        md, _ = tdlib.build_model(segs, g["seq"], g["offs"], 0.05, 0.1)
        md.update(threshold=0.0, minlen=16, dust=0)
        res, _, _ = pyoracle.label_batch(pyoracle.OracleModel(md), g["seq"], g["offs"], 0.0, 16, 0, 4)
        b = res["b_score"]
        total = np.float32(0.0)
        interval = n // T
        for t in range(T):
            lo, hi = t * interval, (n if t == T - 1 else (t + 1) * interval)
            part = np.float32(0.0)
            for x in b[lo:hi]:
                part = np.float32(part + x)
            total = np.float32(total + part)
        tot.append(total)
    lsum = pyoracle.lib().tdo_logsum
    s_ = tot[0]
    for x in tot[1:]:
        s_ = np.float32(lsum(float(s_), float(x)))
    norm = [np.float32(x - s_) for x in tot]
    z = np.float32(-np.inf)
    for x in norm:
        z = np.float32(lsum(float(z), float(x)))
    want = np.array([np.float32(np.exp(np.float64(np.float32(x - z)))) if np.isfinite(x - z) else np.float32(0) for x in norm], np.float32)
    assert np.array_equal(post.view(np.uint32), want.view(np.uint32)), (post, want)


def test_libc_rng_variant_matches_the_reference_binary(tmp_path):
    """The production path (C library rand(), 400 000 simulated reads): the threshold td_calibration_emit + scoring +
    td_calibration_select give must be the one the reference's regular (non -DRTEST) binary logs for the same input
    and -seed.  Needs oracle/_ref/tagdust (built in the build container by `make -C oracle ref`); the emitted reads are
    scored by the CPU oracle here, by the device in td_estimate_threshold."""
    import os
    import re
    import subprocess
    from conftest import REPO
    from oracle import pyoracle
    exe = os.path.join(REPO, "oracle", "_ref", "tagdust")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/tagdust not built")
    g = load_golden("c2_b4_r")
    fq = str(tmp_path / "in.fq")
    names = bytes(g["names"]).split(b"\n")
    offs = g["offs"]
    with open(fq, "wb") as fh:
        for i in range(int(g["n_reads"])):
            s = bytes(np.frombuffer(b"ACGTN", np.uint8)[g["seq"][offs[i]:offs[i + 1]]])
            fh.write(b"@" + names[i] + b"\n" + s + b"\n+\n" + bytes(g["qual"][offs[i]:offs[i + 1]]) + b"\n")
    p = subprocess.run([exe, "-t", "8"] + str(g["cmdline"]).split() + [fq, "-o", str(tmp_path / "out")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-1500:]
    m = re.search(r"Selected Threshold::? ([0-9.eE+-]+)", p.stdout.decode(errors="replace"))
    assert m, p.stdout.decode(errors="replace")[-1500:]
    want = float(m.group(1))
    segs = _segments(g)
    codes, offs2, is_random = tdlib.calibration_emit(segs, g["seq"], g["offs"], float(g["d"]), seed=42, n_reads=400000, rng=0)
    scoring, _ = tdlib.build_model(segs, g["seq"], g["offs"], 0.05, float(g["d"]))
    res, _, _ = pyoracle.label_batch(pyoracle.OracleModel(scoring), codes, offs2, 0.0, int(g["minlen"]), 0, 8)
    thr = tdlib.calibration_select(res["Q"], is_random)
    assert abs(thr - want) < 5e-7 * max(1.0, abs(want)), (thr, want)       # the log prints %f (six decimals)
