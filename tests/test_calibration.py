"""Threshold calibration (include/tagdust_model.h: td_calibration_emit / td_calibration_select /
td_estimate_threshold) against the reference's estimateQthreshold() (src/calibrateQ.c:17-235).  The fixtures were
produced by the reference's -DRTEST build with -seed 42 (4000 simulated reads, private LCG), so emitting with the same
generator, scoring the emitted reads and running the sweep must give the fixture's threshold bit for bit."""
import numpy as np
import pytest

from conftest import load_golden, GOLDEN_NAMES
from tagdust_amd import lib as tdlib
from test_model_builder import _segments

CALIBRATED = [n for n in GOLDEN_NAMES if n != "short_q_given"]   # that one was run with -Q (no calibration)


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
