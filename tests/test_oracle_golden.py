"""Pins the CPU oracle (oracle/td_oracle.c) bit-for-bit against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py through oracle/_ref/ref_dump_rtest)."""
import numpy as np

from conftest import golden_artifacts, golden_window
from oracle import pyoracle


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_logsum_table_matches_reference_formula():
    t = pyoracle.logsum_table()
    i = np.arange(16000, dtype=np.float64)
    # misc.c:57-63: (float) log(1. + exp((double) -i / 1000.0f))
    want = np.log(1.0 + np.exp(-i / 1000.0)).astype(np.float32)
    assert np.array_equal(_bits(t), _bits(want))


def test_oracle_bit_exact_vs_reference(golden):
    g = golden
    model = pyoracle.OracleModel(g)
    art = golden_artifacts(g)  # artifact matching depends on the thread split: use the fixture's thread count
    res, labels, seq_after = pyoracle.label_batch(model, g["seq"], g["offs"], float(g["threshold"]),
                                                  minlen=int(g["minlen"]), dust=int(g["dust"]),
                                                  n_threads=art[3] if art else 2,
                                                  artifacts=art[:3] if art else None, window=golden_window(g))
    assert np.array_equal(_bits(res["b_score"]), _bits(g["b_score"]))
    assert np.array_equal(_bits(res["f_score"]), _bits(g["f_score"]))
    assert np.array_equal(_bits(res["r_score"]), _bits(g["r_score"]))
    assert np.array_equal(_bits(res["bar_prob"]), _bits(g["bar_prob"].astype(np.float32)))
    assert np.array_equal(g["bar_prob"].astype(np.float32).astype(np.float64), g["bar_prob"])  # it IS a float
    assert np.array_equal(labels, g["labels"])
    assert np.array_equal(_bits(res["Q"]), _bits(g["mapq"]))
    assert np.array_equal(res["read_type"], g["read_type"])
    assert np.array_equal(res["barcode"], g["barcode"])
    assert np.array_equal(res["fingerprint"], g["fingerprint"])
    assert np.array_equal(seq_after, g["seq_after"])


def test_oracle_thread_split_is_partition_independent(golden):
    g = golden
    model = pyoracle.OracleModel(g)
    a = pyoracle.label_batch(model, g["seq"], g["offs"], float(g["threshold"]), int(g["minlen"]), int(g["dust"]), 1, window=golden_window(g))
    b = pyoracle.label_batch(model, g["seq"], g["offs"], float(g["threshold"]), int(g["minlen"]), int(g["dust"]), 5, window=golden_window(g))
    assert a[0].tobytes() == b[0].tobytes()
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_artifact_left_over_reads_take_the_other_routine():
    """match_to_reference pairs reads in fours per thread range; the left-over reads of a range go through
    bpm_check_error (first hit, 31-character cap), so the outcome of a read can depend on the thread count --
    the fixture (3 threads) pins that, and a different split must change at least the left-over set's treatment
    without touching anything but read_type."""
    from conftest import load_golden
    g = load_golden("artifacts_b_r")
    art = golden_artifacts(g)
    model = pyoracle.OracleModel(g)
    base = pyoracle.label_batch(model, g["seq"], g["offs"], float(g["threshold"]), int(g["minlen"]), int(g["dust"]),
                                n_threads=art[3], artifacts=art[:3])
    assert np.array_equal(base[0]["read_type"], g["read_type"])
    assert ((g["read_type"] & 0xFF) == 5).sum() > 20 and len(set((g["read_type"] >> 8).tolist())) == 4
    none = pyoracle.label_batch(model, g["seq"], g["offs"], float(g["threshold"]), int(g["minlen"]), int(g["dust"]), 3)
    changed = none[0]["read_type"] != base[0]["read_type"]
    assert np.array_equal(changed, (g["read_type"] & 0xFF) == 5)           # the filter only turns successes into 5s
    for k in ("b_score", "f_score", "Q", "barcode", "fingerprint"):
        assert np.array_equal(none[0][k], base[0][k])
    assert np.array_equal(none[1], base[1]) and np.array_equal(none[2], base[2])   # seq is restored after the in-place reverse complement
