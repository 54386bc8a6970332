"""Host model construction (include/tagdust_model.h, tagdust_amd/csrc/td_model.cpp) against the reference's own
tables: for every fixture, parsing the architecture, computing the sequence statistics from the fixture's reads and
building the model must reproduce init_model_bag()'s tables bit for bit (tests/golden/*.npz hold what the
reference built)."""
import numpy as np

from conftest import golden_window
from tagdust_amd import lib as tdlib


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _segments(g):
    """Re-create the -1..-N option arguments from the fixture's parsed read structure (decoys stripped)."""
    segs = []
    for t, grp in zip(g["seg_type"], str(g["seg_seqs"]).split(";")):
        t = chr(int(t))
        seqs = grp.split(",")
        if t in "BS":
            seqs = seqs[:-1]          # the all-N decoy interface.c adds
        segs.append("%s:%s" % (t, ",".join(seqs)))
    return segs


def test_sequence_stats_match_reference(golden):
    g = golden
    _, st = tdlib.build_model(_segments(g), g["seq"], g["offs"], float(g["e"]), float(g["d"]), window=golden_window(g))
    assert np.array_equal(np.array(st["background"]), g["ssi_background"])
    for k in ("expected_5_len", "expected_3_len", "mean_5_len", "stdev_5_len", "mean_3_len", "stdev_3_len", "average_length"):
        assert st[k] == float(g["ssi_" + k]), k


def test_model_tables_match_reference(golden):
    g = golden
    # the reference inflates max_seq_len during calibration only; it does not enter any table
    md, _ = tdlib.build_model(_segments(g), g["seq"], g["offs"], float(g["e"]), float(g["d"]), window=golden_window(g))
    for k in ("S", "H", "C", "avg_len"):
        assert int(md[k]) == int(g[k]), k
    for k in ("n_hmm", "n_col", "seg_type", "label"):
        assert np.array_equal(md[k], g[k]), k
    for k in ("bg", "skip", "trans", "eM", "eI", "sM", "sI", "A"):
        assert np.array_equal(_bits(md[k]), _bits(g[k])), k
