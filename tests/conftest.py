import glob
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
GOLDEN_NAMES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


def golden_artifacts(g):
    """(string, s_index, filter_error, n_threads) of a fixture that was run with -ref, else None."""
    if "art_n" not in g:
        return None
    return g["art_string"], g["art_index"], int(g["art_filter_error"]), int(g["art_threads"])


@pytest.fixture(params=GOLDEN_NAMES)
def golden(request):
    d = load_golden(request.param)
    d["name"] = request.param
    return d


def golden_window(g):
    """(matchstart, matchend) of a fixture that was run with -start / -end, else None."""
    ms, me = int(g.get("matchstart", -1)), int(g.get("matchend", -1))
    return (ms, me) if (ms != -1 or me != -1) else None
