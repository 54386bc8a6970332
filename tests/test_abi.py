"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/tagdust_hip.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tagdust_amd import lib as tdlib
from tagdust_amd import build as tdbuild

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def library():
    tdbuild.build()
    return tdlib.load_library()


def test_header_symbols_all_exported(library):
    declared = set()
    for name, want in (("tagdust_hip.h", tdlib.ABI_SYMBOLS), ("tagdust_model.h", tdlib.MODEL_ABI_SYMBOLS),
                       ("tagdust_io.h", tdlib.IO_ABI_SYMBOLS), ("tagdust_multi.h", tdlib.MULTI_ABI_SYMBOLS)):
        hdr = open(os.path.join(REPO, "include", name)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        found = set(re.findall(r"\b(td_[a-z_0-9]+)\s*\(", hdr))
        assert found == set(want), name
        declared |= found
    assert declared, "no prototypes parsed"
    for name in sorted(declared):
        assert hasattr(library, name), name


def test_logsum_table_host_copy(library):
    # init_logsum(), src/misc.c:57-63
    p = library.td_logsum_table()
    t = np.ctypeslib.as_array(p, shape=(16000,))
    i = np.arange(16000, dtype=np.float64)
    want = np.log(1.0 + np.exp(-i / 1000.0)).astype(np.float32)
    assert np.array_equal(t.view(np.uint32), want.view(np.uint32))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback(library):
    with pytest.raises(tdlib.TdError) as e:
        tdlib.TagdustHip(0)
    assert "no CPU path" in str(e.value) or "HIP" in str(e.value)


def test_no_undeclared_c_entry_points(library):
    """Every td_* C symbol the library exports is declared in include/*.h (internal helpers are hidden)."""
    import re
    import subprocess
    from tagdust_amd import lib as tdlib
    out = subprocess.run(["nm", "-D", "--defined-only", tdlib.LIB_PATH], stdout=subprocess.PIPE, check=True).stdout.decode()
    exported = {l.split()[2] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T" and l.split()[2].startswith("td_")}
    declared = set()
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    for h in os.listdir(inc):
        declared |= set(re.findall(r"\b(td_[a-z_0-9]+)\s*\(", open(os.path.join(inc, h)).read()))
    assert exported <= declared, sorted(exported - declared)
