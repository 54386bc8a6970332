"""CPU-side checks of the model-specialised kernel's source generation (no GPU needed): the model section states the
tables exactly (hex-float literals round-trip), and the assembled source compiles for gfx950."""
import os
import re
import sys

import numpy as np
import pytest

from conftest import load_golden, REPO
from tagdust_amd import lib as tdlib
from tagdust_amd import build as tdbuild


@pytest.fixture(scope="module", autouse=True)
def _built():
    tdbuild.build()


def test_model_section_is_exact():
    g = load_golden("scen2_endloss")
    src = tdlib.spec_source(g)
    m = re.search(r"static constexpr GCol kCols\[(\d+)\] = \{\n(.*?)\n\};", src, re.S)
    assert m and int(m.group(1)) == int(g["C"])
    rows = m.group(2).strip().split("\n")
    assert len(rows) == int(g["C"])

    def val(tok):
        tok = tok.strip()
        if "inff" in tok:
            return -np.inf if tok.startswith("(-") else np.inf
        return float.fromhex(tok.rstrip("f"))
    for k, row in enumerate(rows):
        nums = [val(t) for t in re.findall(r"\(-__builtin_inff\(\)\)|-?0x[0-9a-fA-F.]+p[-+]?\d+f", row)]
        want = np.concatenate([g["trans"][k], [g["sM"][k], g["sI"][k]], g["eM"][k], g["eI"][k]])
        assert np.array_equal(np.array(nums, np.float32).view(np.uint32), want.astype(np.float32).view(np.uint32)), k
    assert "#define TD_S %d" % int(g["S"]) in src and "#define TD_H %d" % int(g["H"]) in src


def test_specialised_source_compiles_for_gfx950():
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import spec_check
    dt, info, nsrc = spec_check.check("umi_f_s_r", outdir=os.path.join(REPO, "tagdust_amd", "csrc", ".spec_check"))
    assert nsrc > 10000 and any("VGPRs" in i for i in info)
