"""Several devices driven from one process (include/tagdust_multi.h).  A GPU box of this pool has one MI355X, so what can
run here is N = 1 (must equal the single-context path bit for bit) and two contexts on the same device (the split, the
per-device windows of the artifact filter, the input-order merge and the counter sum; the RCCL all-reduce itself needs
distinct devices and is exercised by the driver's multi-GPU runs only)."""
import numpy as np
import pytest

from conftest import load_golden, golden_artifacts, golden_window

pytestmark = pytest.mark.gpu


def _single(g, seq, offs):
    from tagdust_amd import TagdustHip
    c = TagdustHip(0)
    try:
        art = golden_artifacts(g)
        if art:
            c.set_artifacts(art[0], art[1], art[2], art[3])
        c.upload_model(g)
        c.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
        c.set_window(*(golden_window(g) or (-1, -1)))
        c.upload_batch(seq, offs)
        c.counts_reset()
        c.run()
        return c.download() + (c.counts(),)
    finally:
        c.close()


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]], ids=["N1", "N2-same-device", "N3-same-device"])
@pytest.mark.parametrize("name", ["c3_b6_s_r_p", "artifacts_b_r", "c2_indel_varlen", "window_b_r"])
def test_multi_equals_single_context(name, devices):
    from tagdust_amd.lib import TagdustMulti
    g = load_golden(name)
    res1, lab1, seq1, cnt1 = _single(g, g["seq"], g["offs"])
    m = TagdustMulti(devices)
    try:
        assert not m.uses_rccl()          # one physical device: host sum
        art = golden_artifacts(g)
        if art:
            m.set_artifacts(art[0], art[1], art[2], art[3])
        m.upload_model(g)
        m.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
        m.set_window(*(golden_window(g) or (-1, -1)))
        m.counts_reset()
        res, lab, seq = m.decode(g["seq"], g["offs"])
        cnt = m.counts()
    finally:
        m.close()
    assert res.tobytes() == res1.tobytes()
    assert np.array_equal(lab, lab1) and np.array_equal(seq, seq1)
    assert np.array_equal(cnt, cnt1)
    assert np.array_equal(res["read_type"], g["read_type"]) and np.array_equal(lab, g["labels"])
    # the library's host-side counting agrees with the device counters
    from tagdust_amd import lib as tdlib
    assert np.array_equal(tdlib.count_outcomes(res, g["lens"]), cnt)


def test_multi_decode_ascii_and_empty_shards():
    """Fewer reads than devices (empty ranges for all but the last), FASTQ text input."""
    from tagdust_amd.lib import TagdustMulti
    g = load_golden("c2_b4_r")
    offs = g["offs"][:3].astype(np.int64)
    seq = g["seq"][:offs[-1]]
    text = np.frombuffer(b"ACGTN", np.uint8)[seq]
    m = TagdustMulti([0, 0, 0])
    try:
        m.upload_model(g)
        m.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
        res, lab, sq = m.decode(text, offs, ascii=True)
    finally:
        m.close()
    assert np.array_equal(res["read_type"], g["read_type"][:2])
    assert np.array_equal(lab, g["labels"][:offs[-1] + 2]) and np.array_equal(sq, g["seq_after"][:offs[-1]])


@pytest.mark.parametrize("pieces,devices", [("3", [0]), ("5", [0, 0]), ("64", [0])])
@pytest.mark.parametrize("name", ["artifacts_b_r", "c2_indel_varlen", "window_b_r"])
def test_multi_decode_in_pipelined_pieces(name, pieces, devices, monkeypatch):
    """td_multi_decode puts a device's range through td_submit / td_wait in pieces of whole tiles (TD_MULTI_PIECES; 2^17 reads each
    by default): ragged reads, an artifact filter whose thread ranges are those of the whole batch, a -start/-end
    window -- same bytes and counters as one synchronous call on one context; more pieces than the pipeline is deep, more
    pieces than tiles."""
    from tagdust_amd.lib import TagdustMulti
    monkeypatch.setenv("TD_MULTI_PIECES", pieces)
    g = load_golden(name)
    res1, lab1, seq1, cnt1 = _single(g, g["seq"], g["offs"])
    m = TagdustMulti(devices)
    try:
        art = golden_artifacts(g)
        if art:
            m.set_artifacts(art[0], art[1], art[2], art[3])
        m.upload_model(g)
        m.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
        m.set_window(*(golden_window(g) or (-1, -1)))
        m.counts_reset()
        res, lab, seq = m.decode(g["seq"], g["offs"])
        cnt = m.counts()
    finally:
        m.close()
    assert res.tobytes() == res1.tobytes()
    assert np.array_equal(lab, lab1) and np.array_equal(seq, seq1)
    assert np.array_equal(cnt, cnt1)
