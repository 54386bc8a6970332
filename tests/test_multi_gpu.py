"""Several devices driven from one process (include/tagdust_multi.h).  A GPU box of this pool has one MI355X, so what can
run here is N = 1 (must equal the single-context path bit for bit), two contexts on the same device (the split, the
per-device windows of the artifact filter, the input-order merge and the counter sum) and the library's RCCL path on a
1-rank communicator (TD_MULTI_FORCE_RCCL=1: librccl.so loaded, ncclCommInitAll, ncclAllReduce on the device counters).
test_multi_distinct_devices runs wherever two GPUs are visible."""
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


def test_counts_through_rccl_on_a_one_rank_communicator(monkeypatch):
    """The collective path of the library itself on the one-GPU box: td_multi_create builds an RCCL communicator for the single
    device, td_multi_counts sums the device counters with ncclAllReduce into the second buffer and reads that back; the
    result equals td_counts_get of the context and serial counting, twice in a row (the running counters stay untouched)."""
    from tagdust_amd import lib as tdlib
    from tagdust_amd.lib import TagdustMulti
    import ctypes as C
    monkeypatch.setenv("TD_MULTI_FORCE_RCCL", "1")
    g = load_golden("c3_b6_s_r_p")
    m = TagdustMulti([0])
    try:
        assert m.uses_rccl()
        m.upload_model(g)
        m.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
        m.counts_reset()
        res, lab, seq = m.decode(g["seq"], g["offs"])
        cnt = m.counts()
        direct = np.zeros(tdlib.NUM_COUNTERS, np.int64)
        assert m.lib.td_counts_get(C.c_void_p(m.lib.td_multi_ctx(m.h, 0)), direct.ctypes.data) == 0
        assert np.array_equal(cnt, direct) and np.array_equal(cnt, tdlib.count_outcomes(res, g["lens"]))
        assert int(cnt[:8].sum()) == int((g["lens"] > 0).sum())
        res2, _, _ = m.decode(g["seq"], g["offs"])
        assert res2.tobytes() == res.tobytes()
        assert np.array_equal(m.counts(), 2 * cnt)
    finally:
        m.close()
    assert np.array_equal(res["read_type"], g["read_type"]) and np.array_equal(lab, g["labels"])


def _n_devices():
    import torch
    return torch.cuda.device_count()     # (does not initialise the GPU on this image)


@pytest.mark.skipif(_n_devices() < 2, reason="needs two visible GPUs (a GPU box of this pool has one)")
@pytest.mark.parametrize("name", ["c3_b6_s_r_p", "artifacts_b_r", "c2_indel_varlen"])
def test_multi_distinct_devices(name):
    """Two physical devices: RCCL communicator over both, one worker thread per device, DMA from one caller array by both --
    results and counters bit-equal to the single context."""
    from tagdust_amd.lib import TagdustMulti
    g = load_golden(name)
    res1, lab1, seq1, cnt1 = _single(g, g["seq"], g["offs"])
    m = TagdustMulti([0, 1])
    try:
        assert m.uses_rccl()
        art = golden_artifacts(g)
        if art:
            m.set_artifacts(art[0], art[1], art[2], art[3])
        m.upload_model(g)
        m.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
        m.counts_reset()
        res, lab, seq = m.decode(g["seq"], g["offs"])
        cnt = m.counts()
    finally:
        m.close()
    assert res.tobytes() == res1.tobytes()
    assert np.array_equal(lab, lab1) and np.array_equal(seq, seq1)
    assert np.array_equal(cnt, cnt1)


def test_bind_host_to_device_keeps_the_thread_runnable():
    """td_bind_host_to_device: the NUMA node next to GPU 0 (or -1 when the box does not say); whatever it answers, the thread's
    affinity mask is non-empty and inside the mask it had."""
    import os
    from tagdust_amd import lib as tdlib
    before = os.sched_getaffinity(0)
    try:
        node = tdlib.bind_host_to_device(0)
        after = os.sched_getaffinity(0)
        assert node >= -1 and len(after) >= 1 and after <= before
    finally:
        os.sched_setaffinity(0, before)
