"""Pipelined batches (td_submit / td_wait): several batches in flight, copies on their own streams, device-side ingest and
egress.  Results must equal the reference fixtures and the synchronous path bit for bit, for pageable and page-locked
buffers, ragged and uniform batches, base codes and FASTQ text."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _ctx(g, specialize=1, depth=2):
    from tagdust_amd import TagdustHip
    c = TagdustHip(0)
    c.set_option("specialize", specialize)
    c.set_option("pipeline_depth", depth)
    c.set_option("poison_workspace", 1)
    c.upload_model(g)
    c.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
    return c


def _check_against_fixture(g, res, labels, seq, idx=None):
    sel = (lambda a: a) if idx is None else (lambda a: a[idx])
    for k in ("b_score", "f_score", "r_score"):
        assert np.array_equal(_bits(res[k]), _bits(sel(g[k]))), k
    assert np.array_equal(_bits(res["bar_prob"]), _bits(sel(g["bar_prob"]).astype(np.float32)))
    assert np.allclose(res["mapq"], sel(g["mapq"]), rtol=0, atol=1e-4)
    for k in ("read_type", "barcode", "fingerprint"):
        assert np.array_equal(res[k], sel(g[k])), k


@pytest.mark.parametrize("specialize", [1, 0], ids=["specialised", "generic"])
@pytest.mark.parametrize("name", ["c3_b6_s_r_p", "c2_indel_varlen", "umi_f_s_r"])
def test_submit_wait_equals_fixture(name, specialize):
    from tagdust_amd import RESULT_DTYPE
    g = load_golden(name)
    c = _ctx(g, specialize)
    try:
        n = int(g["n_reads"])
        offs = np.ascontiguousarray(g["offs"], np.int64)
        seq = np.ascontiguousarray(g["seq"], np.uint8)
        res = np.zeros(n, RESULT_DTYPE)
        lab = np.zeros(int(offs[-1]) + n, np.int8)
        sq = np.zeros(int(offs[-1]), np.uint8)
        t = c.submit(seq, offs, res=res, labels=lab, seq_out=sq)
        c.wait(t)
        _check_against_fixture(g, res, lab, sq)
        assert np.array_equal(lab, g["labels"]) and np.array_equal(sq, g["seq_after"])
    finally:
        c.close()


def test_many_batches_in_flight_match_synchronous_path():
    """Eight batches of different sizes and length mixes through a depth-3 pipeline, pageable and page-locked buffers
    alternating, FASTQ text and base codes alternating; every batch equals what the synchronous calls give for it."""
    from tagdust_amd import RESULT_DTYPE, TdError
    from tagdust_amd.lib import PinnedArray
    g = load_golden("c2_b4_r")
    rng = np.random.RandomState(11)
    src_offs = g["offs"]
    batches = []
    for b in range(8):
        n = int(rng.choice([1, 63, 64, 65, 500, 3000, 20000]))
        uniform = b % 3 == 0
        lens = np.full(n, 100) if uniform else rng.randint(20, 101, n)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        codes = rng.randint(0, 4, int(offs[-1])).astype(np.uint8)
        codes[rng.random_sample(len(codes)) < 0.004] = 4
        for i in range(0, n, 2):
            k = i % int(g["n_reads"])
            s_ = g["seq"][src_offs[k]:src_offs[k + 1]][:lens[i]]
            codes[offs[i]:offs[i] + len(s_)] = s_
        batches.append((codes, offs))
    c = _ctx(g, 1, depth=3)
    try:
        # synchronous reference results
        want = []
        for codes, offs in batches:
            c.upload_batch(codes, offs)
            c.run()
            want.append(c.download())
        c.counts_reset()
        keep, tickets, got = [], [], []
        for b, (codes, offs) in enumerate(batches):
            n = len(offs) - 1
            ascii_ = b % 2 == 1
            data = np.frombuffer(b"ACGTN", np.uint8)[codes] if ascii_ else codes
            if b % 4 < 2:      # page-locked in and out
                pin = [PinnedArray(data.shape, np.uint8), PinnedArray((n,), RESULT_DTYPE), PinnedArray((int(offs[-1]) + n,), np.int8),
                       PinnedArray((int(offs[-1]),), np.uint8)]
                pin[0].array[:] = data
                keep.append(pin)
                data, res, lab, sq = (p.array for p in pin)
            else:
                res, lab, sq = np.zeros(n, RESULT_DTYPE), np.zeros(int(offs[-1]) + n, np.int8), np.zeros(int(offs[-1]), np.uint8)
            if len(tickets) == 3:          # the pipeline is full: a fourth submit must be refused, not block or overwrite
                with pytest.raises(TdError, match="pipeline slots"):
                    c.submit(data, offs, res=res, labels=lab, seq_out=sq, ascii=ascii_)
                c.wait(tickets.pop(0))
            tickets.append(c.submit(data, offs, res=res, labels=lab, seq_out=sq, ascii=ascii_))
            keep.append((data, offs))
            got.append((res, lab, sq))
        for t in tickets:
            c.wait(t)
        for b, ((res, lab, sq), (wres, wlab, wsq)) in enumerate(zip(got, want)):
            assert res.tobytes() == wres.tobytes(), b
            assert np.array_equal(lab, wlab), b
            assert np.array_equal(sq, wsq), b
        cnt = c.counts()
        assert int(cnt[:8].sum()) == sum(len(o) - 1 for _, o in batches)
        with pytest.raises(TdError, match="no batch with ticket"):
            c.wait(12345)
    finally:
        c.close()
        for pin in keep:
            if isinstance(pin, list):
                for p in pin:
                    p.free()


def test_submit_prob_mode_and_partial_outputs():
    """TD_MODE_GET_PROB through the pipeline (what threshold calibration uses), records only."""
    from tagdust_amd import RESULT_DTYPE, MODE_GET_PROB
    g = load_golden("scen2_p_b_r_p")
    c = _ctx(g)
    try:
        n = int(g["n_reads"])
        offs = np.ascontiguousarray(g["offs"], np.int64)
        seq = np.ascontiguousarray(g["seq"], np.uint8)
        res = np.zeros(n, RESULT_DTYPE)
        c.wait(c.submit(seq, offs, mode=MODE_GET_PROB, res=res))
        assert np.array_equal(_bits(res["f_score"]), _bits(g["f_score"]))
        assert np.allclose(res["mapq"], g["mapq"], rtol=0, atol=1e-4)
    finally:
        c.close()


def test_new_entry_points_report_misuse():
    """TD_FAIL with a message, never a crash: pipelined calls without a model, bad option / window values, a device that
    does not exist for the multi-device driver."""
    from tagdust_amd import TagdustHip, TdError, RESULT_DTYPE
    from tagdust_amd.lib import TagdustMulti
    g = load_golden("c2_b4_r")
    c = TagdustHip(0)
    try:
        offs = np.ascontiguousarray(g["offs"], np.int64)
        seq = np.ascontiguousarray(g["seq"], np.uint8)
        res = np.zeros(int(g["n_reads"]), RESULT_DTYPE)
        with pytest.raises(TdError, match="no model"):
            c.submit(seq, offs, res=res)
        with pytest.raises(TdError, match="pipeline_depth"):
            c.set_option("pipeline_depth", 9)
        with pytest.raises(TdError, match="matchstart"):
            c.set_window(10, 5)
        c.upload_model(g)
        c.set_params(float(g["threshold"]), 16, 100)
        with pytest.raises(TdError, match="unsupported mode"):
            c.submit(seq, offs, mode=3, res=res)
        bad = offs.copy()
        bad[5] = bad[4] - 1                       # a negative read length
        with pytest.raises(TdError, match="has length"):
            c.submit(seq, bad, res=res)
        t = c.submit(seq, offs, res=res)          # the context still works afterwards
        c.wait(t)
        assert np.array_equal(res["read_type"], g["read_type"])
        # a rejected model upload (a ticket is out / an invalid description) leaves the context as it was
        t3 = c.submit(seq, offs, res=res)
        with pytest.raises(TdError, match="tickets are outstanding"):
            c.upload_model(g)
        c.wait(t3)
        assert c.last_kernel_ms() >= 0.0          # the batch in flight kept its record
        badm = dict(g)
        badm["H"] = int(g["H"]) + 1               # sizes that do not add up
        badm["label"] = np.concatenate([g["label"], g["label"][:1]])
        badm["A"] = np.eye(int(g["H"]) + 1, dtype=np.float32)
        with pytest.raises(TdError, match="inconsistent sizes"):
            c.upload_model(badm)
        res[:] = 0
        c.wait(c.submit(seq, offs, res=res))      # still the old model, no re-upload needed
        assert np.array_equal(res["read_type"], g["read_type"])
        with pytest.raises(TdError, match="host_threads"):
            c.set_option("host_threads", 0)
        c.set_option("host_threads", 3)
        assert c.get_option("host_threads") == 3
        with pytest.raises(TdError, match="tickets are outstanding|no batch with ticket"):
            t2 = c.submit(seq, offs, res=res)
            try:
                c.upload_batch(seq, offs)         # synchronous calls are refused while a ticket is out
            finally:
                c.wait(t2)
            c.wait(t2)                            # waiting twice for one ticket is an error too
    finally:
        c.close()
    with pytest.raises(TdError, match="device"):
        TagdustMulti([0, 99])


@pytest.mark.parametrize("overlap", [1, 0])
def test_overlapping_launches_equal_single_stream(overlap):
    """Consecutive pipelined batches alternate between two compute streams / workspaces ("overlap_decode"): eleven batches of
    different sizes through a depth-4 pipeline, with a model re-upload and a counter reset in between, give the bytes the
    synchronous path gives, and the counters add up over both streams."""
    from tagdust_amd import RESULT_DTYPE
    g = load_golden("c3_b6_s_r_p")
    rng = np.random.RandomState(31 + overlap)
    n0 = int(g["n_reads"])
    reads = [np.asarray(g["seq"][g["offs"][i]:g["offs"][i + 1]], np.uint8) for i in range(n0)]
    c = _ctx(g, 1, depth=4)
    ref = _ctx(g, 1, depth=1)
    try:
        c.set_option("overlap_decode", overlap)
        assert c.get_option("overlap_decode") == overlap
        total = 0
        for rnd in range(2):
            if rnd == 1:
                c.upload_model(g)               # waits for both streams, batches are staged per model
                c.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
            c.counts_reset()
            batches, tickets, outs = [], [], []
            for b in range(11):
                pick = rng.randint(0, n0, int(rng.choice([1, 63, 64, 65, 700, 3000])))
                seq = np.concatenate([reads[i] for i in pick])
                offs = np.concatenate([[0], np.cumsum([len(reads[i]) for i in pick])]).astype(np.int64)
                res = np.zeros(len(pick), RESULT_DTYPE)
                lab = np.zeros(int(offs[-1]) + len(pick), np.int8)
                sq = np.zeros(int(offs[-1]), np.uint8)
                batches.append((seq, offs)); outs.append((res, lab, sq))
                tickets.append(c.submit(seq, offs, res=res, labels=lab, seq_out=sq))
                if len(tickets) - sum(t is None for t in tickets) >= 4:
                    k = next(i for i, t in enumerate(tickets) if t is not None)
                    c.wait(tickets[k]); tickets[k] = None
            for k, t in enumerate(tickets):
                if t is not None:
                    c.wait(t)
            cnt = c.counts()
            total = sum(len(o[0]) for o in outs)
            assert int(cnt[:8].sum()) == total
            for (seq, offs), (res, lab, sq) in zip(batches, outs):
                ref.upload_batch(seq, offs)
                ref.run()
                r2, l2, s2 = ref.download()
                assert res.tobytes() == r2.tobytes() and np.array_equal(lab, l2) and np.array_equal(sq, s2)
    finally:
        c.close()
        ref.close()


def test_page_locked_input_may_be_reused_after_submit():
    """td_submit "returns once the reads have left the caller's buffers" -- also when the buffer is page-locked and the DMA
    engine reads it directly: the producer overwrites its buffer right after every submit, the results are those of the
    reads it handed over."""
    from tagdust_amd import RESULT_DTYPE
    from tagdust_amd.lib import PinnedArray
    import bench
    bench.select_workload("c3")
    g = bench.load_model()
    n, L = 1 << 18, bench.READ_LEN
    batches = [bench.synth_batch(n, 4242 + k).reshape(-1) for k in range(3)]
    offs = np.arange(n + 1, dtype=np.int64) * L
    c = _ctx(g, 1, depth=3)
    ref = _ctx(g, 1, depth=1)
    pin = PinnedArray((n * L,), np.uint8)
    try:
        outs, tickets = [], []
        for b in batches:
            pin.array[:] = b
            res = np.zeros(n, RESULT_DTYPE)
            sq = np.zeros(n * L, np.uint8)
            outs.append((res, sq))
            tickets.append(c.submit(pin.array, offs, res=res, seq_out=sq))
            pin.array[:] = 4                      # the producer refills its buffer: all N
        for t in tickets:
            c.wait(t)
        for b, (res, sq) in zip(batches, outs):
            ref.upload_batch(b, offs)
            ref.run()
            r2, _, s2 = ref.download(labels=False)
            assert res.tobytes() == r2.tobytes() and np.array_equal(sq, s2)
    finally:
        c.close()
        ref.close()
        pin.free()


def test_stable_page_locked_input_keeps_the_compact_egress():
    """Option "stable_input": the caller leaves its page-locked input alone until td_wait -- td_submit then neither waits for the
    upload nor falls back to full device-side copies of the rewritten sequences (the keep bits are applied to the caller's own
    buffer).  Same bytes as the synchronous path, for base codes and for sequence text, labels included; and without the promise
    (default) a refilled buffer still gives the right bytes (test above)."""
    from tagdust_amd import RESULT_DTYPE
    from tagdust_amd.lib import PinnedArray
    import bench
    bench.select_workload("c3")
    g = bench.load_model()
    n, L = 1 << 17, bench.READ_LEN
    offs = np.arange(n + 1, dtype=np.int64) * L
    c = _ctx(g, 1, depth=3)
    ref = _ctx(g, 1, depth=1)
    pins = [PinnedArray((n * L,), np.uint8) for _ in range(3)]
    try:
        c.set_option("stable_input", 1)
        assert c.get_option("stable_input") == 1
        for ascii_ in (False, True):
            batches = [bench.synth_batch(n, 777 + k).reshape(-1) for k in range(3)]
            outs, tickets = [], []
            for b, pin in zip(batches, pins):
                pin.array[:] = np.frombuffer(b"ACGTN", np.uint8)[b] if ascii_ else b
                res = np.zeros(n, RESULT_DTYPE); sq = np.full(n * L, 99, np.uint8); lab = np.full(n * (L + 1), 99, np.int8)
                outs.append((res, sq, lab))
                tickets.append(c.submit(pin.array, offs, res=res, seq_out=sq, labels=lab, ascii=ascii_))
            for t in tickets:
                c.wait(t)
            for b, (res, sq, lab) in zip(batches, outs):
                ref.upload_batch(b, offs)
                ref.run()
                r2, l2, s2 = ref.download()
                assert res.tobytes() == r2.tobytes() and np.array_equal(sq, s2) and np.array_equal(lab, l2)
    finally:
        c.close()
        ref.close()
        for p_ in pins:
            p_.free()


def test_length_outliers_get_their_own_wave_slots(monkeypatch):
    """A batch of 150-base reads with 0.1 % reads of 1000 bases: the workspace keeps the geometry of the many (the long tiles get
    wave slots of their own), both pipeline streams stay in use, and every output byte equals what one geometry for all gives
    (TD_NO_LENGTH_CLASSES=1) -- and the oracle, on a sample that holds every long read."""
    import bench
    from oracle import pyoracle
    from tagdust_amd import TagdustHip, RESULT_DTYPE
    bench.select_workload("c3")
    g = bench.load_model()
    rng = np.random.default_rng(77)
    n, L = 1 << 17, bench.READ_LEN
    short = bench.synth_batch(n, 99)
    n_long = n // 1000
    at = set(rng.choice(n, n_long, replace=False).tolist())
    reads = []
    for i in range(n):
        if i in at:                     # the same kind of read with a 1000-base insert
            r = np.concatenate([short[i][:9], rng.integers(0, 4, 1000 - L, dtype=np.uint8), short[i][9:]])
            reads.append(r)
        else:
            reads.append(short[i])
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    seq = np.concatenate(reads).astype(np.uint8)

    def run(no_classes):
        if no_classes:
            monkeypatch.setenv("TD_NO_LENGTH_CLASSES", "1")
        else:
            monkeypatch.delenv("TD_NO_LENGTH_CLASSES", raising=False)
        c = _ctx(g, 1, depth=3)
        try:
            outs = []
            for _ in range(2):          # two batches: both streams / workspaces
                res = np.zeros(n, RESULT_DTYPE)
                lab = np.zeros(int(offs[-1]) + n, np.int8)
                sq = np.zeros(int(offs[-1]), np.uint8)
                outs.append((res, lab, sq, c.submit(seq, offs, res=res, labels=lab, seq_out=sq)))
            for o in outs:
                c.wait(o[3])
            return outs, c.get_option("length_classes"), c.batch_info()[1], c.get_option("overlap_active")
        finally:
            c.close()

    one, k1, ws1, ov1 = run(True)
    two, k2, ws2, ov2 = run(False)
    assert k1 == 0 and 1 <= k2 <= 4 * (n_long + 2) and ov2 == 1
    assert ws2 * 3 < ws1                                     # (1000-base geometry for every slot: 6.6 x the memory)
    for a, b in zip(one, two):
        assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert one[0][0].tobytes() == one[1][0].tobytes()
    pick = np.sort(np.concatenate([np.fromiter(at, np.int64), rng.choice(n, 400, replace=False)]))
    pick = np.unique(pick)
    pseq = np.concatenate([reads[i] for i in pick]).astype(np.uint8)
    poffs = np.concatenate([[0], np.cumsum([len(reads[i]) for i in pick])]).astype(np.int64)
    om = pyoracle.OracleModel(g)
    ores, olab, oseq = pyoracle.label_batch(om, pseq, poffs, float(g["threshold"]), int(g["minlen"]), int(g["dust"]), 8)
    res = two[0][0][pick]
    for k, ok_ in (("b_score", "b_score"), ("f_score", "f_score"), ("r_score", "r_score"), ("bar_prob", "bar_prob"), ("mapq", "Q")):
        assert np.array_equal(_bits(res[k]), _bits(ores[ok_])), k
    for k in ("read_type", "barcode", "fingerprint"):
        assert np.array_equal(res[k], ores[k])
    lab = np.concatenate([two[0][1][offs[i] + i: offs[i + 1] + i + 1] for i in pick])
    sq = np.concatenate([two[0][2][offs[i]: offs[i + 1]] for i in pick])
    assert np.array_equal(lab, olab) and np.array_equal(sq, oseq)


@pytest.mark.parametrize("name", ["c3_b6_s_r_p", "c2_indel_varlen", "window_b_r", "b_r_s_r"])
def test_compact_egress_equals_plain_copies(name, monkeypatch):
    """The rewritten sequences and the labels come back as keep bits and label runs and are rebuilt on the host (td_api.hip
    "Compact egress"); with the plain device-side copies (TD_COMPACT_EGRESS=0), with a run table too small for any read
    (TD_RLE_CAP=1: every batch takes the fall-back to the labels as they are) and with page-locked input (the keep bits cannot be
    used: the caller may have refilled the buffer) the bytes are the same, for base codes and for sequence text."""
    from tagdust_amd import RESULT_DTYPE
    from tagdust_amd.lib import PinnedArray
    from conftest import golden_window
    g = load_golden(name)
    n = int(g["n_reads"])
    offs = np.ascontiguousarray(g["offs"], np.int64)
    seq = np.ascontiguousarray(g["seq"], np.uint8)
    text = np.frombuffer(b"ACGTN", np.uint8)[seq]
    outs = {}
    for label, env, ascii_, pinned in (("compact", {}, False, False), ("plain", {"TD_COMPACT_EGRESS": "0"}, False, False),
                                       ("overflow", {"TD_RLE_CAP": "1"}, False, False), ("text", {}, True, False), ("pinned", {}, False, True)):
        for k in ("TD_COMPACT_EGRESS", "TD_RLE_CAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = _ctx(g, 1)
        pin = PinnedArray((len(seq),), np.uint8) if pinned else None
        try:
            c.set_window(*(golden_window(g) or (-1, -1)))
            src = text if ascii_ else seq
            if pin is not None:
                pin.array[:] = src
                src = pin.array
            res = np.zeros(n, RESULT_DTYPE); lab = np.full(int(offs[-1]) + n, 99, np.int8); sq = np.full(int(offs[-1]), 99, np.uint8)
            c.wait(c.submit(src, offs, res=res, labels=lab, seq_out=sq, ascii=ascii_))
            outs[label] = (res.tobytes(), lab.copy(), sq.copy())
            # the synchronous calls go the same way
            if ascii_:
                c.upload_batch_ascii(text, offs)
            else:
                c.upload_batch(seq, offs)
            c.run()
            r2, l2, s2 = c.download()
            assert r2.tobytes() == outs[label][0] and np.array_equal(l2, lab) and np.array_equal(s2, sq)
        finally:
            c.close()
            if pin is not None:
                pin.free()
    assert np.array_equal(outs["compact"][1], g["labels"]) and np.array_equal(outs["compact"][2], g["seq_after"])
    for k in ("plain", "overflow", "text", "pinned"):
        assert outs[k][0] == outs["compact"][0] and np.array_equal(outs[k][1], outs["compact"][1]) and np.array_equal(outs[k][2], outs["compact"][2]), k


@pytest.mark.gpu
def test_generic_kernel_after_specialised_kernel_in_one_context():
    """One context, one slot: a batch through the model-specialised kernel (which leaves the label runs of the compact egress
    itself), then -- "specialize" switched off, model uploaded again -- a batch of three times as many reads through the generic
    kernel, whose labels the finish kernel has to scan.  The second batch must not be handed the first one's runs (their table
    is sized for the first batch), through the synchronous calls and through the pipelined ones."""
    from tagdust_amd import RESULT_DTYPE
    g = load_golden("c2_b4_r")
    n = int(g["n_reads"])
    offs1 = np.ascontiguousarray(g["offs"], np.int64)
    seq1 = np.ascontiguousarray(g["seq"], np.uint8)
    reps = 3
    seq3 = np.tile(seq1, reps)
    offs3 = np.concatenate([[0], np.cumsum(np.tile(np.diff(offs1), reps))]).astype(np.int64)
    lab3 = np.tile(g["labels"], reps)
    c = _ctx(g, 1, depth=1)
    try:
        for pipelined in (False, True):
            c.set_option("specialize", 1); c.upload_model(g); c.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
            res = np.zeros(n, RESULT_DTYPE); lab = np.full(int(offs1[-1]) + n, 99, np.int8); sq = np.full(int(offs1[-1]), 99, np.uint8)
            c.wait(c.submit(seq1, offs1, res=res, labels=lab, seq_out=sq))
            assert np.array_equal(lab, g["labels"])
            c.set_option("specialize", 0); c.upload_model(g); c.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
            if pipelined:
                res = np.zeros(n * reps, RESULT_DTYPE); lab = np.full(int(offs3[-1]) + n * reps, 99, np.int8); sq = np.full(int(offs3[-1]), 99, np.uint8)
                c.wait(c.submit(seq3, offs3, res=res, labels=lab, seq_out=sq))
            else:
                c.upload_batch(seq3, offs3)
                c.run()
                res, lab, sq = c.download()
            assert np.array_equal(lab, lab3) and np.array_equal(sq, np.tile(g["seq_after"], reps))
    finally:
        c.close()
