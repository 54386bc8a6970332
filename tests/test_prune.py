"""Position pruning of the specialised kernel (td_spec_kernel.inc, "Position pruning").

CPU: the host's bound tables -- the only thing the kernel's exactness argument takes from outside -- dominate every DP value
the oracle (the pinned restatement of the reference) computes, on every fixture and on random architectures.
GPU: a batch decodes to the same bytes with pruning on and off, the pruned path is the one that ran, and every fall-back
route (failed check, spill cut too short) gives the same bytes again."""
import os

import numpy as np
import pytest

from conftest import load_golden, GOLDEN_NAMES


def _margins(md, seq, offs):
    from oracle import pyoracle
    from tagdust_amd import lib as tdlib
    lcap = int(np.diff(offs).max()) + 2
    info = tdlib.spec_prune_info(md, lcap)
    if info["n_seg"] == 0 and info["sfx_first"] == info["S"]:
        return info, None
    assert info["z"] > 103.98 + np.log(2.0)
    return info, pyoracle.bound_margins(pyoracle.OracleModel(md), seq, offs, info)


def _restart_margins(md, seq, offs):
    from oracle import pyoracle
    from tagdust_amd import lib as tdlib
    lcap = int(np.diff(offs).max()) + 2
    info = tdlib.spec_prune_info(md, lcap)
    if info["n_seg"] == 0 and info["sfx_first"] == info["S"]:
        return None
    gq, gf, _ = tdlib.spec_restart_info(md, lcap)
    return pyoracle.restart_margins(pyoracle.OracleModel(md), seq, offs, info["n_seg"], info["sfx_first"], gq, gf)


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_restart_bounds_dominate_reference_values(name):
    """Where a restarted sweep starts its upper ends -- max_k (impulse response[k] + the exact silent row k positions on) + log of
    the number of terms -- lies above every backward value of the leading and every forward value of the trailing segments, for
    every read and every position of every reference fixture.  (A start below a true value would make the interval close on a
    wrong number: this is the test the exactness of the restarts rests on.)"""
    g = load_golden(name)
    mg = _restart_margins(g, g["seq"], g["offs"])
    if mg is not None:
        assert mg.min() > 0.0, (name, mg)


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_bounds_dominate_reference_values(name):
    """fb[i] >= M/I_forward, bwb[len - i] >= M/I_backward of the pruned segments, wa[i] - 15.75 >= the read segment's entry
    term (leading segments), and the same with the exit term for the trailing segments, for every read of every reference
    fixture (the oracle's matrices are the reference's, test_oracle_golden)."""
    g = load_golden(name)
    info, mg = _margins(g, g["seq"], g["offs"])
    if name in ("casava_index", "o_b_s_r"):
        assert info["n_seg"] == 0     # a read segment first / an optional segment in front of it: nothing to prune in front
    else:
        assert info["n_seg"] >= 1, name
    if name in ("c3_b6_s_r_p", "c5_big_b96_f_r_p", "scen2_p_b_r_p"):
        assert info["sfx_first"] == info["S"] - 1      # the 3' adapter
    if mg is not None:
        assert mg.min() > 0.0, (name, mg)


@pytest.mark.parametrize("seed", range(12))
def test_bounds_dominate_on_random_architectures(seed):
    """The same on random segment lists (partial / barcode / fingerprint / spacer / G segments in front of the read
    segment), models from the library's own builder, reads with substitutions, indels, Ns and unrelated sequences."""
    from tagdust_amd import lib as tdlib
    from test_parity_gpu import _random_arch
    rng = np.random.RandomState(7000 + seed)
    segs, parts = _random_arch(rng)
    reads = []
    for i in range(150):
        s_ = "".join(p() for p in parts)
        out = []
        for ch in s_:
            u = rng.random_sample()
            if u < 0.03:
                out.append("ACGTN"[rng.randint(5)])
            elif u < 0.04:
                continue
            elif u < 0.05:
                out.append(ch); out.append("ACGT"[rng.randint(4)])
            else:
                out.append(ch)
        s_ = "".join(out)
        if rng.random_sample() < 0.15 or len(s_) < 12:
            s_ = "".join("ACGT"[x] for x in rng.randint(0, 4, rng.randint(12, 120)))
        reads.append(np.array(["ACGTN".index(ch) for ch in s_], np.uint8))
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    seq = np.concatenate(reads)
    md, _ = tdlib.build_model(segs, seq, offs, 0.05, 0.1)
    info, mg = _margins(md, seq, offs)
    if mg is not None:
        assert mg.min() > 0.0, (segs, mg)
    rm = _restart_margins(md, seq, offs)      # ... and so do the bounds the restarted sweeps start from
    if rm is not None:
        assert rm.min() > 0.0, (segs, rm)


def _long_reads(g, L, n, seed):
    """Reads of ~L bases for the config-3 architecture, the kinds a long-read bound must survive: the architecture's own shape
    with a long insert, inserts that take the read segment's maximum emission at every position (poly-base of the most frequent
    background letter) or are all N, unrelated reads, N-rich reads, and ragged lengths around L."""
    import bench
    rng = np.random.RandomState(seed)
    bg = np.asarray(g["bg"], np.float32)
    top = int(np.argmax(bg[:4]))
    bars = [[bench._CODE[c] for c in b] for b in bench.BARCODES]
    head = lambda: np.array(bars[rng.randint(len(bars))] + [bench._CODE[c] for c in bench.SPACER], np.uint8)
    ad = np.array([bench._CODE[c] for c in bench.ADAPTER], np.uint8)
    reads = []
    for k in range(n):
        ll = L if k % 3 else int(L - rng.randint(0, max(2, L // 10)))
        kind = k % 6
        ins = rng.randint(0, 4, ll).astype(np.uint8)
        if kind == 1:
            ins[:] = top
        elif kind == 2:
            ins[:] = 4
        elif kind == 3:
            ins[rng.random_sample(ll) < 0.2] = 4
        if kind != 4:                                 # (kind 4: an unrelated read)
            h = head()
            ins[:len(h)] = h
            keep = rng.randint(0, len(ad) + 1)
            if keep:
                ins[ll - keep:] = ad[:keep]
        reads.append(ins)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    return np.concatenate(reads), offs


@pytest.mark.parametrize("L", [1000, 3000, 8000])
def test_bounds_dominate_on_long_reads(L):
    """Scores grow by ~1.4 nats per base, and half a float spacing per addition with them (2.4e-4 at |score| = 4096): the
    bound tables carry a slack that grows with the magnitude of the bound (td_jit.hip row_slack / lse_up), so that they still
    dominate every value of the oracle's matrices on reads of thousands of bases -- maximum-emission inserts included, for
    which the bounds have no other slack.  Same for the starts of the restarted sweeps."""
    g = load_golden("c3_b6_s_r_p")
    seq, offs = _long_reads(g, L, 12, 4000 + L)
    info, mg = _margins(g, seq, offs)
    assert info["n_seg"] == 2 and info["sfx_first"] == 3
    assert mg.min() > 0.0, (L, mg)
    rm = _restart_margins(g, seq, offs)
    assert rm.min() > 0.0, (L, rm)


def test_model_section_states_the_pruned_segments():
    from tagdust_amd import lib as tdlib
    g = load_golden("c3_b6_s_r_p")
    src = tdlib.spec_source(g)
    assert "static constexpr int kPruneSegs = 2;" in src and "static constexpr int kSfxFirst = 3;" in src
    g = load_golden("casava_index")
    assert "static constexpr int kPruneSegs = 0;" in tdlib.spec_source(g)


# ---------------------------------------------------------------------------------------------------------------------
def _decode(workload, n, env, stats=True):
    """One resident batch of a bench workload through a fresh context compiled under `env`; (res, labels, seq, diagnostic words:
    td_diag_get, word k = historical slot 192 + k)."""
    import bench
    from tagdust_amd import TagdustHip
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        bench.select_workload(workload)
        model = bench.load_model()
        reads, offs = bench.synth_host_batch(n, 77)
        c = TagdustHip(0)
        try:
            c.set_option("poison_workspace", 1)
            c.upload_model(model)
            c.set_params(float(model["threshold"]), int(model["minlen"]), int(model["dust"]))
            c.upload_batch(reads, offs)
            c.counts_reset()
            c.run()
            res, labels, seq = c.download()
            return res, labels, seq, c.diag()
        finally:
            c.close()
    finally:
        bench.select_workload("c3")
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _same(a, b):
    return a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.gpu
@pytest.mark.parametrize("workload,n", [("c3", 1 << 16), ("c2", 1 << 16), ("c5", 1 << 14)], ids=["config3", "config2", "config5"])
def test_pruned_equals_dense_and_fallbacks(workload, n):
    """Same bytes with pruning off, on, and on with each fall-back route forced: the total_prob check failing for every tile
    (second, dense pass), the spill cut guessed too short (rows missing -> dense pass), a wave giving up after repeated
    failures.  The statistics counters (TD_SPEC_PRUNE_STATS) show which route ran."""
    S0 = 232 - 192    # diagnostic words: decisions, sum of required cuts, spill too short, failed checks, tiles, sum of cuts, dense tiles
    dense = _decode(workload, n, {"TD_SPEC_PRUNE": "0", "TD_SPEC_PRUNE_STATS": "0"})
    on = _decode(workload, n, {"TD_SPEC_PRUNE": "1", "TD_SPEC_PRUNE_STATS": "1"})
    assert _same(dense, on)
    st = on[3][S0:S0 + 8]
    tiles = (n + 63) // 64
    assert st[4] == tiles and st[6] == 0 and st[2] == 0 and st[3] == 0      # every tile pruned, no second pass
    assert st[5] / tiles < 0.5 * (100 if workload == "c2" else 150)          # mean cut well inside the read
    # every tile's total_prob check fails: second pass for the first tiles of a wave, then the wave stops trying
    forced = _decode(workload, n, {"TD_SPEC_PRUNE": "1", "TD_SPEC_PRUNE_STATS": "1", "TD_SPEC_EXTRA_OPTS": "-DTDS_PRUNE_TT=1000.0f"})
    assert _same(dense, forced)
    st = forced[3][S0:S0 + 8]
    assert st[3] > 0 and st[6] == st[4] == tiles
    # guessed far too optimistically: the exact cut needs rows the backward sweep did not spill (leading segments) / the exact
    # stop lies below the position the backward sweep of the trailing segments reached, or the read segment's check fails
    short = _decode(workload, n, {"TD_SPEC_PRUNE": "1", "TD_SPEC_PRUNE_STATS": "1", "TD_SPEC_EXTRA_OPTS": "-DTDS_PRUNE_BGAP=-150.0f -DTDS_PRUNE_MARGIN=0"})
    assert _same(dense, short)
    st = short[3][S0:S0 + 8]
    assert st[2] + short[3][226 - 192] > 0 and st[3] == 0 and st[6] == tiles
    # the trailing segments alone (config 3 and 5 end in a 3' adapter): pruned == dense, the stop lies well inside the read
    if workload != "c2":
        sfx = on[3][224 - 192:228 - 192]
        assert sfx[0] == tiles and sfx[2] == 0 and 30 < sfx[3] / tiles < 140
        only_sfx = _decode(workload, n, {"TD_SPEC_PRUNE": "1", "TD_SPEC_PRUNE_SFX": "0", "TD_SPEC_PRUNE_STATS": "0"})
        assert _same(dense, only_sfx)


@pytest.mark.gpu
def test_pruning_on_ragged_short_and_dead_reads():
    """Tiles whose reads end before, at and after the cut, reads too short for any path (decoded densely), N-rich reads:
    pruned and dense contexts agree, and both equal the oracle."""
    from oracle import pyoracle
    from tagdust_amd import TagdustHip
    g = load_golden("c3_b6_s_r_p")
    rng = np.random.RandomState(99)
    base = [g["seq"][g["offs"][i]:g["offs"][i + 1]] for i in range(len(g["offs"]) - 1)]
    reads = []
    for k in range(1500):
        r = base[rng.randint(len(base))].copy()
        L = int(rng.choice([12, 20, 30, 34, 35, 36, 37, 40, 60, 100, 150]))
        r = r[:L] if len(r) >= L else r
        if rng.random_sample() < 0.1:
            r[rng.randint(0, len(r), max(1, len(r) // 5))] = 4
        if rng.random_sample() < 0.1:
            r = rng.randint(0, 4, len(r)).astype(np.uint8)
        reads.append(r)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    seq = np.concatenate(reads)
    outs = []
    for prune in ("0", "1"):
        os.environ["TD_SPEC_PRUNE"] = prune
        try:
            c = TagdustHip(0)
            c.set_option("poison_workspace", 1)
            c.upload_model(g)
            c.set_params(float(g["threshold"]), int(g["minlen"]), int(g["dust"]))
            c.upload_batch(seq, offs)
            c.run()
            outs.append(c.download())
            c.close()
        finally:
            os.environ.pop("TD_SPEC_PRUNE", None)
    assert _same(outs[0], outs[1])
    ores, olab, oseq = pyoracle.label_batch(pyoracle.OracleModel(g), seq, offs, float(g["threshold"]), int(g["minlen"]), int(g["dust"]), 8)
    res, labels, seq_after = outs[1]
    for k in ("b_score", "f_score", "r_score", "bar_prob"):
        assert np.array_equal(res[k].view(np.uint32), ores[k].view(np.uint32)), k
    assert np.array_equal(labels, olab) and np.array_equal(seq_after, oseq)
    for k in ("read_type", "barcode", "fingerprint"):
        assert np.array_equal(res[k], ores[k]), k


_LONG_SEEDS = list(range(10)) + [455]     # 455: the architecture an alternative machine scheduler miscompiled (td_jit.hip, TD_SPEC_SCHED)
if os.environ.get("TD_FUZZ_SEEDS"):      # e.g. TD_FUZZ_SEEDS=100:400 for a longer one-off run
    _a, _b = os.environ["TD_FUZZ_SEEDS"].split(":")
    _LONG_SEEDS = list(range(int(_a), int(_b)))


@pytest.mark.gpu
@pytest.mark.parametrize("restart", [None, "1"], ids=["restart-by-size", "restart-forced"])
@pytest.mark.parametrize("seed", _LONG_SEEDS)
def test_random_architectures_with_long_inserts(seed, restart, monkeypatch):
    """Random segment lists as in test_parity_gpu, but with inserts of 120-220 bases: every read reaches far beyond the pruning
    cut and far in front of the trailing segments' stop.  Specialised kernel (pruning on) == oracle, bit for bit -- also with
    the restarted sweeps forced on for both ends (TDS_RESTART, TDS_RESTART_FWD: the interval bridges must close on the
    reference's floats or fail over to the plain sweep, whatever the segments look like)."""
    if restart:
        monkeypatch.setenv("TD_SPEC_RESTART", restart)
        monkeypatch.setenv("TD_SPEC_EXTRA_OPTS", "-DTDS_RESTART_FWD=1")
    from oracle import pyoracle
    from tagdust_amd import TagdustHip
    from tagdust_amd import lib as tdlib
    from test_parity_gpu import _random_arch
    rng = np.random.RandomState(9000 + seed)
    segs, parts = _random_arch(rng)
    parts[segs.index("R:N")] = lambda: "".join("ACGT"[x] for x in rng.randint(0, 4, rng.randint(120, 221)))
    reads = []
    for i in range(320):
        s_ = "".join(p() for p in parts)
        out = []
        for ch in s_:
            u = rng.random_sample()
            if u < 0.02:
                out.append("ACGT"[rng.randint(4)])
            elif u < 0.03:
                continue
            elif u < 0.04:
                out.append(ch); out.append("ACGT"[rng.randint(4)])
            elif u < 0.045:
                out.append("N")
            else:
                out.append(ch)
        s_ = "".join(out)
        if rng.random_sample() < 0.1:
            s_ = "".join("ACGT"[x] for x in rng.randint(0, 4, rng.randint(40, 240)))
        reads.append(np.array(["ACGTN".index(ch) for ch in s_], np.uint8))
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    seq = np.concatenate(reads)
    md, _ = tdlib.build_model(segs, seq, offs, 0.05, 0.1)
    thr = float(rng.choice([0.0, 1.5]))
    ores, olab, oseq = pyoracle.label_batch(pyoracle.OracleModel(md), seq, offs, thr, 16, 100, 8)
    c = TagdustHip(0)
    try:
        c.set_option("poison_workspace", 1)
        c.upload_model(md)
        c.set_params(thr, 16, 100)
        c.upload_batch(seq, offs)
        c.run()
        res, labels, seq_after = c.download()
    finally:
        c.close()
    for k in ("b_score", "f_score", "r_score", "bar_prob"):
        assert np.array_equal(res[k].view(np.uint32), ores[k].view(np.uint32)), (k, segs)
    assert np.array_equal(labels, olab), segs
    assert np.allclose(res["mapq"], ores["Q"], rtol=0, atol=1e-4), segs
    for k in ("read_type", "barcode", "fingerprint"):
        assert np.array_equal(res[k], ores[k]), (k, segs)
    assert np.array_equal(seq_after, oseq), segs


@pytest.mark.gpu
def test_reads_beyond_the_bound_tables_decode_densely():
    """Reads of more than 8192 bases: the host does not run the bound recurrences that far, the tables stay zero, which the
    kernel reads as "nothing can be pruned"; a mixed batch (a few 9000-base reads among 150-base ones) equals the oracle."""
    from oracle import pyoracle
    from tagdust_amd import TagdustHip
    g = load_golden("c2_b4_r")
    rng = np.random.RandomState(17)
    src = g["offs"]
    reads = []
    for i in range(96):
        L = 9000 if i % 24 == 5 else int(rng.randint(100, 151))
        r = rng.randint(0, 4, L).astype(np.uint8)
        r[:4] = g["seq"][src[i]:src[i] + 4]       # a real barcode in front
        reads.append(r)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    seq = np.concatenate(reads)
    ores, olab, oseq = pyoracle.label_batch(pyoracle.OracleModel(g), seq, offs, float(g["threshold"]), 16, 100, 8)
    c = TagdustHip(0)
    try:
        c.set_option("poison_workspace", 1)
        c.upload_model(g)
        c.set_params(float(g["threshold"]), 16, 100)
        c.upload_batch(seq, offs)
        c.run()
        res, labels, seq_after = c.download()
    finally:
        c.close()
    for k in ("b_score", "f_score", "r_score", "bar_prob"):
        assert np.array_equal(res[k].view(np.uint32), ores[k].view(np.uint32)), k
    assert np.array_equal(labels, olab) and np.array_equal(seq_after, oseq)
    assert np.array_equal(res["read_type"], ores["read_type"])


@pytest.mark.gpu
@pytest.mark.parametrize("workload,n", [("c3", 1 << 15), ("c2", 1 << 15), ("c5", 1 << 13)], ids=["config3", "config2", "config5"])
def test_restarted_sweeps_are_exact_or_fall_back(workload, n):
    """TDS_RESTART (td_spec_kernel.inc "Restarted sweeps"): the leading segments' backward sweep and -- with TDS_RESTART_FWD -- the
    trailing segments' forward sweep start W positions before the first position their values are used at, from an interval
    whose upper end is the host's relative bound, and hand over to the plain sweep once every interval has closed.  Forced on
    with W = 40 every bridge closes and the outputs are those of the dense sweeps, bit for bit; with W = 2 none can close, every
    tile is decoded again without the restarts (pruned as before), and the outputs are still the same.  Configs 3 and 5 restart by default (48 or more leading columns)."""
    dense = _decode(workload, n, {"TD_SPEC_PRUNE": "1", "TD_SPEC_RESTART": "0", "TD_SPEC_PRUNE_STATS": "0"})
    on = _decode(workload, n, {"TD_SPEC_PRUNE": "1", "TD_SPEC_RESTART": "1", "TD_SPEC_PRUNE_STATS": "1", "TD_SPEC_EXTRA_OPTS": "-DTDS_RESTART_W=40 -DTDS_RESTART_FWD=1"})
    assert _same(dense, on)
    d = on[3]
    tiles = (n + 63) // 64
    assert d[206 - 192] == tiles and d[207 - 192] == 0                     # backward: restarted in every tile, none failed
    assert d[197 - 192] > 0 and 3 < d[196 - 192] / d[197 - 192] < 25       # bridges closed well inside W
    if workload != "c2":                                                   # (config 2 has no segment behind its read segment)
        assert d[198 - 192] == tiles and d[199 - 192] == 0                 # forward likewise
    assert d[238 - 192] == 0                                               # no dense second pass
    short = _decode(workload, n, {"TD_SPEC_PRUNE": "1", "TD_SPEC_RESTART": "1", "TD_SPEC_PRUNE_STATS": "1", "TD_SPEC_EXTRA_OPTS": "-DTDS_RESTART_W=2 -DTDS_RESTART_WMAX=2"})
    assert _same(dense, short)
    d = short[3]
    # every restarted tile failed to close and was decoded again without the restarts -- still pruned, not densely
    assert d[207 - 192] == tiles and d[238 - 192] == 0 and d[237 - 192] / tiles < 0.5 * (100 if workload == "c2" else 150)
    # a wave whose bridges run out of positions widens their window before it gives the restarts up: from W = 2 every wave's
    # first tile fails once or twice, the bridges then close, and no tile is decoded densely
    widen = _decode(workload, n, {"TD_SPEC_PRUNE": "1", "TD_SPEC_RESTART": "1", "TD_SPEC_PRUNE_STATS": "1", "TD_SPEC_EXTRA_OPTS": "-DTDS_RESTART_W=2"})
    assert _same(dense, widen)
    d = widen[3]
    assert d[207 - 192] >= 1 and d[206 - 192] > d[207 - 192] and d[238 - 192] == 0
    default = _decode(workload, n, {"TD_SPEC_PRUNE": "1", "TD_SPEC_PRUNE_STATS": "1"})
    assert _same(dense, default)
    d = default[3]
    if workload != "c2":
        assert d[206 - 192] == tiles and d[207 - 192] == 0 and d[238 - 192] == 0     # on by default, every bridge closes
    else:
        assert d[206 - 192] == 0                                                      # off for a handful of short HMMs (32 columns)


@pytest.mark.gpu
@pytest.mark.parametrize("L", [1000, 3000, 8000])
def test_long_reads_with_pruning_and_restarts_equal_the_oracle(L):
    """Config 3's architecture on reads of 1000 / 3000 / 8000 bases with pruning and the restarted sweeps live (the bound
    tables reach 8192 bases): maximum-emission (poly-base), all-N and N-rich inserts, unrelated reads, ragged lengths.  HIP ==
    oracle bit for bit, and the statistics show that the pruned path (not a dense second pass) produced it."""
    from oracle import pyoracle
    from tagdust_amd import TagdustHip
    g = load_golden("c3_b6_s_r_p")
    n = 192 if L <= 3000 else 128
    seq, offs = _long_reads(g, L, n, 5000 + L)
    thr, minlen, dust = float(g["threshold"]), int(g["minlen"]), int(g["dust"])
    ores, olab, oseq = pyoracle.label_batch(pyoracle.OracleModel(g), seq, offs, thr, minlen, dust, 8)
    old = {k: os.environ.get(k) for k in ("TD_SPEC_PRUNE_STATS", "TD_NO_LENGTH_CLASSES")}
    os.environ["TD_SPEC_PRUNE_STATS"] = "1"
    try:
        c = TagdustHip(0)
        try:
            c.set_option("poison_workspace", 1)
            c.upload_model(g)
            c.set_params(thr, minlen, dust)
            c.upload_batch(seq, offs)
            assert c.get_option("prune_active") == 1
            c.counts_reset()
            c.run()
            res, labels, seq_after = c.download()
            d = c.diag()
        finally:
            c.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for k in ("b_score", "f_score", "r_score", "bar_prob"):
        assert np.array_equal(res[k].view(np.uint32), ores[k].view(np.uint32)), (k, L)
    assert np.array_equal(labels, olab) and np.array_equal(seq_after, oseq)
    assert np.allclose(res["mapq"], ores["Q"], rtol=0, atol=1e-4)
    for k in ("read_type", "barcode", "fingerprint"):
        assert np.array_equal(res[k], ores[k]), (k, L)
    tiles = (n + 63) // 64
    assert d[236 - 192] == tiles                       # every tile took the pruning decision ...
    assert d[237 - 192] / tiles < 0.5 * L              # ... and the cut lies well inside the reads
