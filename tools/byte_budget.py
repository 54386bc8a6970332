#!/usr/bin/env python3
"""Per-array HBM byte budget of the specialised decode kernel for one read of a bench workload (round 4): what
each workspace array is written and read per read, from the kernel's own constants (the model section the kernel is compiled
from: group sizes, stored half-columns, pure / dropped columns) and the positions the launch actually covers (mean pruning
cut, spill cut, trailing stop, bridge window -- from TD_SPEC_PRUNE_STATS, e.g. tools/cut_probe.py), summed and held against the
PMC total of profiles/traffic.json when that record belongs to this kernel source.
usage: tools/byte_budget.py [c3|c2|c5] [cut] [spill cut] [stop] [spill stop] [W]"""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                                    # noqa: E402
from tagdust_amd import lib as tdlib            # noqa: E402

DEFAULTS = {"c3": (31, 37, 110, 107, 20), "c2": (22, 26, 0, 0, 0), "c5": (37, 42, 111, 108, 20)}


def ints(src, name):
    m = re.search(r"static constexpr int %s\[\d+\] = \{([^}]*)\}" % name, src)
    return [int(x) for x in m.group(1).split(",")]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    cut, cutS, stop, stopS, W = [int(x) for x in sys.argv[2:7]] if len(sys.argv) > 6 else DEFAULTS[wl]
    bench.select_workload(wl)
    src = tdlib.spec_source(bench.load_model())
    L = bench.READ_LEN
    nh, nc, grp, grpf = ints(src, "kNHmm"), ints(src, "kNCol"), ints(src, "kGroup"), ints(src, "kGroupF")
    pure, dropm = ints(src, "kPureLast"), ints(src, "kDropM")
    S = len(nh)
    H = sum(nh)
    ps = int(re.search(r"kPruneSegs = (\d+)", src).group(1))
    sf = int(re.search(r"kSfxFirst = (\d+)", src).group(1))
    first_n = int(re.search(r"kFirstN = (\d+)", src).group(1))
    restart = int(re.search(r"#define TDS_RESTART (\d)", src).group(1))
    halves = [2 * (nc[j] - pure[j]) - dropm[j] for j in range(S)]
    rows = []     # (array, what, bytes written, bytes read)

    def add(arr, what, w, r):
        rows.append((arr, what, float(w), float(r)))

    nw2, nw1 = (L + 15) // 16, (L + 31) // 32
    add("packed", "2-bit codes + N mask in", 0, 4 * (nw2 + nw1))
    add("codes", "unpacked base codes (one byte a position)", L + 2, 0)
    groups_b = [-(-nh[j] // grp[j]) for j in range(S)]
    groups_f = [-(-nh[j] // grpf[j]) for j in range(S)]
    for j in range(S - 1, -1, -1):      # backward sweeps
        lead, trail = j < ps, j >= sf
        if trail:
            npos_sweep, npos_spill = L - stopS + 1, L - stopS + 1
        elif lead:
            npos_sweep = (cutS + j * (W + 1)) if restart else L
            npos_spill = cutS
        else:
            npos_sweep, npos_spill = L, L
        gb = groups_b[j]
        add("codes", "backward sweep of segment %d (%d group%s)" % (j, gb, "s" * (gb > 1)), 0, npos_sweep * gb)
        if j < S - 1:
            add("sb", "P = silent_backward[%d] read by segment %d's groups" % (j + 1, j), 0, 4 * npos_sweep * gb)
        cs_pos = 1 if j == 0 else npos_sweep          # the first segment's row is only folded at position 1
        add("sb", "silent_backward[%d] folded by segment %d (write per group, read by all but the first)" % (j, j), 4 * cs_pos * gb, 4 * cs_pos * (gb - 1))
        add("bw", "spilled backward rows of segment %d: %d HMMs x %d half-columns x %d positions" % (j, nh[j], halves[j], npos_spill), 4 * halves[j] * nh[j] * npos_spill, 0)
        if lead and restart:
            add("codes+sb", "bridges of segment %d: %d HMMs x %d positions (base code + silent row; L2-resident)" % (j, nh[j], W), 0, 5 * W * nh[j])
            add("rs", "hand-over rows of segment %d's bridges" % j, 8 * nc[j] * nh[j], 8 * nc[j] * nh[j])
    for j in range(S):                  # forward sweeps
        lead, trail = j < ps, j >= sf
        npos = cut if lead else L
        npos_rows = (L - stop + 1) if trail else npos
        gf = groups_f[j]
        add("codes", "forward sweep of segment %d (%d group%s)" % (j, gf, "s" * (gf > 1)), 0, npos * gf)
        if j > 0:
            add("sf", "P = silent_forward[%d] read by segment %d's groups" % (j - 1, j), 0, 4 * npos * gf)
        cs_pos = 1 if j == S - 1 else npos          # the last segment's row: only where a read ends (TDS_LATE_CS)
        add("sf", "silent_forward[%d] folded by segment %d" % (j, j), 4 * cs_pos * gf, 4 * cs_pos * (gf - 1))
        if pure[j]:
            add("sb", "next segment's silent_backward for the recomputed pure column of segment %d" % j, 0, 4 * npos_rows * gf)
        add("bw", "spilled rows of segment %d read back: %d positions" % (j, npos_rows), 0, 4 * halves[j] * nh[j] * npos_rows)
        if nh[j] > 1:
            add("total", "total_prob of segment %d" % j, 4 * nh[j], 4 * nh[j])
    # posteriors: the rows that exist (a label's row is stored only where some read of the tile has a non-zero posterior)
    hl = H - first_n
    lead_labels = sum(nh[j] for j in range(ps)) - first_n if ps else 0
    add("dp", "posterior rows written (leading labels ~12 positions each, read-segment labels every position)", 4 * (max(lead_labels, 0) * 12 + (hl - max(lead_labels, 0)) * 40 + L), 0)
    add("dp", "posterior rows requested by the label DP (every label up to the cut, the others beyond: loads are unconditional)", 0, 4 * (hl * cut + (hl - max(lead_labels, 0)) * (L - cut)))
    pw = (hl + 3) // 4
    add("path", "label-DP path words (words of predecessor-free labels are not stored)", 4 * max(pw - max(lead_labels, 0) // 4, 1) * L, 4 * max(pw - max(lead_labels, 0) // 4, 1) * L)
    if first_n:
        add("bm/ba", "running maximum of the first segment's label sums", 5 * cut * groups_f[0], 5 * cut * (groups_f[0] - 1) + 5 * L)
    add("labels", "ri->labels written by the traceback, read back for the runs and by extract_reads", L + 1, 2 * (L + 1))
    add("out", "records, keep bits, label runs", 32 + 4 * nw1 + 4 * (S + 2), 4 * nw1)
    tot_w = sum(r[2] for r in rows)
    tot_r = sum(r[3] for r in rows)
    by = {}
    for a, _, w, r in rows:
        by.setdefault(a, [0.0, 0.0])
        by[a][0] += w
        by[a][1] += r
    print("%s: L = %d, cut %d, spill cut %d, stop %d / %d, bridge window %d, restart %d; bytes per read" % (wl, L, cut, cutS, stop, stopS, W, restart))
    for a, what, w, r in rows:
        print("  %-9s %8.0f w %8.0f r   %s" % (a, w, r, what))
    print("  by array: " + "  ".join("%s %.1f KB" % (a, (v[0] + v[1]) / 1e3) for a, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))))
    print("  model total: %.1f KB written + %.1f KB read = %.1f KB per read (without register-spill scratch)" % (tot_w / 1e3, tot_r / 1e3, (tot_w + tot_r) / 1e3))
    rec, note = bench.load_pmc_record(1 << 20 if wl != "c5" else 1 << 18, wl)
    if rec:
        n = rec["reads_per_launch"]
        print("  PMC (%s): %.1f KB written + %.1f KB fetched = %.1f KB per read; unexplained (scratch, re-fetches, unwritten posterior rows) %.1f KB" % (
            rec.get("round", "?"), rec["write_bytes"] / n / 1e3, rec["fetch_bytes_corrected_x2"] / n / 1e3, rec["hbm_bytes_per_launch"] / n / 1e3,
            (rec["hbm_bytes_per_launch"] / n - tot_w - tot_r) / 1e3))
    else:
        print("  PMC: " + note)


if __name__ == "__main__":
    main()
