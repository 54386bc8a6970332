#!/usr/bin/env python3
"""Per-loop summary of a specialised kernel's ISA (tools/spec_check.py <fixture> --keep leaves it in /tmp/td_spec): instructions,
VALU, global loads / stores, scratch (spill) traffic and full drains (s_waitcnt vmcnt(0)) of every innermost loop.  A spill reload
or a drain inside a position loop waits for every row requested ahead -- the loops to look at first.
usage: tools/isa_loops.py /tmp/td_spec/<fixture>-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
cur = None           # (header label, depth) of the block being read
pend_label = None
stats = {}
order = []
for n, l in enumerate(lines):
    m = re.match(r"^(\.LBB0_\d+):\s*(;.*)?$", l)
    if m:
        pend_label = m.group(1)[1:]
        cur = None
        c = m.group(2) or ""
        m2 = re.search(r"in Loop: Header=(BB0_\d+) Depth=(\d+)", c)
        if m2: cur = (m2.group(1), int(m2.group(2)))
        m2 = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", c)
        if m2: cur = (pend_label, int(m2.group(1)))
        continue
    t = l.strip()
    if t.startswith(";"):
        m2 = re.search(r"in Loop: Header=(BB0_\d+) Depth=(\d+)", t)
        if m2: cur = (m2.group(1), int(m2.group(2)))
        m2 = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", t)
        if m2 and pend_label: cur = (pend_label, int(m2.group(1)))
        continue
    if not t or t.startswith("."): continue
    if cur is None: continue
    st = stats.get(cur)
    if st is None:
        st = stats[cur] = dict(first=n + 1, instr=0, valu=0, gld=0, gst=0, sld=0, sst=0, vm0=0, ds=0, last=n + 1)
        order.append(cur)
    st["instr"] += 1; st["last"] = n + 1
    if t.startswith("v_"): st["valu"] += 1
    elif t.startswith("scratch_load"): st["sld"] += 1
    elif t.startswith("scratch_store"): st["sst"] += 1
    elif t.startswith("global_load"): st["gld"] += 1
    elif t.startswith("global_store"): st["gst"] += 1
    elif t.startswith("ds_"): st["ds"] += 1
    elif re.match(r"s_waitcnt.*vmcnt\(0\)", t): st["vm0"] += 1
print("%-10s %5s %13s %6s %6s %4s %4s %7s %7s %6s %5s" % ("header", "depth", "lines", "instr", "valu", "gld", "gst", "scr_ld", "scr_st", "drain", "lds"))
for k in order:
    st = stats[k]
    if st["instr"] >= 30:
        print("%-10s %5d %6d-%6d %6d %6d %4d %4d %7d %7d %6d %5d" % (k[0], k[1], st["first"], st["last"], st["instr"], st["valu"], st["gld"], st["gst"], st["sld"], st["sst"], st["vm0"], st["ds"]))
