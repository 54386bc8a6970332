#!/usr/bin/env python3
"""Where the specialised kernel's scratch (register-spill) accesses sit: compile the kernel for a fixture's model ahead of
time (hipcc, no GPU), read the gfx950 assembly and list, per basic block, the instruction mix and the scratch loads /
stores, with the loop nesting LLVM prints.  The sweep loops are the depth-2 inner loops.
usage: tools/isa_report.py [fixture] > profiles/<tag>_isa_loops_<fixture>.txt"""
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))


# the two VALU issue classes tools/ubench/valu_rate.hip measures on gfx950 (ns per wave64 instruction per SIMD: ~1.0 / ~1.75)
SLOW_CLASS = ("v_max_f32", "v_min_f32", "v_cvt_i32_f32", "v_lshlrev_b32", "v_cndmask_b32", "v_cmp", "v_fma_f32", "v_med3_f32",
              "v_pk_add_f32", "v_pk_mul_f32", "v_max3_f32", "v_min3_f32", "v_cvt_")


def valu_slow_class_share(name="c3_b6_s_r_p"):
    """Static instruction mix of the position sweeps (basic blocks of loop depth >= 2) of the kernel compiled for a fixture's
    model: share of the VALU instructions that belong to the slow issue class.  Instructions the micro-benchmark did not
    time (v_mov, v_or, v_sub_u32, ...) count as fast, so the class-weighted roofline fraction built on this is a lower bound."""
    import spec_check
    out = "/tmp/td_isa_" + name
    spec_check.check(name, keep=True, outdir=out)
    asm = [f for f in os.listdir(out) if f.endswith(".s") and "gfx950" in f and f.startswith(name)]
    depth, in_kernel, n_valu, n_slow = 0, False, 0, 0
    for l in open(os.path.join(out, asm[0])).read().splitlines():
        if l.startswith("td_spec_kernel:"):
            in_kernel = True
        if l.startswith("td_spec_selfcheck:"):
            in_kernel = False
        if not in_kernel:
            continue
        if re.match(r"^\.LBB\d+_\d+:", l):
            depth = 0
        t = l.strip()
        if t.startswith(";") or ";" in l and re.match(r"^\.LBB", l):
            d = None if "Child Loop" in t else re.search(r"Depth[= ](\d+)", l)
            if d:
                depth = max(depth, int(d.group(1)))
            if t.startswith(";"):
                continue
        if not t or t.startswith(".") or depth < 2:
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            n_valu += 1
            if op.startswith(SLOW_CLASS):
                n_slow += 1
    return (n_slow / n_valu) if n_valu else None, n_valu


def main():
    import spec_check
    name = sys.argv[1] if len(sys.argv) > 1 else "c3_b6_s_r_p"
    out = "/tmp/td_isa_" + name
    dt, info, nsrc = spec_check.check(name, keep=True, outdir=out)
    asm = [f for f in os.listdir(out) if f.endswith(".s") and "gfx950" in f and f.startswith(name)]
    lines = open(os.path.join(out, asm[0])).read().splitlines()
    print("# %s: %s" % (name, "; ".join(i.split(":", 3)[-1].strip().split(" [")[0] for i in info[:7])))
    blocks, cur, in_kernel = [], None, False
    for i, l in enumerate(lines):
        if l.startswith("td_spec_kernel:"):
            in_kernel = True
        if l.startswith("td_spec_selfcheck:"):
            in_kernel = False
        if not in_kernel:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            cur = dict(name=m.group(1), cmt=m.group(2).strip(), n=0, sl=0, ss=0, ds=0, gl=0, gs=0, valu=0, salu=0, depth=0, header=False)
            blocks.append(cur)
            continue
        if cur is None:
            cur = dict(name="entry", cmt="", n=0, sl=0, ss=0, ds=0, gl=0, gs=0, valu=0, salu=0, depth=0, header=False)
            blocks.append(cur)
        t = l.strip()
        if t.startswith(";"):
            d = None if "Child Loop" in t else re.search(r"Depth[= ](\d+)", t)
            if d:
                cur["depth"] = max(cur["depth"], int(d.group(1)))
            if "Inner Loop Header" in t:
                cur["header"] = True
            continue
        if not t or t.startswith("."):
            continue
        op = t.split()[0]
        cur["n"] += 1
        if op.startswith("scratch_load"):
            cur["sl"] += 1
        elif op.startswith("scratch_store"):
            cur["ss"] += 1
        elif op.startswith("ds_"):
            cur["ds"] += 1
        elif op.startswith("global_load"):
            cur["gl"] += 1
        elif op.startswith("global_store"):
            cur["gs"] += 1
        elif op.startswith("v_"):
            cur["valu"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
    tot = dict(sl=0, ss=0)
    inner = dict(sl=0, ss=0, n=0)
    print("%-12s %5s %6s %6s %5s %6s %6s %8s %8s  %s" % ("block", "depth", "insts", "valu", "lds", "gload", "gstore", "scr_load", "scr_store", "what"))
    for b in blocks:
        tot["sl"] += b["sl"]; tot["ss"] += b["ss"]
        if b["depth"] >= 2:
            inner["sl"] += b["sl"]; inner["ss"] += b["ss"]; inner["n"] += b["n"]
        if b["n"] >= 120 or b["sl"] + b["ss"] > 0:
            what = ("inner loop " if b["depth"] >= 2 else "") + b["cmt"].lstrip("; ")[:70]
            print("%-12s %5d %6d %6d %5d %6d %6d %8d %8d  %s" % (b["name"], b["depth"], b["n"], b["valu"], b["ds"], b["gl"], b["gs"], b["sl"], b["ss"], what))
    print("# scratch accesses in the whole kernel: %d loads, %d stores; in blocks of loop depth >= 2 (the position sweeps, %d instructions): %d loads, %d stores"
          % (tot["sl"], tot["ss"], inner["n"], inner["sl"], inner["ss"]))


if __name__ == "__main__":
    main()
