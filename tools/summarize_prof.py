#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by tools/profile_gpu.sh into one small JSON/markdown
(kernel average duration, PMC sums per dispatch, HBM traffic with the gfx950 FETCH_SIZE x2 correction
of MI355X_MICROARCH.md 'HBM')."""
import csv
import glob
import json
import os
import sys


def rows(pattern):
    out = []
    for p in glob.glob(pattern, recursive=True):
        with open(p) as fh:
            out += list(csv.DictReader(fh))
    return out


def main():
    d = sys.argv[1]
    kname = sys.argv[2] if len(sys.argv) > 2 else "td_spec_kernel"
    res = {}
    kt = [r for r in rows(os.path.join(d, "trace", "**", "*kernel_trace.csv")) if kname in r.get("Kernel_Name", "")]
    if kt:
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in kt]
        res["kernel_trace"] = {"kernel": kname, "dispatches": len(durs), "avg_ms": sum(durs) / len(durs),
                               "min_ms": min(durs), "max_ms": max(durs),
                               "vgpr": kt[0].get("VGPR_Count"), "accum_vgpr": kt[0].get("Accum_VGPR_Count"),
                               "sgpr": kt[0].get("SGPR_Count"), "lds": kt[0].get("LDS_Block_Size"),
                               "scratch": kt[0].get("Scratch_Size"), "grid": kt[0].get("Grid_Size"),
                               "workgroup": kt[0].get("Workgroup_Size")}
    st = rows(os.path.join(d, "trace", "**", "*kernel_stats.csv"))
    res["kernel_stats"] = [r for r in st][:8]
    pmc = {}
    for name in ("pmc1", "pmc2", "fetch", "write"):
        for r in rows(os.path.join(d, name, "**", "*counter_collection.csv")):
            if kname not in r.get("Kernel_Name", ""):
                continue
            pmc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    res["pmc_per_dispatch_avg"] = {k: sum(v) / len(v) for k, v in pmc.items()}
    p = res["pmc_per_dispatch_avg"]
    if "FETCH_SIZE" in p or "WRITE_SIZE" in p:
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
        f = p.get("FETCH_SIZE", 0.0) * 1024 * 2
        w = p.get("WRITE_SIZE", 0.0) * 1024
        res["hbm_bytes_per_launch"] = {"fetch_corrected_x2": f, "write": w, "total": f + w}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
