#!/usr/bin/env python3
"""Ahead-of-time check of the model-specialised kernel: generate its HIP source for a fixture model
(no GPU needed) and compile it for gfx950 with hipcc.  usage: tools/spec_check.py [fixture] [--keep]"""
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def check(name="c3_b6_s_r_p", keep=False, outdir="/tmp/td_spec"):
    from tagdust_amd import lib as tdlib
    z = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
    md = {k: z[k] for k in z.files}
    src = tdlib.spec_source(md)
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, name + ".hip")
    open(path, "w").write(src)
    t0 = time.time()
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
           "--cuda-device-only", "-c", path, "-o", os.path.join(outdir, name + ".o"), "-Rpass-analysis=kernel-resource-usage"]
    if os.environ.get("TD_SPEC_SLP", "0") == "0":
        cmd += ["-fno-slp-vectorize"]      # as td_jit.hip compiles it
    if os.environ.get("TD_SPEC_MLICM", "0") == "0":
        cmd += ["-mllvm", "-disable-machine-licm"]
    if os.environ.get("TD_SPEC_SCHED", "default") not in ("", "default"):
        cmd += ["-mllvm", "-amdgpu-sched-strategy=" + os.environ["TD_SPEC_SCHED"]]
    cmd += os.environ.get("TD_SPEC_EXTRA_OPTS", "").split()
    if keep:
        cmd += ["-save-temps=obj"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = p.stdout.decode()
    if p.returncode != 0:
        sys.stderr.write(out[-4000:])
        raise SystemExit("specialised kernel for %s did not compile" % name)
    info = [l.split("remark:")[1].strip() for l in out.splitlines() if "remark:" in l and any(
        k in l for k in ("VGPRs:", "AGPRs", "Scratch", "Occupancy", "Spill", "LDS Size"))]
    return time.time() - t0, info, len(src)


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c3_b6_s_r_p"]
    for n in names:
        dt, info, nsrc = check(n, keep="--keep" in sys.argv)
        print("%s: %.1f s, %d bytes of source; %s" % (n, dt, nsrc, "; ".join(info)))
