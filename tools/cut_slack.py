#!/usr/bin/env python3
"""How much of the pruning cut is slack?  The oracle's matrices say where the leading segments' posterior terms really fall below
the reference's zero (forward + backward - b_score < -103.98); the device kernel's cut is the position its host bounds can prove
that for (tools/cut_probe.py prints it: config 3 31, config 5 37).  usage: tools/cut_slack.py  (CPU only)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import pyoracle  # noqa: E402
from tagdust_amd import lib as tdlib  # noqa: E402

for wl, n in (("c3", 4000), ("c5", 600), ("c2", 4000)):
    bench.select_workload(wl)
    model = bench.load_model()
    reads, offs = bench.synth_host_batch(n, 5)
    info = tdlib.spec_prune_info(model, int(np.diff(offs).max()) + 2)
    last, prof = pyoracle.lead_profile(pyoracle.OracleModel(model), reads, offs, info["n_seg"])
    tiles = [int(last[i:i + 64].max()) for i in range(0, n, 64)]
    print("%s: %d leading segments; last position with a live posterior term: mean %.1f, max %d; per tile of 64 reads: mean %.1f, max %d"
          % (wl, info["n_seg"], last.mean(), last.max(), np.mean(tiles), max(tiles)))
    L = int(np.diff(offs).max())
    print("   largest term by position 1..44:   ", " ".join("%5.0f" % v for v in prof[0][1:45]))
    print("   largest forward value:            ", " ".join("%5.0f" % v for v in prof[1][1:45]))
    print("   host bound fb[i]:                 ", " ".join("%5.0f" % v for v in info["fb"][1:45]))
    print("   largest backward value - b_score: ", " ".join("%5.0f" % v for v in prof[2][1:45]))
    print("   host bound bwb[len - i] (len %d): " % L, " ".join("%5.0f" % info["bwb"][L - i] for i in range(1, 45)))
    print("   zero bound z = %.2f" % info["z"])
    o = pyoracle.lead_class_profile(pyoracle.OracleModel(model), reads, offs, info["n_seg"])
    print("   what bounds per state class could give at best -- max over classes of (largest forward + largest backward - b) of the class:")
    print("                                     ", " ".join("%5.0f" % v for v in (o[:, 0, :] + o[:, 1, :]).max(axis=0)[1:45]))
    print("   against the two global maxima:    ", " ".join("%5.0f" % v for v in (o[:, 0, :].max(axis=0) + o[:, 1, :].max(axis=0))[1:45]))
