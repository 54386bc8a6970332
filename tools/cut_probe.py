#!/usr/bin/env python3
"""Position pruning A/B on one resident batch: kernel time, cut statistics (TD_SPEC_PRUNE_STATS) and a bit-for-bit
comparison of every output with the unpruned kernel.  usage: cut_probe.py [c3|c2|c5] [reads]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tagdust_amd import TagdustHip
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
bench.select_workload(wl)
model = bench.load_model()
reads, offs = bench.synth_host_batch(n, 5)
base = None
for prune, stats in ((0, 0), (1, 0), (1, 1), (0, 0), (1, 0)):
    os.environ["TD_SPEC_PRUNE"] = str(prune)
    os.environ["TD_SPEC_PRUNE_STATS"] = str(stats)
    c = TagdustHip(0)
    c.upload_model(model); c.set_params(float(model["threshold"]), 16, 100)
    c.upload_batch(reads, offs)
    ms = []
    for k in range(5):
        c.counts_reset()
        c.run(); c.sync(); ms.append(c.last_kernel_ms())
    cnt = c.diag()
    res, labels, seq = c.download()
    note = ""
    if stats:
        d = cnt[232 - 192:236 - 192]
        t = cnt[236 - 192:240 - 192]
        sx = cnt[224 - 192:228 - 192]
        note = "  trailing: decisions %d (mean required stop %.1f) fall-backs %d mean stop %.1f |" % (sx[0], sx[1] / max(sx[0], 1), sx[2], sx[3] / max(cnt[236 - 192], 1))
        note += "  lanes failing wa/wb/tot %s" % cnt[228 - 192:231 - 192].tolist()
        rsx = cnt[194 - 192:200 - 192]
        note += "  restarts: backward tiles %d (failed %d), bridges %d (mean interval steps %.1f) | forward tiles %d (failed %d), bridges %d (mean interval steps %.1f) |" % (
            cnt[206 - 192], cnt[207 - 192], rsx[3], rsx[2] / max(rsx[3], 1), rsx[4], rsx[5], rsx[1], rsx[0] / max(rsx[1], 1))
        note += "  decisions %d (mean required cut %.1f)  spill too short %d  failed checks %d | tiles %d  mean cut %.1f  dense %d  mean spill cut %.1f" % (
            d[0], d[1] / max(d[0], 1), d[2], d[3], t[0], t[1] / max(t[0], 1), t[2], t[3] / max(t[0], 1))
    if base is None:
        base = (res.copy(), labels.copy(), seq.copy())
    else:
        same = res.tobytes() == base[0].tobytes() and np.array_equal(labels, base[1]) and np.array_equal(seq, base[2])
        note += "  outputs identical: %s" % same
        if not same:
            for f in res.dtype.names:
                d = int((res[f].view(np.uint32) != base[0][f].view(np.uint32)).sum())
                if d: note += " %s:%d" % (f, d)
            note += " labels:%d seq:%d" % (int((labels != base[1]).sum()), int((seq != base[2]).sum()))
    print("%s prune %d: kernel %.2f ms (min of 5; %s)%s" % (wl, prune, min(ms), " ".join("%.1f" % m for m in ms), note), flush=True)
    c.close()
