#!/usr/bin/env python3
"""One run_pHMM-sized call (2^20 reads in, records + labels + rewritten sequences out, pageable buffers) through
td_multi_decode on one device: the device's range in one piece vs in pipelined pieces (TD_MULTI_PIECES), beside the
synchronous single-context calls.  Results compared byte for byte."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (TD_HOST_THREADS as the caller sets it; the library uses up to 16 host threads for its copies by default)
import bench
from tagdust_amd import TagdustHip
from tagdust_amd.lib import TagdustMulti
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
bench.select_workload(wl)
model = bench.load_model()
reads, offs = bench.synth_host_batch(n, 9)
c = TagdustHip(0)
c.upload_model(model); c.set_params(float(model["threshold"]), 16, 100)
from tagdust_amd import RESULT_DTYPE
# the caller's buffers exist once (a fresh np.zeros per call would time the kernel's page faults)
res = np.zeros(n, RESULT_DTYPE); lab = np.zeros(int(offs[-1]) + n, np.int8); sq = np.zeros(int(offs[-1]), np.uint8)
res[:] = res; lab[:] = 1; sq[:] = 1
def sync_call():
    c.upload_batch(reads, offs); c.run()
    c._chk(c.lib.td_batch_download(c.h, res.ctypes.data, lab.ctypes.data, sq.ctypes.data))
    return res.copy(), lab.copy(), sq.copy()
base = sync_call()
ts = []
for k in range(5):
    t = time.perf_counter()
    c.upload_batch(reads, offs); c.run(); c._chk(c.lib.td_batch_download(c.h, res.ctypes.data, lab.ctypes.data, sq.ctypes.data))
    ts.append(time.perf_counter() - t)
print("%s synchronous calls (upload, run, download):        %.1f ms per call (min of 5)" % (wl, min(ts) * 1e3), flush=True)
c.close()
for pieces in ("1", "2", "4", "8"):
    os.environ["TD_MULTI_PIECES"] = pieces
    m = TagdustMulti([0])
    m.upload_model(model); m.set_params(float(model["threshold"]), 16, 100)
    def call():
        m._chk(m.lib.td_multi_decode(m.h, reads.ctypes.data, 0, offs.ctypes.data, n, 1, res.ctypes.data, lab.ctypes.data, sq.ctypes.data))
    call()
    ts = []
    for k in range(5):
        lab[:] = 0
        t = time.perf_counter(); call(); ts.append(time.perf_counter() - t)
    out = (res, lab, sq)
    same = out[0].tobytes() == base[0].tobytes() and np.array_equal(out[1], base[1]) and np.array_equal(out[2], base[2])
    print("%s td_multi_decode, one device, %s piece(s):            %.1f ms per call (min of 5)  identical %s" % (wl, pieces, min(ts) * 1e3, same), flush=True)
    m.close()
