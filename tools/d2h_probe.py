import sys, os, time
import numpy as np
sys.path.insert(0, "/root/repo")
os.environ.setdefault("TD_HOST_THREADS", "2")
import bench
from tagdust_amd import TagdustHip, RESULT_DTYPE
bench.select_workload("c3")
model = bench.load_model()
n = 1 << 20
hbs = [bench.synth_host_batch(n, 100 + b) for b in range(3)]
c = TagdustHip(0)
c.set_option("pipeline_depth", 3)
c.upload_model(model); c.set_params(float(model["threshold"]), 16, 100)
outs = [(np.zeros(n, RESULT_DTYPE), np.zeros(n * 150, np.uint8)) for _ in range(4)]
def run(steps, mode):
    t = []
    for k in range(steps):
        o = outs[k % 4]; hb = hbs[k % 3]
        kw = dict(res=o[0], seq_out=o[1]) if mode == "full" else (dict(res=o[0]) if mode == "res" else dict())
        t.append(c.submit(hb[0], hb[1], **kw))
        if len(t) >= 3: c.wait(t.pop(0))
    for x in t: c.wait(x)
for mode in (sys.argv[1:] or ["full", "res", "none", "full", "none"]):
    run(6, mode); c.sync()
    t0 = time.perf_counter(); run(24, mode); c.sync(); dt = time.perf_counter() - t0
    print("outputs %-5s: %.2f ms per step, %.2f M reads/s" % (mode, dt / 24 * 1e3, n * 24 / dt / 1e6), flush=True)
c.close()
