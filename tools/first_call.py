import sys, os, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
from tagdust_amd import TagdustHip, RESULT_DTYPE
bench.select_workload("c3")
model = bench.load_model()
n = 1 << 20
reads, offs = bench.synth_host_batch(n, 5)
for cand in ("3", "1"):
    os.environ["TD_WS_CANDIDATES"] = cand
    t = time.perf_counter(); c = TagdustHip(0); t_ctx = time.perf_counter() - t
    t = time.perf_counter(); c.upload_model(model); t_model = time.perf_counter() - t
    c.set_params(float(model["threshold"]), 16, 100)
    t = time.perf_counter(); c.upload_batch(reads, offs); t_up = time.perf_counter() - t
    t = time.perf_counter(); c.run(); c.sync(); t_run = time.perf_counter() - t
    res = np.zeros(n, RESULT_DTYPE); sq = np.zeros(n * 150, np.uint8)
    t = time.perf_counter(); tk = c.submit(reads, offs, res=res, seq_out=sq); tk2 = c.submit(reads, offs, res=res, seq_out=sq); c.wait(tk); c.wait(tk2); t_sub = time.perf_counter() - t
    t = time.perf_counter(); c.close(); t_close = time.perf_counter() - t
    print("candidates %s: ctx %.2f s, model upload %.2f s, first upload_batch %.2f s (workspace), first run %.3f s, first two submits (second workspace) %.2f s, close %.2f s" % (cand, t_ctx, t_model, t_up, t_run, t_sub, t_close), flush=True)
