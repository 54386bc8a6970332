import os, sys, time, numpy as np
sys.path.insert(0, '.')
import bench
from tagdust_amd import TagdustHip, lib as tdlib
rng = np.random.default_rng(3)
bars = []
for line in open('tests/golden/EDITTAG_6nt_ed_4_first4.txt'):
    pass
# 96 random distinct 6-mers as barcodes
seen = set()
while len(seen) < 96:
    seen.add("".join("ACGT"[k] for k in rng.integers(0, 4, 6)))
bars = sorted(seen)
p5 = "GACCACCACATAACT"
n, L = 1 << 18, 150
code = {"A": 0, "C": 1, "G": 2, "T": 3}
reads = np.zeros((n, L), np.uint8)
b_idx = rng.integers(0, 96, n)
barr = np.array([[code[c] for c in b] for b in bars], np.uint8)
p5a = np.array([code[c] for c in p5], np.uint8)
reads[:] = rng.integers(0, 4, (n, L))
reads[:, :15] = p5a
reads[:, 15:21] = barr[b_idx]
ad = np.array([code[c] for c in bench.ADAPTER], np.uint8)
reads[:, L - 13:] = ad
offs = np.arange(n + 1, dtype=np.int64) * L
segs = ["P:" + p5, "B:" + ",".join(bars), "R:N", "P:" + bench.ADAPTER]
md, _ = tdlib.build_model(segs, reads[:20000].reshape(-1), offs[:20001], 0.05, 0.1)
md.update(threshold=2.0, minlen=16, dust=100)
c = TagdustHip(0); c.set_option("specialize", 1); c.upload_model(md); c.set_params(2.0, 16, 100)
c.upload_batch(reads.reshape(-1), offs)
c.run(); c.sync(); c.counts_reset(); c.run(); c.sync()
print("H", int(md["H"]), "kernel %.1f ms for 2^18 reads -> %.2f M reads/s" % (c.last_kernel_ms(), n / c.last_kernel_ms() / 1e3))
if os.environ.get("TD_SPEC_PROFILE"):
    t = c.diag()[240 - 192:252 - 192].astype(np.float64)
    names = ["unpack", "backward", "forward", "barprob", "labelDP", "traceback", "extraction", "artifacts", "DUST", "outputs", "-", "between"]
    print({nm: round(100 * x / t.sum(), 1) for nm, x in zip(names, t) if x})
cnt = c.counts(); print("outcomes", cnt[:8].tolist())
c.close()
