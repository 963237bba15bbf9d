import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pagraph_amd import ops
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
n_dst, n_src, K, C = 6000, 9500, 64, 60
cnt = rng.integers(1, 3, n_dst)
indptr = np.zeros(n_dst + 1, np.int32); indptr[1:] = np.cumsum(cnt)
src = rng.integers(0, n_src, int(indptr[-1])).astype(np.int32)
tip, tsr = torch.from_numpy(indptr).to(dev), torch.from_numpy(src).to(dev)
h = torch.randn((n_src, K), device=dev)
lin = torch.nn.Linear(K, C).to(dev)
labels = torch.randint(0, C, (n_dst,), device=dev)
nv = torch.tensor([n_dst], dtype=torch.int32, device=dev)
seed = torch.tensor(1.0, device=dev)
step = torch.zeros(1, dtype=torch.int64, device=dev)
spec = ops.DropoutSpec(0.5, 1, 1, step)
with torch.no_grad():
    f = lambda: ops.gcn_head(tip, tsr, h, lin, labels, nv, seed, -100, "mean", spec, None)
    for _ in range(10): f()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): f()
    b.record(); torch.cuda.synchronize()
print("PG_HEAD_DBG=%s: %.1f us per call (head + sum_partials + allocs)" % (os.environ.get("PG_HEAD_DBG", "0"), a.elapsed_time(b) / 200 * 1e3))
