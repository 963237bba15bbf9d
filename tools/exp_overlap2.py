"""dev experiment: PCIe scatter (long, few waves) beside a chain of ~90 small dependent kernels"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pagraph_amd import _lib as L
lib = L.load(); dev = torch.device("cuda", 0); torch.cuda.set_device(0)
V, F, M = 2_000_000, 600, 8400
tab = torch.rand((V, F)).pin_memory()
pos = torch.arange(M, dtype=torch.int32, device=dev); full = torch.randint(0, V, (M,), device=dev)
cnt = torch.tensor([M], dtype=torch.int32, device=dev); out = torch.empty((M, F), device=dev)
import os as _o
pa, pb = int(_o.environ.get("PA", "0")), int(_o.environ.get("PB", "0"))
sA, sB = torch.cuda.Stream(priority=pa), torch.cuda.Stream(priority=pb)
print("prio scatter", pa, "chain", pb)
def scatter(): L.check(lib.pg_scatter_rows_from_host(L.ptr(tab), F, L.ptr(pos), L.ptr(full), M, L.ptr(cnt), F, L.ptr(out), F, L.stream_ptr(sA)))
small = torch.rand((6000, 64), device=dev); big = torch.rand((12000, 600), device=dev); w = torch.rand((600, 32), device=dev)
def chain(n_small, n_big):
    with torch.cuda.stream(sB):
        for _ in range(n_big): torch.mm(big, w)
        for _ in range(n_small): small.add_(1.0)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / reps * 1e6
with torch.cuda.stream(sB):
    torch.mm(big, w); small.add_(1.0)
torch.cuda.synchronize()
for ns, nb in ((90, 0), (0, 4), (60, 2)):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=sB):
        for _ in range(nb): torch.mm(big, w)
        for _ in range(ns): small.add_(1.0)
    def gr():
        with torch.cuda.stream(sB): g.replay()
    print(f"chain small={ns} big={nb}: scatter alone {timed(scatter):.0f} | eager chain alone {timed(lambda: chain(ns, nb)):.0f} | graph alone {timed(gr):.0f} | "
          f"scatter+eager {timed(lambda: (scatter(), chain(ns, nb))):.0f} | scatter+graph {timed(lambda: (scatter(), gr())):.0f} | graph first {timed(lambda: (gr(), scatter())):.0f}")
# ---- same low-occupancy long kernel but reading a DEVICE table (no PCIe): is the interference PCIe-specific?
tabd = torch.rand((400_000, F), device=dev)
fulld = torch.randint(0, 400_000, (M * 40,), device=dev); posd = (torch.arange(M * 40, device=dev) % M).to(torch.int32)
cntd = torch.tensor([M * 40], dtype=torch.int32, device=dev)
def scatter_dev(): L.check(lib.pg_scatter_rows_from_host(L.ptr(tabd), F, L.ptr(posd), L.ptr(fulld), M * 40, L.ptr(cntd), F, L.ptr(out), F, L.stream_ptr(sA)))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=sB):
    for _ in range(90): small.add_(1.0)
def gr():
    with torch.cuda.stream(sB): g.replay()
print(f"DEVICE-table long kernel alone {timed(scatter_dev):.0f} | graph(90 small) alone {timed(gr):.0f} | both {timed(lambda: (scatter_dev(), gr())):.0f}")
