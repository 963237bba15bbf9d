"""dev experiment: do two streams overlap on this box? zero-copy scatter-like PCIe read kernel vs a GEMM loop"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pagraph_amd import _lib as L
lib = L.load(); dev = torch.device("cuda", 0); torch.cuda.set_device(0)
V, F, M = 2_000_000, 600, 8400
tab = torch.rand((V, F)).pin_memory()
pos = torch.arange(M, dtype=torch.int32, device=dev)
full = torch.randint(0, V, (M,), device=dev)
cnt = torch.tensor([M], dtype=torch.int32, device=dev)
out = torch.empty((M, F), device=dev)
a = torch.rand((12000, 600), device=dev); w = torch.rand((600, 32), device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def scatter(stream):
    L.check(lib.pg_scatter_rows_from_host(L.ptr(tab), F, L.ptr(pos), L.ptr(full), M, L.ptr(cnt), F, L.ptr(out), F, L.stream_ptr(stream)))
def gemms(stream, n=8):
    with torch.cuda.stream(stream):
        for _ in range(n): torch.mm(a, w)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e6
print("scatter alone  %.0f us" % timed(lambda: scatter(sA)))
print("gemms alone    %.0f us" % timed(lambda: gemms(sB)))
print("both, 2 streams %.0f us" % timed(lambda: (scatter(sA), gemms(sB))))
print("both, 1 stream  %.0f us" % timed(lambda: (scatter(sA), gemms(sA))))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=sB):
    for _ in range(8): torch.mm(a, w)
def graph_on_B():
    with torch.cuda.stream(sB): g.replay()
print("graph alone    %.0f us" % timed(graph_on_B))
print("scatter + graph, 2 streams %.0f us" % timed(lambda: (scatter(sA), graph_on_B())))
