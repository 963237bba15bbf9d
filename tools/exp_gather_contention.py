"""dev experiment: event-bracketed k_gather duration alone vs with a busy compute stream"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pagraph_amd import _lib as L
lib = L.load(); dev = torch.device("cuda", 0); torch.cuda.set_device(0)
V, F, R = 8_500_000, 600, 18700
ncache = int(V * 0.3)
fused = torch.rand((ncache, 608), device=dev)
slot = torch.full((V,), -1, dtype=torch.int32, device=dev)
cached = torch.randperm(V, device=dev)[:ncache].contiguous()
slot[cached] = torch.arange(ncache, dtype=torch.int32, device=dev)
nid_map = torch.arange(V, device=dev)
ids = cached[torch.randint(0, ncache, (R,), device=dev)].contiguous()
out = torch.empty((R, F), device=dev)
mpos = torch.empty(R, dtype=torch.int32, device=dev); mfull = torch.empty(R, dtype=torch.int64, device=dev)
mcnt = torch.zeros(1, dtype=torch.int32, device=dev); slots = torch.empty(R, dtype=torch.int32, device=dev)
fields, nf = L.make_fields([(fused[:, :F], out, F, 608, F)])
sL = torch.cuda.Stream(priority=-1); sC = torch.cuda.Stream()
a = torch.rand((12000, 600), device=dev); w = torch.rand((600, 32), device=dev); b = torch.rand((12000, 64), device=dev)
def gather_timed(n=30):
    ts = []
    for _ in range(n):
        t = L.vp(); L.check(lib.pg_timer_create(ctypes.byref(t)))
        L.check(lib.pg_gather_rows(L.ptr(ids), R, L.ptr(slot), L.ptr(nid_map), fields, nf, ctypes.byref(L.miss_list(mpos, mfull, mcnt)), L.ptr(slots), None, t, None, L.stream_ptr(sL)))
        ts.append(t); time.sleep(0.0005)
    torch.cuda.synchronize()
    out_ms = []
    for t in ts:
        v = ctypes.c_float(); L.check(lib.pg_timer_elapsed_ms(t, ctypes.byref(v))); out_ms.append(v.value * 1e3); lib.pg_timer_destroy(t)
    return np.mean(out_ms), np.min(out_ms), np.max(out_ms)
print("alone: mean/min/max us", gather_timed())
def busy(kind, iters):
    with torch.cuda.stream(sC):
        for _ in range(iters):
            if kind == "gemm": torch.mm(a, w)
            elif kind == "small":
                for _ in range(10): b.add_(1.0)
            else: torch.mm(a, w); b.add_(1.0); torch.relu(b)
for kind in ("gemm", "small", "mix"):
    busy(kind, 3000)
    print(f"with busy {kind} stream:", gather_timed())
    torch.cuda.synchronize()
# ---- closer to the bench: gather followed by the zero-copy miss scatter on the same stream, back to back
Vh, M = 2_000_000, 3300
tab = torch.rand((Vh, F)).pin_memory()
pos = torch.arange(M, dtype=torch.int32, device=dev); fullh = torch.randint(0, Vh, (M,), device=dev)
cnt = torch.tensor([M], dtype=torch.int32, device=dev)
def loop(n, with_scatter, with_busy, gap):
    ts = []
    if with_busy: busy("mix", 400)
    for _ in range(n):
        t = L.vp(); L.check(lib.pg_timer_create(ctypes.byref(t)))
        L.check(lib.pg_gather_rows(L.ptr(ids), R, L.ptr(slot), L.ptr(nid_map), fields, nf, ctypes.byref(L.miss_list(mpos, mfull, mcnt)), L.ptr(slots), None, t, None, L.stream_ptr(sL)))
        if with_scatter:
            L.check(lib.pg_scatter_rows_from_host(L.ptr(tab), F, L.ptr(pos), L.ptr(fullh), M, L.ptr(cnt), F, L.ptr(out), F, L.stream_ptr(sL)))
        ts.append(t)
        if gap: time.sleep(gap)
    torch.cuda.synchronize()
    ms = []
    for t in ts:
        v = ctypes.c_float(); L.check(lib.pg_timer_elapsed_ms(t, ctypes.byref(v))); ms.append(v.value * 1e3); lib.pg_timer_destroy(t)
    return "mean %.1f min %.1f max %.1f us" % (np.mean(ms), np.min(ms), np.max(ms))
for ws in (False, True):
    for wb in (False, True):
        for gap in (0, 0.0005):
            print(f"scatter={ws} busy={wb} gap={gap}: {loop(40, ws, wb, gap)}")
# ---- HBM-heavy neighbour: dropout-sized elementwise traffic on the compute stream
big = torch.rand((24000, 600), device=dev); big2 = torch.empty_like(big)
def busy_hbm(iters):
    with torch.cuda.stream(sC):
        for _ in range(iters):
            torch.mul(big, 1.0001, out=big2)      # 115 MB of HBM traffic, ~25 us
busy_hbm(4000)
print("with HBM-heavy elementwise stream:", loop(40, False, False, 0.0002))
torch.cuda.synchronize()
