"""dev experiment: in-loop-shaped gather (34K rows, ~25% misses): where does the time go?"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pagraph_amd import _lib as L
lib = L.load(); dev = torch.device("cuda", 0); torch.cuda.set_device(0)
V, F, R = 8_500_000, 600, 34205
ncache = int(V * 0.3)
cache = torch.rand((ncache, F), device=dev); cnorm = torch.rand((ncache, 1), device=dev)
slot = torch.full((V,), -1, dtype=torch.int32, device=dev)
cached = torch.randperm(V, device=dev)[:ncache]
slot[cached] = torch.arange(ncache, dtype=torch.int32, device=dev)
nid_map = torch.arange(V, device=dev)
def ids_with_miss(frac):
    nm = int(R * frac)
    unc = torch.nonzero(slot < 0).squeeze(1)
    a = cached[torch.randint(0, ncache, (R - nm,), device=dev)]
    b = unc[torch.randint(0, unc.numel(), (nm,), device=dev)]
    x = torch.cat([a, b]); return x[torch.randperm(R, device=dev)].contiguous()
out = torch.empty((R, F), device=dev); onorm = torch.empty((R, 1), device=dev)
mpos = torch.empty(R, dtype=torch.int32, device=dev)
mfull_d = torch.empty(R, dtype=torch.int64, device=dev)
mfull_p = torch.empty(R, dtype=torch.int64).pin_memory()
mcnt = torch.zeros(1, dtype=torch.int32, device=dev)
slots = torch.empty(R, dtype=torch.int32, device=dev)
sp = L.stream_ptr()
fields, nf = L.make_fields([(cache, out, F, F, F), (cnorm, onorm, 1, 1, 1)])
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for frac in (0.0, 0.25, 1.0):
    ids = ids_with_miss(frac)
    for name, mf in (("dev", mfull_d), ("pinned", mfull_p)):
        t = timeit(lambda: L.check(lib.pg_gather_rows(L.ptr(ids), R, L.ptr(slot), L.ptr(nid_map), fields, nf, ctypes.byref(L.miss_list(mpos, mf, mcnt)), L.ptr(slots), None, None, None, sp)))
        hits = R - int(mcnt.item())
        print(f"miss_frac={frac:.2f} missbuf={name:6s}: {t:6.1f} us/call (memset+kernel, back-to-back)  hits={hits}  alg GB/s={(hits*8*601+R*17)/t/1e3:.0f}")
