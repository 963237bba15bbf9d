import ctypes
"""PMC traffic measurement of k_gather (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes).
Launch order: 3x calibration copy (known bytes), 3x gather uniform ids, 3x gather degree-skewed ids, at R rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pagraph_amd import _lib as L
lib = L.load(); dev = torch.device("cuda", 0); torch.cuda.set_device(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
V, F = 8_531_099, 600
ncache = int(V * 0.3)
fused = torch.rand((ncache, 608), device=dev); cache = fused[:, :F]; cnorm = fused[:, F:F + 1]   # product layout
slot = torch.full((V,), -1, dtype=torch.int32, device=dev)
cached = torch.randperm(V, device=dev)[:ncache].contiguous()
sp = L.stream_ptr()
L.check(lib.pg_slot_map_assign(L.ptr(slot), L.ptr(cached), ncache, sp))
nid_map = torch.arange(V, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
u = torch.rand(R, device=dev, generator=g)
ids_u = cached[(u * ncache).long().clamp(max=ncache - 1)].contiguous()
ids_s = cached[((u ** 4) * ncache).long().clamp(max=ncache - 1)].contiguous()
out = torch.empty((R, F), device=dev); onorm = torch.empty((R, 1), device=dev)
mpos = torch.empty(R, dtype=torch.int32, device=dev); mfull = torch.empty(R, dtype=torch.int64, device=dev)
mcnt = torch.zeros(1, dtype=torch.int32, device=dev)
slots = torch.empty(R, dtype=torch.int32, device=dev)
fields, nf = L.make_fields([(cache, out, F, 608, F), (cnorm, onorm, 1, 608, 1)])
src = torch.rand((R, F), device=dev)
torch.cuda.synchronize()
for _ in range(3):
    out.copy_(src)                       # calibration: reads R*2400 B, writes R*2400 B
torch.cuda.synchronize()
for ids in (ids_u, ids_s):
    for _ in range(3):
        L.check(lib.pg_gather_rows(L.ptr(ids), R, L.ptr(slot), L.ptr(nid_map), fields, nf, ctypes.byref(L.miss_list(mpos, mfull, mcnt)), L.ptr(slots), None, None, None, sp))
    torch.cuda.synchronize()
print("rows", R, "copy bytes each way", R * F * 4, "gather algorithmic bytes", R * (8 * 601 + 17))
