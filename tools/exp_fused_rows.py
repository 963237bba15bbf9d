"""pg_spmm_fwd_rows alone at the in-loop shape (18.7 K source rows, 9.5 K destinations x 2 edges, 82 % hits, dropout on):
with / without pre-composed edge slots; and the unfused pair (pg_gather_rows + pg_spmm_fwd_drop) for reference"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pagraph_amd import _lib as L
lib = L.load(); dev = torch.device("cuda", 0); torch.cuda.set_device(0)
F, ncache, V = 600, 2_559_329, 8_531_099
n_src, n_dst = 18_700, 9_500
cap_dst = int(sys.argv[1]) if len(sys.argv) > 1 else n_dst          # 12000 = the padded launch of the hipGraph loop
g = torch.Generator(device=dev).manual_seed(0)
fused = torch.rand((ncache, 608), device=dev); cache = fused[:, :F]
slots = torch.randint(0, ncache, (n_src,), device=dev, dtype=torch.int32, generator=g)
miss = torch.rand(n_src, device=dev, generator=g) < 0.18
m = int(miss.sum()); slots[miss] = -(torch.arange(m, device=dev, dtype=torch.int32) + 3)
staged = torch.rand((max(m, 1), F), device=dev)
deg = torch.full((cap_dst,), 2, dtype=torch.int32, device=dev); deg[n_dst:] = 0
indptr = torch.zeros(cap_dst + 1, dtype=torch.int32, device=dev); indptr[1:] = torch.cumsum(deg, 0)
src = torch.randint(0, n_src, (2 * n_dst,), device=dev, dtype=torch.int32, generator=g)
es = torch.empty_like(src)
sp = L.stream_ptr()
L.check(lib.pg_compose_edge_slots(L.ptr(src), src.numel(), L.ptr(slots), n_src, L.ptr(es), sp))
out = torch.empty((cap_dst, F), device=dev)
step = torch.tensor([5], dtype=torch.int64, device=dev)
drop = L.PgDropout(13107, 1, 1234, L.ptr(step))
def run(use_es, reps=200, with_drop=True):
    dp = ctypes.byref(drop) if with_drop else None
    rs = L.PgRowSource(slots.data_ptr(), cache.data_ptr(), staged.data_ptr(), 608, F, es.data_ptr() if use_es else 0)
    for _ in range(10):
        L.check(lib.pg_spmm_fwd_rows(L.ptr(indptr), L.ptr(src), ctypes.byref(rs), cap_dst, F, 0, L.ptr(out), F, dp, None, 0, sp))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.pg_spmm_fwd_rows(L.ptr(indptr), L.ptr(src), ctypes.byref(rs), cap_dst, F, 0, L.ptr(out), F, dp, None, 0, sp))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
nbytes = 2 * n_dst * (4 * F + 8) + n_dst * (4 * F + 4)
print("kernel:", "generic (PG_FWD_ROWS_GENERIC)" if os.environ.get("PG_FWD_ROWS_GENERIC") else "wave-uniform")
us = run(False, with_drop=False)
print(f"pg_spmm_fwd_rows NO dropout cap_dst={cap_dst}: {us:.1f} us back-to-back  -> {nbytes / us / 1e3:.0f} GB/s ({nbytes / us / 1e3 / 8000:.2f} of peak)")
for use_es in (False, True):
    us = run(use_es)
    print(f"pg_spmm_fwd_rows edge_slots={use_es} cap_dst={cap_dst}: {us:.1f} us back-to-back  -> {nbytes / us / 1e3:.0f} GB/s ({nbytes / us / 1e3 / 8000:.2f} of peak)")
# the same launch over ROTATING row sets: 8 x 68 MB > the 256 MB Infinity Cache, so every repetition reads rows that are
# not resident (what the training loop does: each minibatch touches different rows); the loop above re-reads one set
NS = 8
sets = []
for i in range(NS):
    sl = torch.randint(0, ncache, (n_src,), device=dev, dtype=torch.int32, generator=g)
    sl[miss] = -(torch.arange(m, device=dev, dtype=torch.int32) + 3)
    sets.append((sl, L.PgRowSource(sl.data_ptr(), cache.data_ptr(), staged.data_ptr(), 608, F, 0)))
def run_rot(reps=200, with_drop=True):
    dp = ctypes.byref(drop) if with_drop else None
    for i in range(16):
        L.check(lib.pg_spmm_fwd_rows(L.ptr(indptr), L.ptr(src), ctypes.byref(sets[i % NS][1]), cap_dst, F, 0, L.ptr(out), F, dp, None, 0, sp))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        L.check(lib.pg_spmm_fwd_rows(L.ptr(indptr), L.ptr(src), ctypes.byref(sets[i % NS][1]), cap_dst, F, 0, L.ptr(out), F, dp, None, 0, sp))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for wd in (False, True):
    us = run_rot(with_drop=wd)
    print(f"pg_spmm_fwd_rows rotating row sets (cold rows) dropout={wd}: {us:.1f} us back-to-back -> {nbytes / us / 1e3:.0f} GB/s ({nbytes / us / 1e3 / 8000:.2f} of peak)")
# reference: materialised frame + pg_spmm_fwd_drop
h = torch.rand((n_src, F), device=dev)
for _ in range(10):
    L.check(lib.pg_spmm_fwd_drop(L.ptr(indptr), L.ptr(src), L.ptr(h), F, cap_dst, F, 0, L.ptr(out), F, ctypes.byref(drop), sp))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    L.check(lib.pg_spmm_fwd_drop(L.ptr(indptr), L.ptr(src), L.ptr(h), F, cap_dst, F, 0, L.ptr(out), F, ctypes.byref(drop), sp))
e1.record(); torch.cuda.synchronize()
print(f"pg_spmm_fwd_drop on a materialised frame: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us")
