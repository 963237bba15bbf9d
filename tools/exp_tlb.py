"""Why does the fused gather+aggregate kernel take 25 us per launch when the cache is 59.6 GB (config 5's graph on one GPU)
and 14 us when it is 6.2 GB?  The kernel alone, the in-loop shape (18.7 K source rows -> 9.5 K destinations
x 2 edges, all hits, dropout on), over caches of growing size and over slot distributions that touch fewer pages:

  uniform        slots uniform over the whole cache (worst case: every row on a page of its own)
  hot10          90 % of the slots in the first 10 % of the cache (the cache is filled in descending-degree order, so the
                 training loop's accesses look more like this than like `uniform`)
  first6g        slots uniform over the first 6.2 GB only, cache still allocated at full size (same pages as the small
                 cache: if this matches the small cache's time, reach of the address translation is the cause, not the
                 allocation's size as such)
  sorted         uniform, but the destinations' rows are visited in ascending slot order (neighbouring waves walk
                 neighbouring pages)

usage: python tools/exp_tlb.py [cache GB ...]     (default 6.2 24 59.6; rows are 608 floats = 2432 B)
Under `rocprofv3 --pmc <translation counters> --kernel-trace` the same launches give the miss counts (tools/exp_r04_tlb.sh)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pagraph_amd import _lib as L

lib = L.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
F, STRIDE = 600, 608
n_src, n_dst = 18_700, 9_500
sizes = [float(x) for x in sys.argv[1:]] or [6.2, 24.0, 59.6]
reps = int(os.environ.get("REPS", 200))
NS = 8                                     # rotating row sets: 8 x 68 MB > the 256 MB Infinity Cache
g = torch.Generator(device=dev).manual_seed(0)
sp = L.stream_ptr()
indptr = torch.arange(0, 2 * n_dst + 1, 2, dtype=torch.int32, device=dev)
out = torch.empty((n_dst, F), device=dev)
step = torch.tensor([5], dtype=torch.int64, device=dev)
drop = L.PgDropout(13107, 1, 1234, L.ptr(step))
nbytes = 2 * n_dst * (4 * F + 8) + n_dst * (4 * F + 4)


def time_sets(cache, make_slots, label, sort_edges=False):
    sets = []
    for i in range(NS):
        sl = make_slots().to(torch.int32)
        src = torch.randint(0, n_src, (2 * n_dst,), device=dev, dtype=torch.int32, generator=g)
        if sort_edges:               # edge e reads row src[e]: make the visit order ascending in slot
            order = torch.argsort(sl[src.long()].to(torch.int64))
            src = src[order].contiguous()
        sets.append((sl, src, L.PgRowSource(sl.data_ptr(), cache.data_ptr(), 0, STRIDE, F, 0)))
    dp = ctypes.byref(drop)

    def launch(i):
        sl, src, rs = sets[i % NS]
        L.check(lib.pg_spmm_fwd_rows(L.ptr(indptr), L.ptr(src), ctypes.byref(rs), n_dst, F, 0, L.ptr(out), F, dp, None, 0, sp))
    for i in range(16):
        launch(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        launch(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"  {label:10s} {us:6.1f} us per launch back to back -> {nbytes / us / 1e3:5.0f} GB/s ({nbytes / us / 1e3 / 8000:.2f} of peak)",
          flush=True)
    return us


for gb in sizes:
    rows = int(gb * 1e9 / (STRIDE * 4))
    try:
        cache = torch.empty((rows, STRIDE), device=dev)
    except RuntimeError as e:
        print(f"cache {gb} GB: allocation failed ({e})")
        continue
    # touch every page once (a fresh allocation may not be mapped until written) without a 60 GB temporary
    for lo in range(0, rows, 1 << 20):
        cache[lo:lo + (1 << 20)].fill_(0.5)
    torch.cuda.synchronize()
    print(f"cache {gb:.1f} GB = {rows} rows, base address {cache.data_ptr():#x} (mod 2 MiB = {cache.data_ptr() % (1 << 21)}, "
          f"mod 1 GiB = {cache.data_ptr() % (1 << 30)})", flush=True)
    small = int(6.2e9 / (STRIDE * 4))
    time_sets(cache, lambda: torch.randint(0, rows, (n_src,), device=dev, generator=g), "uniform")
    hot = max(1, rows // 10)
    time_sets(cache, lambda: torch.where(torch.rand(n_src, device=dev, generator=g) < 0.9,
                                         torch.randint(0, hot, (n_src,), device=dev, generator=g),
                                         torch.randint(0, rows, (n_src,), device=dev, generator=g)), "hot10")
    if rows > small:
        time_sets(cache, lambda: torch.randint(0, small, (n_src,), device=dev, generator=g), "first6g")
    time_sets(cache, lambda: torch.randint(0, rows, (n_src,), device=dev, generator=g), "sorted", sort_edges=True)
    del cache
    torch.cuda.empty_cache()
