"""pg_agg_linear_fwd (layer 0's aggregation + NodeUpdate in one kernel) against the pair pg_spmm_fwd_rows + pg_linear_fwd at the
in-loop shape (18.7 K source rows, 9.5 K destinations x 2 edges, 82 % hits, dropout on, K = 600, N = 32, skip-concat), over
ROTATING row sets (cold rows: what a minibatch is). Also checks that both routes produce the same bits."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pagraph_amd import _lib as L
lib = L.load(); dev = torch.device("cuda", 0); torch.cuda.set_device(0)
F, N, ncache = 600, 32, 2_559_329
n_src, n_dst = 18_700, 9_500
cap_dst = int(sys.argv[1]) if len(sys.argv) > 1 else n_dst
g = torch.Generator(device=dev).manual_seed(0)
fused = torch.rand((ncache, 608), device=dev); cache = fused[:, :F]
miss = torch.rand(n_src, device=dev, generator=g) < 0.18
m = int(miss.sum())
staged = torch.rand((max(m, 1), F), device=dev)
deg = torch.full((cap_dst,), 2, dtype=torch.int32, device=dev); deg[n_dst:] = 0
indptr = torch.zeros(cap_dst + 1, dtype=torch.int32, device=dev); indptr[1:] = torch.cumsum(deg, 0)
src = torch.randint(0, n_src, (2 * n_dst,), device=dev, dtype=torch.int32, generator=g)
sp = L.stream_ptr()
W = (torch.rand((N, F), device=dev) - 0.5) * 0.1
b = torch.rand(N, device=dev)
agg = torch.empty((cap_dst, 608), device=dev); y = torch.empty((cap_dst, 2 * N), device=dev)
agg2 = torch.empty((cap_dst, 608), device=dev); y2 = torch.empty((cap_dst, 2 * N), device=dev)
step = torch.tensor([5], dtype=torch.int64, device=dev)
drop = L.PgDropout(13107, 1, 1234, L.ptr(step))
NS = 8
sets = []
for i in range(NS):
    sl = torch.randint(0, ncache, (n_src,), device=dev, dtype=torch.int32, generator=g)
    sl[miss] = -(torch.arange(m, device=dev, dtype=torch.int32) + 3)
    sets.append((sl, L.PgRowSource(sl.data_ptr(), cache.data_ptr(), staged.data_ptr(), 608, F, 0)))

def pair(i, dp, a=agg, yy=y):
    L.check(lib.pg_spmm_fwd_rows(L.ptr(indptr), L.ptr(src), ctypes.byref(sets[i % NS][1]), cap_dst, F, 0, L.ptr(a), 608, dp, None, 0, sp))
    L.check(lib.pg_linear_fwd(L.ptr(a), 608, L.ptr(W), L.ptr(b), L.ptr(yy), 2 * N, cap_dst, F, N, 2, sp))
def fusedk(i, dp, a=agg2, yy=y2):
    L.check(lib.pg_agg_linear_fwd(L.ptr(indptr), L.ptr(src), ctypes.byref(sets[i % NS][1]), cap_dst, F, 0, dp, L.ptr(W), L.ptr(b), N, 2,
                                  L.ptr(a), 608, L.ptr(yy), 2 * N, None, 0, sp))
def rows_only(i, dp):
    L.check(lib.pg_spmm_fwd_rows(L.ptr(indptr), L.ptr(src), ctypes.byref(sets[i % NS][1]), cap_dst, F, 0, L.ptr(agg), 608, dp, None, 0, sp))
def lin_only(i, dp):
    L.check(lib.pg_linear_fwd(L.ptr(agg), 608, L.ptr(W), L.ptr(b), L.ptr(y), 2 * N, cap_dst, F, N, 2, sp))

def timeit(fn, dp, reps=200):
    for i in range(16):
        fn(i, dp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i, dp)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

nbytes = 2 * n_dst * (4 * F + 8) + n_dst * (4 * F + 4)
for wd in (True, False):
    dp = ctypes.byref(drop) if wd else None
    pair(3, dp); fusedk(3, dp); torch.cuda.synchronize()
    same = torch.equal(agg[:n_dst, :F], agg2[:n_dst, :F]) and torch.equal(y[:n_dst], y2[:n_dst])
    t_rows, t_lin, t_pair, t_fused = (timeit(f, dp) for f in (rows_only, lin_only, pair, fusedk))
    print(f"dropout={wd} cap_dst={cap_dst}: rows {t_rows:.1f} us, linear {t_lin:.1f} us, pair {t_pair:.1f} us, fused {t_fused:.1f} us "
          f"({(nbytes + n_dst * 8 * N) / t_fused / 1e3:.0f} GB/s = {(nbytes + n_dst * 8 * N) / t_fused / 8e6:.2f} of peak)  bit-identical: {same}")

# where a block of the fused kernel spends its time (debug stamps, 100 MHz clock)
import ctypes as C
if hasattr(lib, "pg_debug_agg_stamps") or True:
    try:
        f = lib.pg_debug_agg_stamps
        nb = max(512, (cap_dst + 31) // 32)
        st = torch.zeros(nb * 4, dtype=torch.int64, device=dev)
        f.argtypes = [C.c_void_p]; f(C.c_void_p(st.data_ptr()))
        dp = ctypes.byref(drop)
        for i in range(4):
            fusedk(i, dp)
        torch.cuda.synchronize()
        t = st.view(nb, 4).cpu().double()
        t0 = t[:, 0].min()
        t = t[t[:, 0] > 0]                      # the blocks that ran
        print("fused kernel, per block (us): start %.2f +- %.2f | phase 1 %.2f | phase 2 %.2f | phase 3 %.2f | last end %.2f" % (
            ((t[:, 0] - t0).mean() / 100), ((t[:, 0] - t0).std() / 100), ((t[:, 1] - t[:, 0]).mean() / 100),
            ((t[:, 2] - t[:, 1]).mean() / 100), ((t[:, 3] - t[:, 2]).mean() / 100), ((t[:, 3].max() - t0) / 100)))
        q = torch.tensor([0.1, 0.5, 0.9], dtype=torch.float64)
        print("  start quantiles", ((t[:, 0] - t0) / 100).quantile(q).tolist(), "end quantiles", ((t[:, 3] - t0) / 100).quantile(q).tolist())
        f(C.c_void_p(0))
    except Exception as e:
        print("no debug stamps:", e)
