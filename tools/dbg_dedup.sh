#!/bin/bash
O=gpurun_out/lin; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "linear or model or head or graphed or virtual or reddit or deferred or golden" > $O/tests.log 2>&1; tail -3 $O/tests.log
for w in 4 8 16; do echo "PG_LINEAR_WAVES=$w"; PG_LINEAR_WAVES=$w python - <<'PY'
import torch, ctypes, os, sys
sys.path.insert(0, os.getcwd())
from pagraph_amd import _lib as L
lib = L.load(); dev = torch.device("cuda", 0)
def t(n, K, K2, N, act):
    pad = (K + 7) & ~7
    x = torch.rand((n, pad), device=dev)[:, :K]; w = torch.rand((N, K), device=dev); b = torch.rand(N, device=dev)
    x2 = torch.rand((n, K2), device=dev) if K2 else None; w2 = torch.rand((N, K2), device=dev) if K2 else None
    y = torch.empty((n, 2 * N), device=dev)
    def run():
        if K2: L.check(lib.pg_linear2_fwd(L.ptr(x), x.stride(0), L.ptr(w), L.ptr(b), K, L.ptr(x2), K2, L.ptr(w2), L.ptr(b), K2, L.ptr(y), 2 * N, n, N, act, L.stream_ptr()))
        else: L.check(lib.pg_linear_fwd(L.ptr(x), x.stride(0), L.ptr(w), L.ptr(b), L.ptr(y), 2 * N, n, K, N, act, L.stream_ptr()))
    for _ in range(10): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print(f"  n={n} K={K}+{K2} N={N}: {us:.1f} us  {n * (K + K2) * 4 / us / 1e3:.0f} GB/s")
for args in [(12000, 600, 0, 32, 2), (12000, 602, 0, 32, 2), (12000, 600, 600, 16, 2), (6000, 600, 600, 16, 2), (6000, 64, 0, 60, 0)]:
    t(*args)
PY
done
pick='import json,sys,statistics as st; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d["ms_per_step_windows"]; print(round(d["ms_per_step"],4), "median window", round(st.median(w),4))'
B="python bench.py --steps 1084 --warmup 20 --skip-cpu-baseline --skip-microbench --skip-opt-hit --skip-reference-equivalent"
echo "== gcn full cache"; timeout 300 $B --cache-ratio 1.0 2>/dev/null | python -c "$pick"
echo "== graphsage full cache"; timeout 300 $B --model graphsage --cache-ratio 1.0 2>/dev/null | python -c "$pick"
