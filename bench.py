#!/usr/bin/env python3
"""bench.py — the reference's headline measurement on MI355X: epoch time of the
PaGraph minibatch training loop (examples/profile/pa_gcn.py:82-106), cache-hit % and
feature-gather GB/s, on the synthetic RMAT 10M-vertex / 100M-edge graph, feat = 600,
30 % hot-degree cache (BASELINE.json configs[2]; model selectable, default GCN as in
BASELINE.json's `metric`).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  N = 1: the whole train set on one GPU ("1naive": closure of
every train vertex, hash.py with --partition 1).  N > 1: rank 0 runs dg (C++), every
rank builds the closure of its own partition, gradients all-reduced by DDP/RCCL; the
graph is fixed, so scaling is strong.  A "step" = one minibatch: sample -> fetch_data
(gather + miss path) -> forward/backward/Adam.  Inputs (graph, cache, host table) are
resident before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# (before the first HIP call: see pagraph_amd/__init__.py — more than ~6 hardware queues per process cost every side-stream
# kernel ~45 us; the N > 1 step's extra streams push the pipeline over that)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic(kernel="k_gather"):
    """HBM bytes per in-loop launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/rNN/pmc_<kernel>_inloop.json: FETCH_SIZE x2 + WRITE_SIZE, KiB, separate passes —
    MI355X_MICROARCH.md §HBM). bench.py cannot collect PMC counters itself; None when absent. The figure belongs
    to the default workload (10M/100M GCN, 30 % cache): other workloads get None."""
    import glob
    name = {"k_gather": "pmc_gather_inloop.json", "k_spmm_fwd_rows": "pmc_spmm_fwd_rows_inloop.json"}[kernel]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    return d.get("hbm_bytes_per_launch"), os.path.relpath(files[-1], ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cgroup_cpu_stat():
    """(nr_throttled, throttled_usec, usage_usec) of this process's cgroup, or None"""
    try:
        d = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)), int(d.get("usage_usec", 0))
    except (OSError, ValueError):
        return None


def host_info(cacher=None):
    """what the host side of the step had to work with: CPU model, cores visible, cgroup CPU quota,
    threads of the miss path's row gather"""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else int(q) / int(per)
    except (OSError, ValueError):
        pass
    return {"cpu_model": model, "cpus_online": os.cpu_count(), "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
            "cpus_affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cgroup_cpu_quota": quota, "miss_gather_threads": getattr(cacher, "host_threads", None)}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None, help="timed steps (default: one whole epoch of the rank's seeds)")
    p.add_argument("--warmup", type=int, default=10, help="untimed steady-state steps right before the timed region "
                   "(honoured as given; the one-off set-up steps — eager warm-up, hipGraph capture per ring slot, cache "
                   "fill after the first step — come before them and are reported as config.setup_steps)")
    p.add_argument("--window", type=int, default=20, help="steps per entry of ms_per_step_windows")
    p.add_argument("--cold-start", action="store_true", help="drain the pipeline before the timed region (every batch of "
                   "the timed region pays the sample -> gather -> miss-path latency from an empty pipeline; diagnosis only)")
    p.add_argument("--model", default="gcn", choices=["gcn", "graphsage"])
    p.add_argument("--vertices", type=int, default=10_000_000)
    p.add_argument("--edges", type=int, default=100_000_000)
    p.add_argument("--feat-size", type=int, default=600)
    p.add_argument("--n-classes", type=int, default=60)
    p.add_argument("--batch-size", type=int, default=6000)
    p.add_argument("--num-neighbors", type=int, default=2)
    p.add_argument("--cache-ratio", type=float, default=0.30)
    p.add_argument("--presample-epochs", type=int, default=8, help="epochs the presample policy counts accesses over")
    p.add_argument("--cache-policy", choices=("degree", "presample"), default="degree",
                   help="degree = the reference's rule (storage.py:97-104, the headline); presample = opt-in: the vertices "
                        "a presampled epoch (another sampler seed) looked up most often")
    p.add_argument("--miss-mode", default=None, choices=["staged", "zerocopy", "async"],
                   help="default: async (worker-thread queue) on one GPU. With --gpus > 1 both zerocopy and async are "
                        "timed for 40 steps after the warm-up and every rank keeps the faster one: on a single GPU one "
                        "more busy stream beside the async path (RCCL adds some) was measured to make the step 1.5-3x "
                        "slower, and the multi-GPU runs cannot be tried from the build container")
    p.add_argument("--host-threads", type=int, default=None, help="threads of the miss path's CPU row gather "
                   "(default: from the process's CPU quota, see storage.default_host_threads)")
    p.add_argument("--no-overlap", action="store_true")
    p.add_argument("--skip-cpu-baseline", action="store_true")
    p.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    p.add_argument("--cpu-baseline-min-steps", type=int, default=0, help="run at least this many CPU minibatches whatever the time")
    p.add_argument("--cpu-baseline-one-leg", action="store_true", help="the 16-thread leg only (no second leg on every core)")
    p.add_argument("--skip-microbench", action="store_true")
    p.add_argument("--dg-hops", type=int, default=None,
                   help="hops used by dg's affinity score (dg.py --num-hops). Default: 2 (README.md:117, the value for a "
                        "2-layer model without preprocessing = BASELINE configs 4 and 5)")
    p.add_argument("--fetch-all", action="store_true",
                   help="fetch every layer and field like the reference (default: only what the model reads, SURVEY 8f-2)")
    p.add_argument("--skip-reference-equivalent", action="store_true", help="skip the short run that fetches every layer "
                   "and field like the reference and counts the cache-hit rate its way")
    p.add_argument("--no-fuse-gather", action="store_true", help="materialise layer 0 (pg_gather_rows + pg_spmm_fwd_drop) "
                   "instead of aggregating it straight from the cache (pg_spmm_fwd_rows)")
    p.add_argument("--skip-opt-hit", action="store_true", help="skip the oracle cache-hit upper bound (opt_cache_hit.py)")
    p.add_argument("--ring", type=int, default=None, help="sampler ring slots (in-flight minibatches)")
    p.add_argument("--no-graph", action="store_true", help="eager reference-style loop instead of hipGraph replay")
    p.add_argument("--timeline", action="store_true", help="print a per-stream event timeline of a few steps (stderr)")
    p.add_argument("--no-transpose", action="store_true", help="sampler does not emit source-major blocks "
                   "(backward aggregation falls back to the atomic scatter form)")
    p.add_argument("--cpu-share", type=float, default=None, help="async miss path: share of every miss list that goes "
                   "through the worker thread; the rest is read by the device over PCIe (1.0 = all). Given: fixed. "
                   "Default: start at 1.0 and adapt after the set-up steps")
    p.add_argument("--inline-transpose", action="store_true", help="build the source-major blocks inside the sampler's "
                   "chain instead of on the trainer's load stream")
    p.add_argument("--probe-miss-mode", action="store_true", help="time a few steps with the zero-copy and with the async "
                   "miss path after the warm-up and keep the faster one (default when --gpus > 1 and no --miss-mode)")
    p.add_argument("--lookahead", type=int, default=None, help="batches prepared ahead of the one being computed "
                   "(default 2 with the async miss queue, else 1); the sampler ring needs lookahead + 2 slots")
    p.add_argument("--profile-host", action="store_true", help="cProfile the timed region (stderr)")
    p.add_argument("--dist-backend", default="nccl", help="gloo lets two ranks share one GPU (testing only)")
    p.add_argument("--no-epoch-leg", action="store_true", help="with --steps < one epoch: do not time a whole epoch for "
                   "`value`, scale the --steps window instead (the old behaviour)")
    p.add_argument("--skip-preflight", action="store_true", help="--gpus > 1: skip the ~10 s small-graph run through the whole "
                   "N-rank path (collectives, dg, closures, miss queues, captures, replica check) before the big set-up")
    p.add_argument("--no-configs", action="store_true", help="do not append the `configs` block (BASELINE configs 2 and 3, each "
                   "timed by a child run of this script) to the headline line")
    p.add_argument("--as-rank-of", type=int, default=0, metavar="P",
                   help="ONE GPU only (round 6): run ONE rank's share of a P-GPU job — dg over P partitions on the host, the rank "
                        "with the largest closure (or --which-rank), its own closure / local-out-degree cache / seeds, the N > 1 "
                        "step shape over a one-rank RCCL group (flat gradient buffer, all-reduce in the step), one epoch of the "
                        "step count all P ranks would run after MAX-equalisation. Reports a PROJECTED epoch time (equalised "
                        "steps x this rank's ms/step), labelled as such: RCCL across ranks is the one thing it cannot show")
    p.add_argument("--which-rank", type=int, default=None, help="--as-rank-of: the partition to run (default: largest closure)")
    p.add_argument("--dist-step", action="store_true", help="one GPU: the N > 1 step shape (one-rank RCCL group, trainer "
                   "world_size 2) over the 1naive partition — the A/B partner of the default one-GPU step")
    p.add_argument("--tape-collectives", action="store_true", help="A/B (N > 1 step): replay the captured step that holds the "
                   "all-reduce as plain launches too (GraphedTrainer.tape_collectives; default: that graph keeps hipGraphLaunch)")
    p.add_argument("--extra-streams", type=int, default=0, help="diagnosis: create and use N more streams before the trainer")
    p.add_argument("--no-fuse-partials", action="store_true", help="A/B: the weight gradients' ordered partial sums as launches "
                   "of their own (+ AccumulateGrad's adds when N > 1) instead of inside the optimiser's launch")
    p.add_argument("--no-adapt-cpu-share", action="store_true", help="keep --cpu-share as given; default: after the set-up "
                   "steps every rank sets it from its own CPU-gather rate vs PCIe (GraphCacheServer.adapt_cpu_share)")
    return p.parse_args()


# ----------------------------------------------------------------------------- host feature table
def make_host_table(V, Fdim, rank, local_rank, world, dev, tag):
    """the 'graph store': ONE host copy of the [V, F] table per node, like the reference's
    shared-memory store (pa_server.py:33-54). world>1: /dev/shm file mapped by every rank."""
    from pagraph_amd.data import synthetic as syn
    if world == 1:
        t0 = time.time()
        if os.environ.get("PG_HOST_TABLE_THP", "1") != "0":
            # (round 5) the table in transparent huge pages, registered with the runtime: the CPU row gather of
            # the miss path is 25-30 % faster out of them (storage.huge_page_tensor, profiles/r05/host_gather_sweep.txt).
            # PG_HOST_TABLE_THP=0: a hipHostMalloc'ed table as in rounds 1-4
            from pagraph_amd.storage import huge_page_tensor
            tab, ok = huge_page_tensor((V, Fdim))
            if ok:
                syn.fill_random_features(tab, device=dev)
                try:
                    thp = [l for l in open("/proc/self/smaps_rollup") if "AnonHugePages" in l][0].split()[1]
                except Exception:
                    thp = "?"
                log(f"[bench] host table {V}x{Fdim} in huge pages (AnonHugePages {thp} kB), registered, filled in {time.time()-t0:.1f}s")
                return tab, True
            del tab
            log("[bench] hipHostRegister of the huge-page table failed -> pinned allocation")
        tab = torch.empty((V, Fdim), dtype=torch.float32, pin_memory=True)
        syn.fill_random_features(tab, device=dev)
        log(f"[bench] host table {V}x{Fdim} pinned + filled in {time.time()-t0:.1f}s")
        return tab, True
    path = f"/dev/shm/pagraph_bench_{os.environ.get('MASTER_PORT', '0')}_{tag}.bin"
    ok = torch.ones(1, dtype=torch.int32, device=dev)
    tab = None
    if local_rank == 0:
        try:
            import glob
            for stale in glob.glob("/dev/shm/pagraph_bench_*.bin"):      # left behind by a crashed run
                if time.time() - os.path.getmtime(stale) > 600:
                    os.unlink(stale)
            tab = torch.from_file(path, shared=True, size=V * Fdim, dtype=torch.float32).view(V, Fdim)
            syn.fill_random_features(tab, device=dev)
        except Exception as e:
            # /dev/shm too small (a container's default is 64 MB) or absent: a file-backed shared mapping in the temp
            # directory is the same single copy in the page cache — never one private copy per rank by accident
            log(f"[bench] rank {rank}: /dev/shm table failed ({e}); trying a file-backed shared mapping")
            try:
                import tempfile
                path = os.path.join(tempfile.gettempdir(), os.path.basename(path))
                tab = torch.from_file(path, shared=True, size=V * Fdim, dtype=torch.float32).view(V, Fdim)
                syn.fill_random_features(tab, device=dev)
            except Exception as e2:
                log(f"[bench] rank {rank}: file-backed table failed too ({e2})")
                ok.zero_()
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    shared = bool(ok.item())
    paths = [path]
    dist.broadcast_object_list(paths, src=0)       # (one node: local rank 0 is rank 0)
    path = paths[0]
    if shared and local_rank != 0:
        tab = torch.from_file(path, shared=True, size=V * Fdim, dtype=torch.float32).view(V, Fdim)
    if not shared:
        private_gb = V * Fdim * 4 * world / 2 ** 30
        budget_gb = 64.0
        if private_gb > budget_gb:
            raise SystemExit(f"bench.py: no shared mapping for the host feature table and {world} private copies would take "
                             f"{private_gb:.0f} GB (> {budget_gb:.0f}): refusing")
        log(f"[bench] rank {rank}: every rank keeps a private copy of the table ({private_gb:.1f} GB in total)")
        tab = torch.empty((V, Fdim), dtype=torch.float32)
        syn.fill_random_features(tab, device=dev)
        path = None
    dist.barrier()
    if shared and local_rank == 0:
        os.unlink(path)          # every rank holds its mapping; nothing is left behind if the run dies
        path = None
    registered = False
    try:
        rc = torch.cuda.cudart().cudaHostRegister(tab.data_ptr(), tab.numel() * 4, 0)
        registered = int(rc) == 0
        log(f"[bench] rank {rank}: hipHostRegister rc={rc}")
    except Exception as e:  # staged mode works from unpinned memory too
        log(f"[bench] rank {rank}: host register unavailable ({e})")
    return tab, registered


# ----------------------------------------------------------------------------- CPU baseline
def cpu_baseline(args, g, sub2full_h, seeds_h, feat_tab, norm_tab, labels_dense_h, steps_per_epoch, budget_s, threads=16, min_steps=0):
    """The reference's CPU path restated, timed on this box's host cores on a
    bounded sample of the same workload: C/OpenMP sampler (oracle) + torch CPU `table[nid_map[ids]]`
    for every NodeFlow row and field (dgl_gcn.py:83 / storage.py:117-131) + H2D + torch-CPU model
    forward/backward/Adam.  16 threads = the reference's sampler num_workers (pa_gcn.py:148)."""
    from oracle import oracle
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(threads)
    except OSError:
        pass
    torch.set_num_threads(threads)
    indptr_h = g.indptr.cpu().numpy()
    indices_h = g.indices.cpu().numpy()
    k, hops, B = args.num_neighbors, 2, args.batch_size
    Fdim, C = args.feat_size, args.n_classes
    hid = 32 if args.model == "gcn" else 16
    torch.manual_seed(0)
    if args.model == "gcn":
        lins = [torch.nn.Linear(Fdim, hid), torch.nn.Linear(2 * hid, C)]
    else:
        lins = [torch.nn.Linear(Fdim, hid), torch.nn.Linear(Fdim, hid), torch.nn.Linear(2 * hid, C),
                torch.nn.Linear(2 * hid, C)]
    params = [p for l in lins for p in l.parameters()]
    opt = torch.optim.Adam(params, lr=3e-2)
    nid_map = torch.from_numpy(sub2full_h)

    def mean_agg(ip, src, h, nd):
        deg = (ip[1:] - ip[:-1]).clamp(min=1).unsqueeze(1).float()
        dst = torch.repeat_interleave(torch.arange(nd), (ip[1:] - ip[:-1]).long())
        out = torch.zeros((nd, h.size(1))).index_add_(0, dst, h[src.long()])
        return out / deg

    t_s = t_l = t_m = 0.0
    done = 0
    t_begin = time.time()
    while done < 64 and ((time.time() - t_begin) < budget_s or done < min_steps) and done * B < len(seeds_h):
        t0 = time.time()
        nf = oracle.sample_nodeflow(indptr_h, indices_h, seeds_h[done * B:(done + 1) * B], k, hops, 0, 0, done)
        t1 = time.time()
        ids = torch.from_numpy(nf["node_mapping"])
        full = nid_map[ids]
        feats = feat_tab[full]                       # the reference's CPU fancy-index (storage.py:128)
        norm = norm_tab[full]
        if torch.cuda.is_available():
            feats_d = feats.cuda(non_blocking=True); norm_d = norm.cuda(non_blocking=True)
            torch.cuda.synchronize()
            del feats_d, norm_d
        t2 = time.time()
        o = nf["layer_offsets"]
        blocks = [(torch.from_numpy(ip.astype(np.int64)), torch.from_numpy(sr)) for ip, sr in nf["blocks"]]
        h = feats[o[0]:o[1]]
        if args.model == "gcn":
            z = lins[0](mean_agg(*blocks[0], h, o[2] - o[1]))
            h1 = torch.cat([z, F.relu(z)], 1)
            pred = lins[1](mean_agg(*blocks[1], h1, o[3] - o[2]))
        else:
            f1, f2 = feats[o[1]:o[2]], feats[o[2]:o[3]]
            z1 = lins[0](f1) + lins[1](mean_agg(*blocks[0], h, o[2] - o[1])); a1 = torch.cat([z1, F.relu(z1)], 1)
            z2 = lins[0](f2) + lins[1](mean_agg(*blocks[1], f1, o[3] - o[2])); a2 = torch.cat([z2, F.relu(z2)], 1)
            pred = lins[2](a2) + lins[3](mean_agg(*blocks[1], a1, o[3] - o[2]))
        y = labels_dense_h[ids[o[2]:o[3]]]
        loss = F.cross_entropy(pred, y)
        opt.zero_grad(); loss.backward(); opt.step()
        t3 = time.time()
        t_s += t1 - t0; t_l += t2 - t1; t_m += t3 - t2
        done += 1
    per_step = (t_s + t_l + t_m) / max(1, done)
    return {"value": per_step * steps_per_epoch, "unit": "s/epoch (extrapolated)", "cores": threads, "kind": "port",
            "cpu_model": host_info()["cpu_model"],
            "sample": f"{done} minibatches of the same workload (B={B}, fanout={k}); per step: "
                      f"sample {t_s/done*1e3:.1f} ms + feature load+H2D {t_l/done*1e3:.1f} ms + model {t_m/done*1e3:.1f} ms",
            "ms_per_step": per_step * 1e3}


# ----------------------------------------------------------------------------- gather micro-benchmark
def gather_microbench(cacher, g, dev, rows_list=(42000, 1 << 18, 1 << 20, 1 << 22)):
    """cached-feature gather (all hits) at the sizes SURVEY 8d names — the real step shape, 262 144, 1 048 576
    and 4 194 304 rows — ids drawn (i) degree-proportionally and (ii) uniformly from the cached set (seed 0).
    HIP events attached to each dispatch on the launch stream. Keys: rows (degree-proportional), (rows, 'uniform')."""
    from pagraph_amd import _lib as L
    lib = L.load()
    names = list(cacher.dims)
    D = sum(cacher.dims.values())
    cached_ids = torch.nonzero(cacher.slot_map >= 0).squeeze(1)
    # degree-proportional draw by inverse CDF (torch.multinomial stops at 2^24 categories; config 5 caches 24.5 M rows)
    cdf = torch.cumsum(g.out_degrees()[cached_ids].double() + 1.0, 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    res = {}
    stream = torch.cuda.current_stream(dev)
    sp = L.stream_ptr(stream)
    def draw(R, dist_name):
        if dist_name == "degree":
            u = torch.rand(R, dtype=torch.float64, device=dev, generator=gen) * cdf[-1]
            return cached_ids[torch.searchsorted(cdf, u).clamp_(max=cached_ids.numel() - 1)].contiguous()
        return cached_ids[torch.randint(0, cached_ids.numel(), (R,), device=dev, generator=gen)].contiguous()

    for R, dist_name in [(R, d) for R in rows_list for d in ("degree", "uniform")]:
        bytes_ = R * (8 * D + 17)
        # A launch that moves less than the 256 MB Infinity Cache re-reads rows (and rewrites a frame) that the previous
        # repetition left there — round 3's 42 K-row figure was a MALL number. Small shapes therefore ROTATE
        # over NS id sets and NS output frames, NS x bytes > 2 x the cache, so that every repetition finds its rows in HBM.
        NS = 1 if bytes_ >= (512 << 20) else min(16, -(-(640 << 20) // bytes_))
        idsets = [draw(R, dist_name) for _ in range(NS)]
        outs = [{n: torch.empty((R, cacher.dims[n]), dtype=torch.float32, device=dev) for n in names} for _ in range(NS)]
        mpos = torch.empty(R, dtype=torch.int32, device=dev)
        mfull = torch.empty(R, dtype=torch.int64, device=dev)
        mcnt = torch.zeros(1, dtype=torch.int32, device=dev)
        slots = torch.empty(R, dtype=torch.int32, device=dev)
        fsets = [L.make_fields((cacher.gpu_fix_cache[n], o[n], cacher.dims[n], cacher.gpu_fix_cache[n].stride(0),
                                o[n].stride(0)) for n in names) for o in outs]
        ml = L.miss_list(mpos, mfull, mcnt)
        call = lambda i, tmr=None: L.check(lib.pg_gather_rows(L.ptr(idsets[i % NS]), R, L.ptr(cacher.slot_map), L.ptr(cacher.nid_map),
                                                              fsets[i % NS][0], fsets[i % NS][1], ctypes.byref(ml), L.ptr(slots),
                                                              None, tmr, None, sp))
        for i in range(max(3, NS)):
            call(i)
        reps = max(20, 2 * NS)
        timers = []
        for i in range(reps):
            t = L.vp(); L.check(lib.pg_timer_create(ctypes.byref(t)))
            call(i, t)
            timers.append(t)
        ms = []
        for t in timers:
            v = ctypes.c_float(); L.check(lib.pg_timer_elapsed_ms(t, ctypes.byref(v))); ms.append(v.value)
            lib.pg_timer_destroy(t)
        assert int(mcnt.item()) == 0
        avg = float(np.mean(ms))
        res[R if dist_name == "degree" else (R, "uniform")] = {
            "rows": R, "ids": dist_name, "avg_ms": avg, "GBps": bytes_ / avg / 1e6, "frac": bytes_ / avg / 1e6 / HBM_PEAK_GBPS,
            "rotating_sets": NS}
        del outs, fsets, idsets
    return res


def gather_launch_stats(cacher, prof, D):
    tries_total, miss_total = cacher._stats.tolist()
    return gather_launch_stats_from(tries_total, miss_total, prof, D)


def gather_launch_stats_from(tries_total, miss_total, prof, D):
    """(avg ms, mean rows, mean misses, mean algorithmic bytes) of the in-loop k_gather launches recorded in `prof`
    (HIP events attached to each dispatch); rows / misses per launch from the device-side counters (padding ids
    of the fixed-shape path do not count; in zero-copy mode the per-launch miss count never reaches the host)"""
    from pagraph_amd import _lib as L
    lib = L.load()
    n_launch = max(1, len(prof))
    ms = []
    share = 1.0
    for entry in prof:
        v = ctypes.c_float()
        L.check(lib.pg_timer_elapsed_ms(entry[0], ctypes.byref(v)))
        lib.pg_timer_destroy(entry[0])
        ms.append(v.value)
        if len(entry) > 3:          # the copy kernel covered only the layers that are read row by row (the leading
            share = entry[3]        # ones are aggregated in place): scale the counters of the whole split to it
    R = tries_total / n_launch * share
    m = miss_total / n_launch * share
    nbytes = (R - m) * 8 * D + R * 17 + m * 12               # algorithmic bytes of one launch
    return (float(np.mean(ms)) if ms else float("nan")), R, m, nbytes


def reference_equivalent_leg(args, model, loss_fcn, optimizer, cacher, g, subtrain, labels, dev, world, rank, steps=100):
    """The work the REFERENCE's fetch_data does per step — every layer, every field, ~42 K rows per launch
    (storage.py:173-204) — through the same pipeline, timed for a short run beside the optimised one, with the
    cache-hit rate counted the reference's way: over every row of every layer, duplicates across layers included
    (storage.py:203-204,219-227)."""
    from pagraph_amd.sampling import NeighborSampler
    from pagraph_amd.trainer import GraphedTrainer, cycle_batches
    B, k = args.batch_size, args.num_neighbors
    sampler = NeighborSampler(g, B, k, neighbor_type='in', shuffle=True, num_workers=16, num_hops=2,
                              seed_nodes=subtrain, prefetch=True, seed=rank + 1000, copy_out=True, static=True,
                              transpose=None if args.no_transpose else 'auto', defer_transpose=not args.inline_transpose)
    tr = GraphedTrainer(model, loss_fcn, optimizer, cacher, sampler, labels, dev, need=None, world_size=world,
                        keep_losses=False)
    S = 3 + 2 * len(sampler.slots)
    WARM = 20
    it = cycle_batches(sampler, S + WARM + steps + 16)
    tr.keep_primed = True
    tr.run_steps(it, S)
    tr.run_steps(it, WARM)                  # steady state before the clock starts, like the main loop's --warmup
    tr.synchronize()
    torch.cuda.synchronize()
    cacher._stats.zero_()
    cacher.profile = []
    mq0_ = cacher.miss_queue_stats()
    wev = [torch.cuda.Event(enable_timing=True)]
    def on_step(done_, loss_):
        if done_ % 20 == 0 or done_ == steps:
            e_ = torch.cuda.Event(enable_timing=True)
            e_.record(tr.compute_stream)
            wev.append(e_)
    tr.on_step = on_step
    wev[0].record(tr.compute_stream)
    t0 = time.time()
    tr.run_steps(it, steps)
    cacher.drain_misses()
    torch.cuda.synchronize()
    dt = time.time() - t0
    tr.on_step = None
    wins = [round(wev[i - 1].elapsed_time(wev[i]) / (min(i * 20, steps) - (i - 1) * 20), 5) for i in range(1, len(wev))]
    prof, cacher.profile = cacher.profile, None
    if cacher.misses_timed_out():
        raise SystemExit("bench.py: reference-equivalent leg: a device-side wait for miss rows timed out")
    D = cacher.total_dim
    avg_ms, R, m, nbytes = gather_launch_stats(cacher, prof, D)
    miss_rate = cacher.get_miss_rate()
    while tr._prepared:                     # hand the ring slots back before the sampler goes away
        tr.sampler.release(tr._prepared.pop(0).nf_cur)
    torch.cuda.synchronize()
    mq1 = cacher.miss_queue_stats()
    moved = (mq1["rows_per_job"] * mq1["jobs"] - mq0_["rows_per_job"] * mq0_["jobs"]) / steps if mq1 and mq0_ else None
    gather_us = ((mq1["us_cpu_gather"] * mq1["jobs"] - mq0_["us_cpu_gather"] * mq0_["jobs"]) / max(1, mq1["jobs"] - mq0_["jobs"])
                 if mq1 and mq0_ else None)
    pcie_ms = moved * 4 * D / 53e6 if moved else None      # the leg's PCIe floor at the 53 GB/s the copies reach
    return {"fetch": "every layer and field (storage.py:173-204)", "steps": steps, "warmup": WARM, "ms_per_step": dt / steps * 1e3,
            "ms_per_step_windows": wins, "ms_per_step_median_window": float(np.median(wins)) if wins else None,
            # which leg of the miss path bounds this step on this host: the worker's CPU row gather per job vs the
            # list's time on PCIe (r02's driver line read 0.468 ms with a 0.6 ms gather on a busy host, the builder's
            # 0.354-0.383 on quiet ones)
            "us_cpu_gather_per_job": gather_us, "pcie_floor_ms_per_step": pcie_ms, "cpu_share": cacher.cpu_share,
            "rows_per_launch": R, "miss_rows_per_launch": m,
            "miss_rows_over_pcie_per_launch_after_index_dedup": moved,
            "cache_hit_pct": 100.0 * (1.0 - miss_rate),
            "gather_avg_launch_ms": avg_ms, "gather_algorithmic_bytes_per_launch": nbytes,
            "gather_GBps": nbytes / avg_ms / 1e6, "gather_frac_of_hbm_peak": nbytes / avg_ms / 1e6 / HBM_PEAK_GBPS}


# ----------------------------------------------------------------------------- N > 1 preflight
def preflight(world, rank, gpu, dev, args):
    """~10 s, N > 1 only (VERDICT r03 #2b): the WHOLE N-rank path on a 300 K-vertex graph before the big set-up — collectives,
    dg on rank 0 + broadcast, per-rank closures, every rank's async miss queue (all SDMA calibrations of the node at once), the
    captures with the all-reduce probe, 40 steps — so that the first multi-GPU box fails in seconds, naming the rank and the
    phase, instead of minutes into the 10 M-vertex set-up. Replicas must be bit-identical afterwards (the data path has no
    collective; the only exchange is the flat gradient's all-reduce) and the loss finite. A phase that HANGS is ended by
    faulthandler after PG_BENCH_PREFLIGHT_TIMEOUT seconds (default 180) with every thread's traceback."""
    import faulthandler
    from pagraph_amd import parallel
    from pagraph_amd.data import synthetic as syn
    from pagraph_amd.model import GCNSampling
    from pagraph_amd.optim import Adam
    from pagraph_amd.partition.dg import dg_raw
    from pagraph_amd.partition.utils import closure_device
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.storage.storage import default_host_threads
    from pagraph_amd.trainer import GraphedTrainer, cycle_batches
    V, E, Fd, C, B, k = 300_000, 3_000_000, 64, 8, 1000, 2
    phase = ["start"]
    t_all = time.time()
    times = {}
    faulthandler.dump_traceback_later(float(os.environ.get("PG_BENCH_PREFLIGHT_TIMEOUT", 180)), exit=True, file=sys.stderr)

    def enter(name):
        times[phase[0]] = round(time.time() - enter.t0, 2)
        phase[0] = name
        enter.t0 = time.time()
        log(f"[preflight] rank {rank}: {name}")
    enter.t0 = time.time()
    try:
        enter("collectives")
        t = torch.full((4,), float(rank + 1), device=dev)
        dist.all_reduce(t)
        assert float(t[0]) == world * (world + 1) / 2, f"all_reduce gave {float(t[0])}"
        b = torch.full((4,), float(rank), device=dev)
        dist.broadcast(b, src=0)
        assert float(b[0]) == 0.0
        enter("graph")
        indptr, indices = syn.rmat_graph(V, E, device=dev)
        train_mask, _, _ = syn.split_dataset(V)
        train_full = torch.nonzero(train_mask).squeeze(1)
        labels_full = syn.random_labels(V, C)
        g_full = DeviceGraph.from_csc(indptr, indices, V)
        enter("dg (rank 0) + broadcast")
        belongs = torch.empty(V, dtype=torch.int8)
        if rank == 0:
            belongs = torch.from_numpy(dg_raw(world, indptr, indices, V, train_full.numpy(), 2, want_r_mask=False)[0])
        belongs = belongs.to(dev)
        parallel.broadcast_tensor(belongs, src=0)
        my_train = torch.nonzero(belongs == rank).squeeze(1).cpu()
        assert my_train.numel() > 0, "dg gave this rank no train vertices"
        enter("closure")
        sub_indptr, sub_indices, sub2full, subtrain = closure_device(g_full, my_train, 2)
        Vs = sub2full.numel()
        g = DeviceGraph.from_csc(sub_indptr, sub_indices, Vs)
        labels = torch.zeros(int(subtrain.max().item()) + 1, dtype=torch.int64, device=dev)
        labels[subtrain] = labels_full.to(dev)[sub2full[subtrain]]
        enter("host table + async miss queue")
        tab = torch.empty((V, Fd), dtype=torch.float32, pin_memory=True)
        syn.fill_random_features(tab, device=dev)
        store = HostFeatureStore({"features": tab}, pin=False, device_visible={"features": True})
        cacher = GraphCacheServer(store, Vs, sub2full, gpu, miss_mode="async", host_threads=min(4, default_host_threads(world)))
        cacher.init_field(["features"])
        if world > torch.cuda.device_count():
            cacher.host_wait = True
        enter("trainer set-up (captures, all-reduce probe)")
        torch.manual_seed(rank)                       # different initial parameters per rank: the broadcast must fix them
        model = GCNSampling(Fd, 16, C, 1, F.relu, 0.0).to(dev)
        opt = Adam(model.parameters(), lr=1e-2)
        sampler = NeighborSampler(g, B, k, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=subtrain, prefetch=True,
                                  seed=rank, static=True, defer_transpose=True)
        steps_per_epoch = parallel.equalize_steps(len(sampler), device=dev)
        tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), opt, cacher, sampler, labels, dev,
                            need=model.required_inputs(3), world_size=world, keep_losses=True)
        tr.after_first_step = lambda: cacher.auto_cache(g, ["features"], cache_ratio=0.3)
        S = 3 + 2 * len(sampler.slots)
        it = cycle_batches(sampler, S + 40 + 8)
        losses = []
        tr.on_step = lambda done_, loss_: losses.append(loss_)
        tr.run_steps(it, S)
        enter("40 replayed steps")
        tr.run_steps(it, 40)
        tr.synchronize()
        torch.cuda.synchronize()
        enter("replica check")
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        identical = bool(torch.equal(lo, hi))
        lv = torch.stack([l_.detach().float().reshape(()) for l_ in losses]).cpu()
        finite = bool(torch.isfinite(lv).all()) and bool(torch.isfinite(flat).all())
        assert identical, "the ranks' parameters differ after 40 all-reduced steps"
        assert finite, "a loss or a parameter is not finite"
        rec = {"ok": True, "seconds": round(time.time() - t_all, 1), "vertices": V, "steps": S + 40, "steps_per_epoch": steps_per_epoch,
               "allreduce_in_graph": bool(tr.allreduce_in_graph), "replicas_identical": identical,
               "loss_first": float(lv[0]), "loss_last": float(lv[-1]), "phase_seconds": None}
        enter("teardown")
        cacher.shutdown_miss_queue()
        del tr, sampler, cacher, store, tab, g, g_full
        torch.cuda.empty_cache()
        dist.barrier()
        enter("done")
        times.pop("start", None)
        rec["phase_seconds"] = times
        log(f"[preflight] rank {rank}: ok in {rec['seconds']} s (all-reduce in graph: {rec['allreduce_in_graph']})")
        return rec
    except BaseException as e:
        log(f"[preflight] rank {rank}: FAILED in phase '{phase[0]}' after {time.time() - t_all:.1f} s: {type(e).__name__}: {e}")
        raise
    finally:
        faulthandler.cancel_dump_traceback_later()


def device_identity(gpu):
    """what tells two ranks' devices apart in the JSON line: name, UUID, PCI bus id (whatever this torch build exposes)"""
    pr = torch.cuda.get_device_properties(gpu)
    rec = {"index": gpu, "name": pr.name, "total_memory_gb": round(pr.total_memory / 2 ** 30, 1)}
    for k_ in ("uuid", "pci_bus_id", "pci_device_id", "gcnArchName", "multi_processor_count"):
        v_ = getattr(pr, k_, None)
        if v_ is not None:
            rec[k_] = str(v_) if k_ == "uuid" else v_
    return rec


# BASELINE.json's other single-GPU configurations, timed inside the driver's own `bench.py --gpus 1` run (they
# used to exist only as builder-run files under profiles/): each is a CHILD run of this script — its own process, its own graph,
# cacher and captured steps, so nothing of the headline run's state leaks into it — whose line is condensed into one entry.
CONFIG_LEGS = (
    # config 2: Reddit's shape (232 965 vertices, mean degree 492, feat 602, 41 classes), table resident in HBM, 2-layer GCN —
    # 26 steps per epoch, ten epochs (examples/profile/pa_gcn.py:144-150 defaults; SURVEY 8d "Config 2")
    ("config2_reddit_shape_full_cache_gcn",
     # ... with BASELINE config 1 beside it (round 6): the reference's CPU plumbing restated on the SAME graph on the host cores —
     # the oracle's sampler + torch CPU table[nid_map[ids]] for every layer and field (dgl_gcn.py:83, storage.py:126-131) + the
     # model on the CPU, 16 threads, at least one epoch's 26 minibatches
     ["--vertices", "232965", "--edges", "57300000", "--feat-size", "602", "--n-classes", "41", "--cache-ratio", "1.0",
      "--model", "gcn", "--steps", "260", "--cpu-baseline-seconds", "4", "--cpu-baseline-min-steps", "26", "--cpu-baseline-one-leg"]),
    # config 3: the headline graph with config 3's own model — GraphSAGE-mean, hidden 16, lr 1e-2 (pa_gs.py:134,141) — and the
    # 30 % hot-degree cache: one whole epoch (1 084 steps)
    ("config3_rmat_10M_graphsage_30pct_cache", ["--model", "graphsage", "--cache-ratio", "0.30"]),
    # config 4 (round 6): the headline graph partitioned by dg x 4 with --num-hops 2 (README.md:117), ONE rank's share — the
    # partition with the largest closure, its cache by LOCAL out-degree (storage.py:100) — through the N > 1 step shape over a
    # one-rank RCCL group, for the step count the four ranks would agree on. A projection of the 4-GPU epoch, not a measurement
    ("config4_rank_of_4", ["--as-rank-of", "4", "--dg-hops", "2", "--cache-ratio", "0.30"]),
    # config 5: 10^8 vertices / 10^9 edges, dg x 8, one rank's share, features in host DRAM (240 GB) with the async miss path
    ("config5_rank_of_8", ["--vertices", "100000000", "--edges", "1000000000", "--as-rank-of", "8", "--cache-ratio", "0.30"]),
)


def config_legs(args, budget_s=420.0):
    import subprocess
    out = {}
    t_all = time.time()
    for name, flags in CONFIG_LEGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--warmup", "10", "--no-configs", "--skip-microbench",
               "--skip-opt-hit", "--skip-reference-equivalent"] + flags
        if "--cpu-baseline-min-steps" not in flags:
            cmd.append("--skip-cpu-baseline")
        if args.host_threads:
            cmd += ["--host-threads", str(args.host_threads)]
        t0 = time.time()
        left = budget_s - (t0 - t_all)
        if left < 20:
            out[name] = {"error": "skipped: the configs block's time budget was spent"}
            continue
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=left)
            if r.returncode != 0:
                raise RuntimeError(f"rc {r.returncode}: {r.stderr[-600:]}")
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rf = d["roofline"]
            out[name] = {
                "workload": d["config"]["workload"], "command": " ".join(["bench.py"] + cmd[2:]),
                "ms_per_step": d["config"]["epoch_ms_per_step"], "steps_timed": d["config"]["epoch_steps_timed"],
                "steps_per_epoch": d["config"]["steps_per_epoch"], "epoch_time_s": d["value"],
                "ms_per_step_window_quantiles": d["ms_per_step_window_quantiles"],
                "cache_hit_pct": d["cache_hit_pct"], "miss_mode": d["config"]["miss_mode"],
                "early_layer0_aggregation": d["config"]["early_layer0_aggregation"],
                "trained": d["trained"], "misses_timed_out": d["misses_timed_out"],
                "host_issue_ms_per_step": d["host_issue_ms_per_step"],
                "cpus_used": (d["host"]["timed_region_cgroup"] or {}).get("cpus_used"),
                "roofline": {k_: rf.get(k_) for k_ in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                                       "launches_timed", "algorithmic_bytes_per_launch", "kernel_body_ms")},
                "leg_wall_s": time.time() - t0}
            if d.get("cpu_baseline"):
                out[name]["cpu_baseline"] = d["cpu_baseline"]        # (config 1: the CPU plumbing on this leg's graph)
            if d["config"].get("rank_of_P"):
                out[name]["rank_of_P"] = d["config"]["rank_of_P"]
                out[name]["step_shape"] = d["config"]["step_shape"]
                out[name]["step_replay"] = d["config"]["step_replay"]
                out[name]["allreduce_in_graph"] = d["config"]["allreduce_in_graph"]
                out[name]["projected_epoch_s"] = d["config"]["rank_of_P"]["projected_epoch_s"]
        except Exception as e:                      # a leg that fails must not take the headline line with it
            out[name] = {"error": f"{type(e).__name__}: {str(e)[-800:]}", "leg_wall_s": time.time() - t0}
        log(f"[bench] configs block: {name} -> {json.dumps(out[name])[:300]}")
    return out


def main():
    # the JSON line is the ONLY thing on stdout: the library's reference-style prints
    # ('total dims', 'Cache Memory', ...) go to stderr
    real_stdout = sys.stdout
    sys.stdout = sys.stderr
    # ... also what C libraries print: RCCL writes its version banner to the C stdout, which libc flushes at exit — BEHIND the
    # JSON line when stdout is a pipe. File descriptor 1 points at stderr for the whole run and comes back for the line.
    sys.stdout.flush()
    saved_fd1 = os.dup(1)
    os.dup2(2, 1)
    try:
        out = run()
        if out is not None and out.pop("_wants_configs", False):
            # the headline run's device memory, pinned table and worker threads are gone with run()'s frame
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["configs"] = config_legs(parse())
    finally:
        try:
            ctypes.CDLL(None).fflush(None)           # whatever C code buffered for "stdout" goes to stderr now
        except Exception:
            pass
        os.dup2(saved_fd1, 1)
        os.close(saved_fd1)
        sys.stdout = real_stdout
    if out is not None:
        print(json.dumps(out), flush=True)


def run():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: pagraph_amd has no CPU fallback")
    gpu = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    # ---- one rank's share of a P-GPU job on this one GPU (--as-rank-of P) / the N > 1 step shape alone (--dist-step) ----
    emul_P = int(args.as_rank_of) if world == 1 else 0
    dist_step = bool(world == 1 and (emul_P > 1 or args.dist_step))
    step_world = world if world > 1 else (max(2, emul_P) if dist_step else 1)     # what the trainer is told
    if dist_step:
        import socket
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port_ = s_.getsockname()[1]; s_.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_), RANK="0", WORLD_SIZE="1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1)      # REAL RCCL, of one rank: all a one-GPU box can host

    from pagraph_amd import _lib as L
    from pagraph_amd import parallel
    from pagraph_amd.data import synthetic as syn
    from pagraph_amd.model import GCNSampling, GraphSageSampling
    import importlib
    dgmod = importlib.import_module("pagraph_amd.partition.dg")
    dg_raw = dgmod.dg_raw
    from pagraph_amd.partition.utils import closure_device
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.storage.storage import default_host_threads
    from pagraph_amd.trainer import GraphedTrainer, MinibatchTrainer, cycle_batches
    L.load()
    dist_rec = None
    if world > 1:
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl = None
        dist_rec = {"world_size": dist.get_world_size(), "backend": str(dist.get_backend()), "rccl_version": rccl,
                    "gpus_visible": torch.cuda.device_count(), "ranks_share_a_gpu": world > torch.cuda.device_count()}
        log(f"[bench] rank {rank}/{dist_rec['world_size']} on {device_identity(gpu)} backend {dist_rec['backend']} RCCL {rccl}")
        if not args.skip_preflight:
            dist_rec["preflight"] = preflight(world, rank, gpu, dev, args)

    V, E, Fdim, C, B, k = args.vertices, args.edges, args.feat_size, args.n_classes, args.batch_size, args.num_neighbors
    if args.dg_hops is None:
        args.dg_hops = 2        # README.md:117: --num-hops of a 2-layer model without preprocessing (BASELINE configs 4 and 5)
    n_layers = 1
    num_hops = n_layers + 1                                   # pa_gcn.py:52
    hidden = 32 if args.model == "gcn" else 16                # pa_gcn.py:130 / pa_gs.py:134
    lr = 3e-2 if args.model == "gcn" else 1e-2                # pa_gcn.py:137 / pa_gs.py:141

    # ---- dataset (preprocess.py recipe, seeded) ------------------------------------------
    t0 = time.time()
    indptr, indices = syn.rmat_graph(V, E, device=dev)
    torch.cuda.synchronize()
    log(f"[bench] rank {rank}: RMAT V={V} nnz={indices.numel()} in {time.time()-t0:.1f}s")
    train_mask, _, _ = syn.split_dataset(V)
    labels_full = syn.random_labels(V, C)
    train_full = torch.nonzero(train_mask).squeeze(1)
    g_full = DeviceGraph.from_csc(indptr, indices, V)
    in_deg = (indptr[1:] - indptr[:-1]).float()

    # what dg.py --num-hops 2 (README.md:117: the value for a 2-layer model) has to walk on this graph: the two-hop in-neighbourhood
    # of every train vertex = sum(deg^2) adjacency entries. Round 6: the device builds those sets (pg_dg_partition_gpu, ~1.4e10
    # entries per second: 8.5 s on the 10M / 100M graph, 133 s on config 5's — profiles/r06/dg_gpu_*.json); round 3's host team
    # took 68 s at 10M through one committer and was never run at 10^8 (~1000 s).
    deg_d = (indptr[1:] - indptr[:-1]).double()
    sum_deg_sq = float((deg_d * deg_d).sum().item())
    del deg_d
    # ---- partition ------------------------------------------------------------------------
    t0 = time.time()
    share_rec_p = None
    if world == 1 and emul_P > 1:
        # dg over P partitions exactly as rank 0 of a P-GPU run does it (dg.py:59-103), then EVERY partition's closure — their
        # sizes are what the question is about — and the step count the P ranks would agree on (parallel.equalize_steps = MAX)
        t_dg = time.time()
        # (the synthetic graph is a pure function of V and E: a second run on the same box — the profiling passes — reuses dg's
        # result; the record says so)
        import tempfile
        dg_file = os.path.join(tempfile.gettempdir(), f"pagraph_bench_dg_{V}_{E}_P{emul_P}_h{args.dg_hops}.npz")
        dg_cached = os.path.exists(dg_file)
        dg_stats = None
        if dg_cached:
            z_ = np.load(dg_file)
            b, p_vnum, r_vnum, dg_s = z_["belongs"], z_["p_vnum"], z_["r_vnum"], float(z_["seconds"])
        else:
            b, _, p_vnum, r_vnum = dg_raw(emul_P, indptr, indices, V, train_full.numpy(), args.dg_hops, want_r_mask=False)
            dg_s = time.time() - t_dg
            dg_stats = dgmod.LAST_GPU_STATS
            try:
                np.savez(dg_file, belongs=b, p_vnum=p_vnum, r_vnum=r_vnum, seconds=dg_s)
            except OSError:
                pass
        belongs_d = torch.from_numpy(b).to(dev)
        log(f"[bench] dg P={emul_P} hops={args.dg_hops}: {dg_s:.1f}s p_vnum={p_vnum.tolist()} r_vnum={r_vnum.tolist()}")
        parts = []
        t_cl = time.time()
        for r_ in range(emul_P):
            tr_ = torch.nonzero(belongs_d == r_).squeeze(1)
            _ip, _ix, s2f_, st_ = closure_device(g_full, tr_.cpu(), num_hops)
            parts.append({"rank": r_, "train_vertices": int(tr_.numel()), "partition_vertices": int(s2f_.numel()),
                          "partition_nnz": int(_ix.numel()), "steps": -(-int(st_.numel()) // B)})
            del _ip, _ix, s2f_, st_
        cl_s = time.time() - t_cl
        which = args.which_rank if args.which_rank is not None else max(parts, key=lambda e: e["partition_vertices"])["rank"]
        eq_steps = max(e["steps"] for e in parts)
        share_rec_p = {"P": emul_P, "dg_hops": args.dg_hops, "dg_seconds": dg_s,
                       "dg_on": "device-assisted (pg_dg_partition_gpu)" if dg_stats else ("reused" if dg_cached else "host (pg_dg_partition_mt)"),
                       "dg_device_stats": dg_stats, "dg_result_reused_from_an_earlier_run_on_this_box": bool(dg_cached),
                       "closures_seconds": cl_s, "p_vnum": p_vnum.tolist(),
                       "r_vnum": r_vnum.tolist(), "partitions": parts, "rank_run": which,
                       "rank_chosen_by": "--which-rank" if args.which_rank is not None else "largest closure",
                       "equalised_steps_per_epoch": eq_steps, "equalisation": "MAX over the P partitions' own step counts "
                       "(parallel.equalize_steps); a rank that runs out of seeds wraps around"}
        my_train = torch.nonzero(belongs_d == which).squeeze(1).cpu()
        del belongs_d
    elif world == 1:
        my_train = train_full
    else:
        belongs = torch.empty(V, dtype=torch.int8)
        log(f"[bench] rank {rank}: dg P={world} hops={args.dg_hops} on rank 0 (neighbour sets built on its GPU: ~4 s at 10M / 100M, "
            f"~1 min at 10^8 / 10^9 with hops 2); the other ranks wait in a broadcast meanwhile")
        if rank == 0:
            b, _, p_vnum, r_vnum = dg_raw(world, indptr, indices, V, train_full.numpy(), args.dg_hops, want_r_mask=False)
            belongs = torch.from_numpy(b)
            log(f"[bench] dg P={world} hops={args.dg_hops}: {time.time()-t0:.1f}s p_vnum={p_vnum.tolist()} r_vnum={r_vnum.tolist()}")
        belongs = belongs.to(dev)
        parallel.broadcast_tensor(belongs, src=0)
        my_train = torch.nonzero(belongs == rank).squeeze(1).cpu()
    sub_indptr, sub_indices, sub2full, subtrain = closure_device(g_full, my_train, num_hops)
    Vs = sub2full.numel()
    del g_full, indptr, indices
    torch.cuda.empty_cache()
    g = DeviceGraph.from_csc(sub_indptr, sub_indices, Vs)
    # the reference's cache-size rule (storage.py:70-84) subtracts the PEAK allocation; building the synthetic
    # graph on the GPU is not part of the trainer process it was written for
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    log(f"[bench] rank {rank}: closure V_sub={Vs} nnz_sub={sub_indices.numel()} train={subtrain.numel()} in {time.time()-t0:.1f}s")
    # labels in local-id space (pa_gcn.py:38-41)
    labels = torch.zeros(int(subtrain.max().item()) + 1, dtype=torch.int64, device=dev)
    labels[subtrain] = labels_full.to(dev)[sub2full[subtrain]]

    # ---- feature provider (pa_server.py:38-54) --------------------------------------------
    feat_tab, table_device_visible = make_host_table(V, Fdim, rank, local_rank, world, dev, "feat")
    # one GPU or many: the async queue (worker thread + one calibrated SDMA engine) unless told otherwise. The
    # zero-copy / async comparison that used to run at the start of every multi-GPU run is now opt-in.
    probe_modes = args.probe_miss_mode
    if args.miss_mode is None:
        args.miss_mode = "zerocopy" if probe_modes else "async"
    if probe_modes:
        args.miss_mode = "zerocopy"                          # start here; the async path is tried after the warm-up
    if not table_device_visible and args.miss_mode == "zerocopy":
        log(f"[bench] rank {rank}: host table is not device-addressable -> staged miss path")
        args.miss_mode = "staged"                       # a zero-copy read of unregistered memory would fault
    fields = {"features": feat_tab}
    embed_names = ["features"]
    norm_tab = None
    if args.model == "gcn":
        norm_tab = (1.0 / in_deg).unsqueeze(1).cpu().pin_memory()   # pa_server.py:43 (inf for isolated vertices, as there)
        fields["norm"] = norm_tab
        embed_names = ["features", "norm"]                    # pa_gcn.py:46
    store = HostFeatureStore(fields, pin=False,               # both tables are already pinned / registered
                             device_visible={"features": table_device_visible, "norm": True})
    cacher = GraphCacheServer(store, Vs, sub2full, gpu, miss_mode=args.miss_mode,
                              host_threads=args.host_threads or default_host_threads(world))
    # (--as-rank-of P: the rank gets THIS box's one-GPU share of host threads — a one-GPU slice of a node; a P-GPU node has to
    # bring P times that for the projection to hold. --host-threads 2 shows the rank on a starved host.)
    cacher.init_field(embed_names)
    cacher.log = True
    if world > torch.cuda.device_count():
        # several ranks on ONE GPU (the gloo path check of the tests / profiles): their device-side waits — a one-wave kernel
        # spinning for up to 3 s — share the device's hardware queues and scheduler time slices with the other rank's streams.
        # Twice this round such a run saw a wait give up although worker and copy were healthy (never on a GPU of its own, in
        # several hundred runs). Ranks that share a device order their consumers after the miss rows on the host instead.
        cacher.host_wait = True
        log(f"[bench] rank {rank}: {world} ranks on {torch.cuda.device_count()} GPU(s) -> host-side waits for miss rows")
    adapt_share = args.cpu_share is None and not args.no_adapt_cpu_share
    cacher.cpu_share = 1.0 if args.cpu_share is None else args.cpu_share
    D = cacher.total_dim
    PROF_RING = 1 << 12          # launches the fused kernel's self-timing ring keeps (2176 bytes each)
    fuse_gather = not args.no_graph and not args.fetch_all and not args.no_fuse_gather
    if fuse_gather:        # the fused gather+aggregate kernel stamps its own start / end / edge count per launch
        cacher.rows_prof = (torch.zeros(L.PG_PROF_WORDS * PROF_RING, dtype=torch.int64, device=dev), PROF_RING)
        if os.environ.get("PG_BENCH_STAMP_SUCCESSOR"):
            # profiling runs only (one more launch per step): a one-thread marker kernel right behind the fused kernel — its
            # stamp ties the stamps' clock to rocprofv3's (tools/join_stamps_trace.py)
            cacher.rows_prof += (True,)

    # ---- model ------------------------------------------------------------------------------
    torch.manual_seed(rank)
    if args.model == "gcn":
        model = GCNSampling(Fdim, hidden, C, n_layers, F.relu, 0.2, False)
    else:
        model = GraphSageSampling(Fdim, hidden, C, n_layers, F.relu, 0.2, 'mean', False)
    model = model.to(dev)
    loss_fcn = torch.nn.CrossEntropyLoss()
    use_graph = not args.no_graph
    if use_graph:
        from pagraph_amd.optim import Adam                    # torch.optim.Adam's arithmetic in one launch
        optimizer = Adam(model.parameters(), lr=lr, weight_decay=0)
    else:
        optimizer = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0, fused=True)
    need = None if args.fetch_all else model.required_inputs(num_hops + 1)
    if world > 1 and not use_graph:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu])
    sampler = NeighborSampler(g, B, k, neighbor_type='in', shuffle=True, num_workers=16, num_hops=num_hops,
                              seed_nodes=subtrain, prefetch=True, seed=rank, copy_out=True, static=use_graph,
                              ring=args.ring if args.ring else (args.lookahead + 3 if args.lookahead else None),
                              transpose=None if args.no_transpose else 'auto',
                              defer_transpose=use_graph and not args.inline_transpose)
    steps_per_epoch = parallel.equalize_steps(len(sampler), device=dev)
    if share_rec_p is not None:
        share_rec_p["own_steps_per_epoch"] = int(steps_per_epoch)
        steps_per_epoch = int(share_rec_p["equalised_steps_per_epoch"])       # what all P ranks would run (cycle_batches wraps)
    # default: one whole epoch of this rank's seeds (1084 steps at N = 1: a quarter of a second), nothing extrapolated
    K = args.steps if args.steps is not None else min(steps_per_epoch, 5000)
    W = args.warmup
    S = 1                                                # set-up steps: the cache is filled after the first one
    if args.extra_streams:
        # diagnosis (tools/exp_hw_queues2.sh): a few more streams that have run one kernel each — what torch.distributed's NCCL
        # stream and a communication stream add to the process — to show what the hardware-queue count does to the step
        _xs = [torch.cuda.Stream(device=dev) for _ in range(args.extra_streams)]
        for x_ in _xs:
            with torch.cuda.stream(x_):
                torch.zeros(8, device=dev).add_(1)
        torch.cuda.synchronize()
    if use_graph:
        trainer = GraphedTrainer(model, loss_fcn, optimizer, cacher, sampler, labels, dev, need=need, world_size=step_world,
                                 keep_losses=False, lookahead=args.lookahead)
        S = 3 + 2 * len(sampler.slots)                   # eager warm-up + one capture and first replay per ring slot
        trainer.keep_primed = not args.cold_start
        trainer.fuse_gather = fuse_gather
        if args.no_fuse_partials:
            trainer.fuse_partials = False
        trainer.tape_collectives = bool(args.tape_collectives)
    else:
        trainer = MinibatchTrainer(model, loss_fcn, optimizer, cacher, sampler, labels, dev, overlap=not args.no_overlap,
                                   need=need)
    PRESAMPLE_SEED = 7919            # the presampled epoch is NOT the epoch that is trained on and timed (another seed)

    def presampled_freq():
        from pagraph_amd import analysis
        t_ = time.time()
        probe_ = NeighborSampler(g, B, k, neighbor_type='in', shuffle=True, num_hops=num_hops, seed_nodes=subtrain,
                                 prefetch=True, seed=rank + PRESAMPLE_SEED)
        f_, _ = analysis.access_frequency(probe_, layers=None if need is None else set(need), epochs=args.presample_epochs)
        del probe_
        log(f"[bench] rank {rank}: presampled {args.presample_epochs} epoch(s) for the cache policy in {time.time()-t_:.2f}s")
        return f_

    def fill_cache():
        if args.cache_policy == "presample" and args.cache_ratio < 1.0:
            cacher.auto_cache(g, embed_names, cache_ratio=args.cache_ratio, policy="presample", freq=presampled_freq())
        else:
            cacher.auto_cache(g, embed_names, cache_ratio=args.cache_ratio)
    trainer.after_first_step = fill_cache                 # pa_gcn.py:99-100
    model.train()
    PROBE = 40
    it = cycle_batches(sampler, 4 * S + W + K + steps_per_epoch + 64 + (4 * PROBE + 64 if probe_modes else 0))

    # ---- set-up (untimed, one-off: the cache is filled after its first step, as in the reference; graph capture) ----
    t0 = time.time()
    trainer.run_steps(it, S)
    if cacher.cached_num == 0:
        fill_cache()
    torch.cuda.synchronize()
    log(f"[bench] rank {rank}: set-up {S} steps + cache fill ({cacher.cached_num} rows) in {time.time()-t0:.1f}s")
    # safety net: a device-side wait for miss rows that timed out during set-up means the runtime put the consuming
    # stream and the copy stream on one hardware queue (seen when every stream has the same priority). Switch to
    # host-side waits (no spin kernel), rebuild the queue and repeat the set-up instead of aborting the run.
    stuck = 1.0 if (use_graph and cacher.misses_timed_out()) else 0.0
    if world > 1:
        stuck = parallel.max_over_ranks(stuck, device=dev)
    if stuck:
        log(f"[bench] rank {rank}: a device-side wait for miss rows timed out during set-up -> host-side waits "
            f"(miss queue: {cacher.miss_queue_stats() if cacher.miss_mode == 'async' else None})")
        # (not trainer.synchronize(): it ends with check_misses(), which raises on the very flag this branch handles)
        cacher.drain_misses()
        torch.cuda.synchronize()
        while trainer._prepared:
            sampler.release(trainer._prepared.pop(0).nf_cur)
        cacher.shutdown_miss_queue()
        cacher.host_wait = True
        trainer.run_steps(it, S)
        torch.cuda.synchronize()
    # ---- every rank sizes its CPU leg of the miss path from what its host can really do (VERDICT r02 #1a) ----
    share_rec = None
    adapt_here = bool(adapt_share and use_graph and cacher.miss_mode == "async" and not cacher.full_cached and table_device_visible)
    if (bool(parallel.max_over_ranks(1.0 if adapt_here else 0.0, device=dev)) if world > 1 else adapt_here):
        trainer.run_steps(it, 12)                          # a dozen steady-state jobs for the worker's counters
        trainer.synchronize()
        share_rec = cacher.adapt_cpu_share(min_jobs=8, apply=False) if adapt_here else None
        # EVERY step contains a gradient all-reduce when world > 1: the ranks must run the same NUMBER of steps whatever each
        # of them decides about its own split. Until round 4 the second window and the re-capture steps below ran only on the
        # ranks whose own measurement asked for them — a rank that disagreed with its peers left them waiting in an all-reduce
        # for ever (the "two ranks on one GPU" time-outs of rounds 3 and 4: 1 run in 5 once the faster CPU gather made the
        # ranks' measurements straddle the threshold; `tools/hunt_two_ranks.sh`). The decisions are now agreed (max over ranks).
        def anywhere(flag):
            return bool(parallel.max_over_ranks(1.0 if flag else 0.0, device=dev)) if world > 1 else bool(flag)
        wants = bool(share_rec and share_rec["cpu_share"] != share_rec["cpu_share_before"])
        if anywhere(wants):
            # A busy moment of a shared host reads like a starved one (profiles/r03: one collection measured 0.075 us per
            # row in this window, 0.041 right after, and ran the whole epoch 9 % slower with 44 % of the rows read by
            # the device). Leaving the all-CPU path takes two windows in a row that say so; the milder of the two wins.
            trainer.run_steps(it, 24)
            trainer.synchronize()
            second = cacher.adapt_cpu_share(min_jobs=8, quiet=True, apply=False)
            if wants:
                share_rec["first_window"] = {k_: share_rec[k_] for k_ in ("us_per_row_cpu_gather", "cpu_share")}
                if second:
                    share_rec["us_per_row_cpu_gather"] = second["us_per_row_cpu_gather"]
                    share_rec["cpu_share"] = max(share_rec["cpu_share"], second["cpu_share"])
                if share_rec["cpu_share"] != share_rec["cpu_share_before"]:
                    cacher.apply_cpu_share(share_rec["cpu_share"])
                log(f"[bench] rank {rank}: cpu_share {share_rec['cpu_share_before']} -> {share_rec['cpu_share']} "
                    f"(windows: {share_rec['first_window']['cpu_share']}, {second['cpu_share'] if second else None})")
        changed = bool(share_rec and share_rec["cpu_share"] != share_rec["cpu_share_before"])
        if anywhere(changed):
            trainer.run_steps(it, S)                       # new fetch plans -> one re-capture per ring slot (where it changed)
            trainer.synchronize()
            if changed:
                again = cacher.adapt_cpu_share(min_jobs=8, quiet=True, apply=False)   # what the new split measures; not chased
                share_rec["after"] = {"us_per_row_cpu_gather": again["us_per_row_cpu_gather"],
                                      "cpu_share_it_would_pick_now": again["cpu_share"]} if again else None
        torch.cuda.synchronize()
    # ---- warm-up: W steady-state steps, untimed ----
    if W > 0:
        trainer.run_steps(it, W)
        torch.cuda.synchronize()
    # ---- which miss path? (multi-GPU default) ------------------------------------------------
    # The async queue is the faster path on one GPU, but it is sensitive to how many streams are busy (a fifth one
    # made it 1.5-3x slower) and RCCL brings its own; that cannot be tried from the build container, so with
    # --gpus > 1 both paths are timed here, on the real system, and every rank keeps the faster one.
    mode_probe = None
    if probe_modes and use_graph and cacher.miss_mode == "zerocopy" and not cacher.full_cached:
        def timed(n):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t_ = time.time()
            trainer.run_steps(it, n)
            torch.cuda.synchronize()
            return parallel.max_over_ranks(time.time() - t_, device=dev) / n * 1e3
        t_zero = timed(PROBE)
        cacher.miss_mode, trainer.lookahead = "async", 2
        trainer.run_steps(it, 16)                              # pipeline refill, miss-queue creation
        t_async = timed(PROBE)
        mode_probe = {"zerocopy_ms_per_step": t_zero, "async_ms_per_step": t_async}
        if t_zero < t_async:
            cacher.miss_mode, trainer.lookahead = "zerocopy", 1
            trainer.run_steps(it, 8)                           # the batches prepared under the async path drain
            torch.cuda.synchronize()
            cacher.shutdown_miss_queue()                       # no worker / gather threads left behind
        args.miss_mode = cacher.miss_mode
        log(f"[bench] rank {rank}: miss path probe zerocopy {t_zero:.3f} ms/step, async {t_async:.3f} ms/step -> {args.miss_mode}")

    # ---- timed regions ------------------------------------------------------------------------
    tl_state = {"armed": bool(args.timeline and use_graph)}

    def timed_region(K_, tag):
        """exactly K_ steps between (barrier +) device-wide syncs; everything the line reports about a region"""
        cacher._stats.zero_()                                    # the reference's try / miss counters, per region
        cacher.profile = []
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tl = None
        if tl_state["armed"]:
            tl_state["armed"] = False
            tl = []
            _prep, _comp = trainer.prepare, trainer.compute
            def prep(nf):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(trainer.load_stream); s_ = _prep(nf); e1.record(trainer.load_stream)
                tl.append(("load", e0, e1)); return s_
            def comp(s_):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                trainer.compute_stream.wait_event(s_.ready)
                e0.record(trainer.compute_stream); r_ = _comp(s_); e1.record(trainer.compute_stream)
                tl.append(("comp", e0, e1)); return r_
            trainer.debug_events = []
            trainer.prepare, trainer.compute = prep, comp
            hostlog = []
            def wrap(obj, name, tag_):
                f = getattr(obj, name)
                def g_(*a, **k):
                    t0_ = time.perf_counter(); r_ = f(*a, **k); hostlog.append((tag_, t0_, time.perf_counter())); return r_
                setattr(obj, name, g_)
            samp_ev = []
            _enq = sampler._enqueue
            def enq(b_, e_):
                a0 = torch.cuda.Event(enable_timing=True); a1 = torch.cuda.Event(enable_timing=True); a2 = torch.cuda.Event(enable_timing=True)
                a0.record(sampler.stream)
                sl = sampler.slots[sampler._ring_pos % len(sampler.slots)]
                if sl.free_recorded:
                    sampler.stream.wait_event(sl.free)
                a1.record(sampler.stream)
                r_ = _enq(b_, e_); a2.record(sampler.stream); samp_ev.append((a0, a1, a2)); return r_
            sampler._enqueue = enq
            wrap(cacher, "wait_misses", "wait_misses"); wrap(cacher, "fetch_data", "fetch_data"); wrap(sampler, "_enqueue", "sampler_enqueue")
            wrap(sampler, "release", "release")
        prof_host = None
        if args.profile_host and tag == "window":
            import cProfile
            prof_host = cProfile.Profile()
            prof_host.enable()
        # per-window step times: one event on the compute stream every `window` steps (first one = start of the region)
        win = max(1, min(args.window, K_))
        cstream = trainer.compute_stream if use_graph else torch.cuda.current_stream(dev)
        wev = [torch.cuda.Event(enable_timing=True)]
        host_t = []
        loss_ev = {}
        p_before = torch.cat([p_.detach().reshape(-1) for p_ in model.parameters()]).clone()
        def on_step(done_, loss_):
            host_t.append(time.perf_counter())
            if done_ == 1 or done_ == K_:   # evidence that the timed steps trained: a private copy of two loss values
                loss_ev[done_] = loss_.detach().clone()          # (the slot's static loss tensor is overwritten 8 steps later)
            if done_ % win == 0 or done_ == K_:
                e_ = torch.cuda.Event(enable_timing=True)
                e_.record(cstream)
                wev.append(e_)
        trainer.on_step = on_step
        if getattr(trainer, "launch_trace", None) is not None:
            trainer.launch_trace = []
        mq0 = cacher.miss_queue_stats()
        cacher.miss_queue_longest(reset=True)
        cg0 = cgroup_cpu_stat()
        pt0 = time.process_time()            # CPU time of THIS process, all its threads (the cgroup's figure covers every rank)
        drop_step0 = int(model._drop_step.item()) if hasattr(model, "_drop_step") else 0
        early0 = getattr(trainer, "early_ordinal", 0)
        wev[0].record(cstream)
        t0 = time.time()
        done = trainer.run_steps(it, K_)
        t_issued = time.time() - t0          # launch thread done; ~= elapsed when the host is the bottleneck
        cacher.drain_misses()                # worker's outstanding copies enqueued (host-side wait, no HIP call) ...
        torch.cuda.synchronize()             # ... then the device
        if world > 1:
            dist.barrier()
        elapsed = time.time() - t0
        trainer.on_step = None
        assert done == K_, (done, K_)
        cg1 = cgroup_cpu_stat()
        cpu_quota = None
        if cg0 and cg1:      # was the process throttled by its CPU quota inside the region? how many CPUs did it use?
            cpu_quota = {"nr_throttled": cg1[0] - cg0[0], "throttled_ms": (cg1[1] - cg0[1]) / 1e3,
                         "cpus_used": (cg1[2] - cg0[2]) / 1e6 / max(1e-9, elapsed)}
        if cpu_quota is None:
            cpu_quota = {}
        cpu_quota["process_cpus_used"] = (time.process_time() - pt0) / max(1e-9, elapsed)     # this rank alone
        # FULL windows only: a region's last, shorter window (4 steps of a 1 084-step epoch) is reported apart. It is the
        # pipeline running dry, not a steady state: the miss rows of the look-ahead batches have landed before their steps
        # start, so those steps run at the table-cached time (r04's unexplained 0.098 ms/step 'window' beside 0.150-0.165).
        windows, tail_window = [], None
        for i_ in range(1, len(wev)):
            n_ = min(i_ * win, K_) - (i_ - 1) * win
            ms_ = round(wev[i_ - 1].elapsed_time(wev[i_]) / max(1, n_), 5)
            if n_ == win:
                windows.append(ms_)
            else:
                tail_window = {"steps": int(n_), "ms_per_step": ms_}
        # the launch thread's three longest iterations (ms, step index): a stall of the host shows here, one of the GPU /
        # miss path only in the windows
        hd = np.diff(np.asarray(host_t)) * 1e3 if len(host_t) > 1 else np.zeros(0)
        host_longest = [[round(float(hd[i_]), 3), int(i_) + 1] for i_ in np.argsort(-hd)[:3]] if len(hd) else []
        launch_split = None
        if getattr(trainer, "launch_trace", None):
            # PG_TRACE_LAUNCH=1: [sample + prepare, compute, release] ms of the trainer's three longest iterations
            lt = np.asarray(trainer.launch_trace)
            launch_split = [[int(i_)] + [round(float(x), 3) for x in lt[i_]] for i_ in np.argsort(-lt.sum(1))[:3]]
        timed_out = bool(cacher.misses_timed_out())
        copy_windows = None
        if os.environ.get("PG_MISSQ_COPYLOG"):
            cl = cacher.miss_copy_log()[-K_:]
            if os.environ.get("PG_MISSQ_COPYLOG") == "2":
                log("[copylog] " + " ".join(f"{b / max(1e-9, m) / 1e6:.0f}" for b, m in cl))
            copy_windows = [round(sum(b for b, _ in cl[i:i + win]) / max(1e-9, sum(m for _, m in cl[i:i + win])) / 1e6, 2)
                            for i in range(0, len(cl), win)]       # GB/s of the H2D copies per window
        mq_stats = cacher.miss_queue_stats()
        if mq_stats and mq0:
            tr_ = {k_: mq_stats[k_] - mq0[k_] for k_ in ("jobs", "waits_by_event", "waits_by_spin_kernel", "rescued_chunks", "spared_jobs")}
            # rows the worker really moved over PCIe (after the miss list's index dedup) per step of the region
            tr_["rows_over_pcie_per_step"] = (mq_stats["rows_per_job"] * mq_stats["jobs"] - mq0["rows_per_job"] * mq0["jobs"]) / max(1, K_)
            tr_["us_cpu_gather"] = ((mq_stats["us_cpu_gather"] * mq_stats["jobs"] - mq0["us_cpu_gather"] * mq0["jobs"])
                                    / max(1, tr_["jobs"]))
            tr_["longest_us"] = cacher.miss_queue_longest()      # where a stall of the miss path sat (one job's worst phase)
            mq_stats["timed_region"] = tr_
        if timed_out:
            raise SystemExit("bench.py: the async miss queue's device-side wait timed out (worker thread dead?) — "
                             "the timed steps trained on rows that never landed; no number is reported")
        if tl and getattr(trainer, "debug_events", None):
            for ev in trainer.debug_events[30:42]:
                log(f"[load-stream] wait(sampler ready) {ev[0].elapsed_time(ev[1])*1e3:8.1f} us | wait(slot done) {ev[1].elapsed_time(ev[2])*1e3:8.1f} us | work {ev[2].elapsed_time(ev[3])*1e3:8.1f} us")
        if tl:
            for a0, a1, a2 in samp_ev[30:40]:
                log(f"[sampler-stream] wait(slot free) {a0.elapsed_time(a1)*1e3:8.1f} us | sampling kernels {a1.elapsed_time(a2)*1e3:8.1f} us")
            hb = hostlog[len(hostlog) // 2][1]
            for tag_, a_, b_ in hostlog[len(hostlog) // 2: len(hostlog) // 2 + 40]:
                log(f"[host] {tag_:16s} +{(a_-hb)*1e6:9.1f} us  took {(b_-a_)*1e6:8.1f} us")
            base = tl[20][1]
            for name, e0, e1 in tl[20:60]:
                log(f"[timeline] {name} start {base.elapsed_time(e0)*1e3:9.1f} us  dur {e0.elapsed_time(e1)*1e3:8.1f} us")
        if prof_host is not None:
            import pstats
            prof_host.disable()
            pstats.Stats(prof_host, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
        elapsed = parallel.max_over_ranks(elapsed, device=dev)
        prof, cacher.profile = cacher.profile, None
        tries_total, miss_total = cacher._stats.tolist()          # accumulated on the device by k_split
        miss_rate = cacher.get_miss_rate() if tries_total else 0.0
        drop_step1 = int(model._drop_step.item()) if hasattr(model, "_drop_step") else 0
        early1 = getattr(trainer, "early_ordinal", 0)
        p_after = torch.cat([p_.detach().reshape(-1) for p_ in model.parameters()])
        trained = {"loss_first": float(loss_ev[1].item()) if 1 in loss_ev else None,
                   "loss_last": float(loss_ev[K_].item()) if K_ in loss_ev else None,
                   "params_finite": bool(torch.isfinite(p_after).all().item()),
                   "param_update_l2": float((p_after - p_before).norm().item()),
                   "optimizer_steps": (int(optimizer.steps_issued()) if hasattr(optimizer, "steps_issued") else None)}
        trained["finite"] = bool(trained["params_finite"] and all(
            v_ is None or np.isfinite(v_) for v_ in (trained["loss_first"], trained["loss_last"])))
        return {"tag": tag, "steps": K_, "trained": trained, "elapsed": elapsed, "ms_per_step": elapsed * 1e3 / K_, "t_issued": t_issued,
                "windows": windows, "tail_window": tail_window, "win": win, "host_longest": host_longest, "launch_split": launch_split,
                "cpu_quota": cpu_quota, "mq_stats": mq_stats, "copy_windows": copy_windows, "prof": prof,
                "tries_total": tries_total, "miss_total": miss_total, "miss_rate": miss_rate,
                "drop_steps": (drop_step0, drop_step1), "early_steps": (early0, early1), "timed_out": timed_out}

    # (1) the driver's window: exactly K steps after W warm-up steps — `ms_per_step`
    reg_win = timed_region(K, "window")
    # (2) one whole epoch of this rank's (equalised) seeds, measured, for `value` — unless the window already was one.
    #     0.16 s at N = 1: there is no reason to extrapolate an epoch from 20 steps.
    reg_epoch = reg_win
    if K < steps_per_epoch and not args.no_epoch_leg:
        reg_epoch = timed_region(steps_per_epoch, "epoch")
    big = reg_epoch if reg_epoch["steps"] >= reg_win["steps"] else reg_win     # the region with more launches behind its statistics
    elapsed, ms_per_step = reg_win["elapsed"], reg_win["ms_per_step"]
    epoch_s = reg_epoch["elapsed"] * steps_per_epoch / reg_epoch["steps"]
    windows, win, host_longest, launch_split = big["windows"], big["win"], big["host_longest"], big["launch_split"]
    cpu_quota, mq_stats, copy_windows, timed_out = big["cpu_quota"], big["mq_stats"], big["copy_windows"], False
    prof, tries_total, miss_total, miss_rate = big["prof"], big["tries_total"], big["miss_total"], big["miss_rate"]
    t_issued = big["t_issued"]
    drop_step0, drop_step1 = big["drop_steps"]
    early_agg = big["early_steps"][1] > big["early_steps"][0]
    if early_agg:
        # GraphedTrainer.early_aggregate: the fused kernel is launched from prepare() on the load stream with a step value the
        # trainer counts itself; its ring entries are indexed by that count (the launches issued inside the region)
        drop_step0, drop_step1 = big["early_steps"]
    K_big = big["steps"]

    # ---- in-loop time of the dominant HBM-bound kernel --------------------------------------------------
    avg_ms, rows_per_launch, miss_per_launch, bytes_per_launch = gather_launch_stats_from(tries_total, miss_total, prof, D)
    achieved = bytes_per_launch / avg_ms / 1e6 if prof else float("nan")
    gather_rec = {"kernel": "k_gather", "achieved": achieved, "frac": achieved / HBM_PEAK_GBPS, "avg_launch_ms": avg_ms,
                  "rows_per_launch": rows_per_launch, "algorithmic_bytes_per_launch": bytes_per_launch,
                  "timing": "HIP events attached to each dispatch on the load stream"} if prof else None
    fused_rec = None
    if fuse_gather and cacher.rows_prof is not None and drop_step1 > drop_step0:
        # pg_spmm_fwd_rows (gather fused into the layer-0 aggregation) runs inside the replayed hipGraph, where HIP
        # events cannot be attached to one kernel: it stamps the device wall clock (100 MHz) at its start and end
        # into ring entry (dropout step % ring); entries [drop_step0+1, drop_step1] are the timed steps
        ring = cacher.rows_prof[0].view(-1, L.PG_PROF_WORDS)
        # (when the optimiser's launch advances the counter it holds the value the NEXT forward uses: shift by one)
        shift = 1 if early_agg else (0 if getattr(model, "_drop_step_primed", False) else 1)
        first = max(drop_step0, drop_step1 - PROF_RING + 2)          # the ring keeps the last PROF_RING - 1 launches
        idx = torch.arange(first + shift, drop_step1 + shift, device=dev) % PROF_RING
        raw = ring[idx].cpu().numpy().astype(np.int64)
        # per launch: first wave's start, successor's first wave's start, edges, marker kernel's stamp, end of the body
        st = np.stack([raw[:, 0], raw[:, L.PG_PROF_END0::L.PG_PROF_SHARD_STRIDE].max(axis=1), raw[:, 2], raw[:, 1],
                       raw[:, 3]], axis=1)
        if os.environ.get("PG_BENCH_DUMP_STAMPS"):
            # every stamped launch of the region, for tools/join_stamps_trace.py:
            # step index, start, body end, edges, successor's start, marker kernel's stamp (0 without one)
            np.save(os.environ["PG_BENCH_DUMP_STAMPS"],
                    np.concatenate([np.arange(first + shift, drop_step1 + shift, dtype=np.int64)[:, None], st], axis=1))
        st = st[(st[:, 1] > st[:, 0]) & (st[:, 0] > 0)]
        if len(st):
            body_ms = float(np.mean(st[:, 1] - st[:, 0])) / 1e5              # 100 MHz ticks -> ms
            has_succ = bool((st[:, 3] > st[:, 1]).mean() > 0.9)
            # THE duration behind `achieved`: first wave's start -> the dependent successor's first wave's start = the time
            # the kernel occupies its stream (body + drain + end-of-kernel release + the next dispatch's launch latency),
            # which is what rocprofv3's End - Start of the same dispatch shows (profiles/r04/fused_stamps_vs_trace_*:
            # within 1 us, launch by launch). The body alone (first wave's start -> last block's end) is reported beside it.
            slot = (st[:, 3] - st[:, 0]).astype(np.float64) / 1e5 if has_succ else None
            f_ms = float(np.mean(slot[st[:, 3] > st[:, 1]])) if has_succ else body_ms
            edges = float(np.mean(st[:, 2]))
            n_dst = float(np.mean([sl.sizes[1].item() for sl in sampler.slots]))     # |layer 1| of the last samples
            Fw = args.feat_size
            # per edge one source row read (4 F) + its position and slot (4 + 4); per destination one row
            # written (4 F) + its indptr entry (4). Rows that miss are read from the staged block instead of the cache
            # (same bytes). Padding destinations of the fixed-shape block write zeros: not counted.
            f_bytes = edges * (4 * Fw + 8) + n_dst * (4 * Fw + 4)
            dur = np.sort(slot[st[:, 3] > st[:, 1]] if has_succ else (st[:, 1] - st[:, 0]).astype(np.float64) / 1e5)
            fused_rec = {"kernel": "k_spmm_fwd_rows", "achieved": f_bytes / f_ms / 1e6, "frac": f_bytes / f_ms / 1e6 / HBM_PEAK_GBPS,
                         "avg_launch_ms": f_ms, "launches_timed": int(len(st)),
                         "launch_ms_min_median_p90_max": [float(dur[0]), float(dur[len(dur) // 2]), float(dur[int(len(dur) * 0.9)]),
                                                          float(dur[-1])], "edges_per_launch": edges,
                         "destinations_per_launch": n_dst, "algorithmic_bytes_per_launch": f_bytes,
                         # second, named figure: the kernel's body only — NOT what `frac` is computed from
                         "kernel_body_ms": body_ms, "frac_of_peak_body_only": f_bytes / body_ms / 1e6 / HBM_PEAK_GBPS,
                         "timing": ("device wall-clock stamps (100 MHz): the kernel's first wave -> the first wave of its "
                                    "dependent successor in the replayed hipGraph (what the stream pays for the launch; "
                                    "agrees with rocprofv3's End - Start of the dispatch, profiles/r04). kernel_body_ms: first "
                                    "wave's start -> last block's end (every block stamps)") if has_succ else
                                   "device wall-clock stamps written by the kernel itself (first wave's start, last block's "
                                   "end): no dependent dense / head launch followed it"}
    traffic, traffic_src = pmc_traffic(fused_rec["kernel"] if fused_rec else "k_gather")
    default_workload = (V, E, Fdim, B, k, args.model, args.cache_ratio) == (10_000_000, 100_000_000, 600, 6000, 2, "gcn", 0.30)
    if not default_workload or world > 1 or dist_step:
        traffic, traffic_src = None, "no PMC pass committed for this workload"

    dom = fused_rec or gather_rec or {"kernel": "k_gather", "achieved": float("nan"), "frac": float("nan")}
    roofline = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": dom["frac"], "traffic": traffic, "traffic_source": traffic_src}
    roofline.update({k_: v_ for k_, v_ in dom.items() if k_ not in ("kernel", "achieved", "frac")})
    if fused_rec and gather_rec:
        roofline["k_gather_in_loop"] = gather_rec       # GraphSAGE: layers 1.. are still materialised

    micro = None
    if not args.skip_microbench and rank == 0 and cacher.cached_num > 0 and not cacher.full_cached:
        micro = gather_microbench(cacher, g, dev)
        roofline["large"] = micro[1 << 20]
        roofline["step_shape_all_hits"] = micro[42000]
        roofline["sizes"] = [micro[k] for k in sorted(micro, key=lambda k: (k[0], 1) if isinstance(k, tuple) else (k, 0))]

    opt_hit = deg_hit = pre_hit = None
    if rank == 0 and not args.skip_opt_hit:
        # oracle upper bound at the same cache ratio on the same access pattern (opt_cache_hit.py:26-31),
        # over one full epoch of sampling; `layers` = what fetch_data really looks up
        from pagraph_amd import analysis
        probe = NeighborSampler(g, B, k, neighbor_type='in', shuffle=True, num_hops=num_hops, seed_nodes=subtrain,
                                prefetch=True, seed=rank)
        freq, _ = analysis.access_frequency(probe, layers=None if need is None else set(need))   # one full epoch
        opt_hit = 100.0 * analysis.optimal_cache_hit(freq, args.cache_ratio)
        deg_hit = 100.0 * analysis.degree_cache_hit(freq, g.out_degrees(), args.cache_ratio)
        # what auto_cache(policy='presample') reaches on this trace when its counts come from ANOTHER epoch
        pre_hit = 100.0 * analysis.presample_cache_hit(freq, presampled_freq(), g.out_degrees(), args.cache_ratio)
        del probe, freq

    ref_eq = None
    if world == 1 and not dist_step and use_graph and not args.skip_reference_equivalent and need is not None and not cacher.full_cached:
        # single-GPU lines only (like cpu_baseline): a second trainer would re-bind the flat gradient buffer
        ref_eq = reference_equivalent_leg(args, model, loss_fcn, optimizer, cacher, g, subtrain, labels, dev, world, rank)
        ref_eq["roofline_frac_optimised_path_for_comparison"] = roofline["frac"]
    elif need is None:
        ref_eq = "this run itself fetches every layer and field"

    cpu = None
    if rank == 0 and world == 1 and not dist_step and not args.skip_cpu_baseline:
        labels_h = labels.cpu()
        cpu_args = (args, g, sub2full.cpu().numpy(), sampler.seeds.cpu().numpy(), feat_tab,
                    norm_tab if norm_tab is not None else torch.zeros((V, 1)), labels_h, steps_per_epoch)
        # leg 1: 16 threads = the reference's sampler num_workers (pa_gcn.py:148); leg 2: every core the process may
        # use — the affinity mask, capped by the cgroup CPU quota, which is what "all cores" means
        # inside this container
        cpu = cpu_baseline(*cpu_args, args.cpu_baseline_seconds, threads=16, min_steps=args.cpu_baseline_min_steps)
        hi = host_info()
        all_cores = int(min(hi["cpus_affinity"] or hi["cpus_online"] or 16, hi["cgroup_cpu_quota"] or 1 << 30))
        cpu["cores_usable"] = all_cores
        if args.cpu_baseline_one_leg:
            cpu["all_cores"] = None
        elif all_cores != 16:
            leg = cpu_baseline(*cpu_args, args.cpu_baseline_seconds / 2, threads=all_cores)
            cpu["all_cores"] = {k_: leg[k_] for k_ in ("value", "unit", "cores", "sample", "ms_per_step")}
        else:
            cpu["all_cores"] = "same as the 16-thread leg: the process may use exactly 16 CPUs (cgroup quota)"

    seeds_total = parallel.sum_over_ranks(K * B, device=dev)
    # ---- what every rank saw (VERDICT r02 #1d): the line used to carry rank 0's host / miss-queue blocks only ----
    mine = {"rank": rank, "gpu": gpu, "device": device_identity(gpu), "trained": reg_epoch["trained"],
            "partition_vertices": Vs, "train_vertices": int(subtrain.numel()),
            "cached_rows": int(cacher.cached_num), "ms_per_step_window_local": reg_win["ms_per_step"],
            "epoch_s_local": reg_epoch["elapsed"] * steps_per_epoch / reg_epoch["steps"],
            "cache_hit_pct_rows_fetched": 100.0 * (1.0 - miss_rate), "cpu_share": cacher.cpu_share, "cpu_share_adapt": share_rec,
            "miss_mode": cacher.miss_mode, "miss_wait": "host" if cacher.host_wait else "device",
            "allreduce_in_graph": getattr(trainer, "allreduce_in_graph", None) if (world > 1 or dist_step) else None,
            "miss_mode_probe": mode_probe, "host": dict(host_info(cacher), timed_region_cgroup=cpu_quota),
            "miss_queue": mq_stats, "host_longest_iterations_ms": host_longest,
            "slowest_window_ms_per_step": max(windows) if windows else None}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    out = None
    if rank == 0:
        out = {
            # value: ONE WHOLE EPOCH, measured (config.epoch_steps_timed steps between device-wide syncs, max over ranks);
            # ms_per_step: the --steps window the driver asked for, timed the same way right before it
            "metric": "epoch_time_s", "value": epoch_s, "unit": "s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": ms_per_step, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"RMAT {V} vertices / {E} undirected edges (nnz {2*E}), feat={Fdim}, "
                                   f"2-layer {'GCN' if args.model == 'gcn' else 'GraphSAGE-mean'} hidden {hidden}, "
                                   f"batch {B}, fan-out {k}, {int(args.cache_ratio*100)}% "
                                   f"{'hot-degree' if args.cache_policy == 'degree' else 'presampled-frequency (opt-in policy)'} cache, "
                                   f"{('dg(hops=%d)' % args.dg_hops) if (world > 1 or emul_P > 1) else '1naive'} partition "
                                   f"x{emul_P if emul_P > 1 else world}, closure hops {num_hops}"
                                   + (f" — ONE rank's share (partition {share_rec_p['rank_run']}: {share_rec_p['rank_chosen_by']}) on one "
                                      f"GPU through the N > 1 step shape over a one-rank RCCL group; NOT a scaling number"
                                      if share_rec_p else (" — N > 1 step shape over a one-rank RCCL group" if dist_step else "")),
                       # --as-rank-of P: what dg did to the closures, the rank that was run, and the projection
                       "rank_of_P": (dict(share_rec_p, ms_per_step=reg_epoch["ms_per_step"],
                                          cache_hit_pct_rows_fetched=100.0 * (1.0 - miss_rate), cached_rows=int(cacher.cached_num),
                                          partition_vertices_run=Vs, host_threads=cacher.host_threads,
                                          projected_epoch_s=share_rec_p["equalised_steps_per_epoch"] * reg_epoch["ms_per_step"] / 1e3,
                                          projected_epoch_s_is="a PROJECTION: equalised steps x the measured ms/step of the rank with "
                                          "the largest closure, alone on its GPU and its host share; the all-reduce ran over a "
                                          "one-rank RCCL group (no xGMI traffic, no waiting for slower ranks)")
                                     if share_rec_p else None),
                       "step_shape": ("N > 1 (flat gradient buffer, reduce-only sums + all-reduce + Adam) over a one-rank RCCL group, "
                                      f"trainer world_size {step_world}" if dist_step else ("N > 1" if world > 1 else "one GPU")),
                       "steps_per_epoch": steps_per_epoch, "epoch_steps_timed": reg_epoch["steps"],
                       "epoch_ms_per_step": reg_epoch["ms_per_step"], "window_ms_per_step": reg_win["ms_per_step"],
                       "window_ms_per_step_windows": reg_win["windows"],
                       "statistics_from": big["tag"],     # which region the windows / roofline / miss-queue blocks describe
                       "cpu_share": cacher.cpu_share, "cpu_share_adapt": share_rec,
                       "setup_steps": S, "pipeline": "cold (drained before the timed region)" if args.cold_start else "primed",
                       "miss_mode": args.miss_mode, "miss_wait": "host" if cacher.host_wait else "device",
                       "miss_mode_probe": mode_probe, "overlap": not args.no_overlap,
                       "partition_vertices": Vs, "dg_hops": args.dg_hops if (world > 1 or emul_P > 1) else None,
                       "dg_hops2_work": {"sum_deg_squared": sum_deg_sq,
                                         "note": "adjacency entries of all two-hop multisets (65 % of them belong to train vertices, which dg --num-hops 2 "
                                                 "walks); pg_dg_partition_gpu expands ~1.2e10 per second"},
                       "hip_graph_step": use_graph,
                       # how a captured step is replayed: its kernels as plain launches (csrc/pg_tape.hip, the default on one GPU)
                       # or hipGraphLaunch (PG_FLAT_REPLAY=0, the N > 1 step with its all-reduce captured inside, or a graph that
                       # is not a chain of kernel / memset nodes)
                       "step_replay": ("plain launches (pg_tape)" if use_graph and any(getattr(s_, "tape", None) is not None
                                                                                         for s_ in trainer.slots.values())
                                       else ("hipGraphLaunch" if use_graph else "eager")),
                       # block 0's aggregation launched ahead of its step on the load stream (GraphedTrainer.early_aggregate:
                       # 'auto' = when the whole table is cached)
                       "early_layer0_aggregation": bool(early_agg),
                       "allreduce_in_graph": getattr(trainer, "allreduce_in_graph", None) if (world > 1 or dist_step) else None,
                       "fetch": "all layers+fields (reference)" if need is None else "only what the model reads"},
            # headline = the reference's counting (every row of every layer, storage.py:203-204,219-227): from the
            # reference-equivalent leg when the timed loop itself fetches only what the model reads
            "cache_hit_pct": (ref_eq["cache_hit_pct"] if isinstance(ref_eq, dict) else 100.0 * (1.0 - miss_rate)),
            "cache_hit_pct_rows_fetched_by_timed_loop": 100.0 * (1.0 - miss_rate),
            "miss_rows_per_step_reference_counting": miss_total / max(1, K_big),
            "miss_list_index_dedup": bool(cacher.dedup_misses and cacher.miss_mode == "async"),
            "reference_equivalent": ref_eq,
            "cache_hit_oracle_upper_bound_pct": opt_hit, "cache_hit_degree_policy_on_trace_pct": deg_hit,
            "cache_hit_presample_policy_on_trace_pct": pre_hit, "cache_policy": args.cache_policy,
            "presample_epochs": args.presample_epochs,
            "feat_gather_GBps": (micro[1 << 20]["GBps"] if micro else achieved),
            "seeds_per_s": seeds_total / elapsed,
            "host_issue_ms_per_step": t_issued / K_big * 1e3,     # launch thread's share; == ms_per_step when it is the bottleneck
            "ms_per_step_windows": windows, "window_steps": win, "tail_window": big["tail_window"],
            "ms_per_step_window_quantiles": ({q_: float(np.percentile(windows, p_)) for q_, p_ in (("min", 0), ("p10", 10), ("p50", 50), ("p90", 90), ("max", 100))}
                                             if windows else None), "host_longest_iterations_ms": host_longest, "launch_thread_longest_split_ms": launch_split,
            "warmup_requested": args.warmup, "misses_timed_out": timed_out,
            "host": dict(host_info(cacher), timed_region_cgroup=cpu_quota), "miss_queue": mq_stats, "miss_copy_GBps_windows": copy_windows,
            "roofline": roofline,
            "cpu_baseline": cpu,
            # evidence that the timed steps trained: first / last loss of the region `value` comes from, finite
            # parameters, and how far the parameters moved inside it
            "trained": reg_epoch["trained"], "trained_window": reg_win["trained"],
            "dist": dist_rec,
            "ranks": per_rank,
            # (main() replaces this by the `configs` block: BASELINE configs 2 and 3 as child runs, for the headline workload
            # on one GPU only — the small-graph runs of the tests and the N > 1 lines do not spawn full-size children)
            "_wants_configs": bool(default_workload and world == 1 and not dist_step and not args.no_configs and use_graph
                                   and not args.fetch_all),
        }
    if world > 1 or dist_step:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
