"""hunt for the flaky NaN loss of test_zerocopy_refused_...: second GraphedTrainer of a process, nothing cached, step index 4"""
import os, sys
import numpy as np, torch, scipy.sparse as spsp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as Fn
from pagraph_amd import _lib as L, ops
from pagraph_amd.model import GCNSampling
from pagraph_amd.optim import Adam
from pagraph_amd.sampling import DeviceGraph, NeighborSampler
from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
from pagraph_amd.trainer import GraphedTrainer, cycle_batches

dev = torch.device("cuda", 0)
rng = np.random.default_rng(8)
V, Fd, C, B = 4000, 600, 9, 500
feats = torch.from_numpy(rng.random((V, Fd), dtype=np.float32))
w = 1.0 / np.arange(1, V + 1) ** 0.9; w /= w.sum()
s_ = rng.choice(V, 30000, p=w); d_ = rng.choice(V, 30000, p=w)
adj = spsp.coo_matrix((np.ones(60000, np.int8), (np.concatenate([s_, d_]), np.concatenate([d_, s_]))), shape=(V, V)).tocsr()
g = DeviceGraph(adj)
labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
CHECK = int(os.environ.get("CHECK_STEP", 5))
for p in range(int(os.environ.get("PASSES", 60))):
    store2 = HostFeatureStore({"features": feats})
    cc = GraphCacheServer(store2, V, torch.arange(V), 0, miss_mode="async")
    cc.init_field(["features"])
    torch.manual_seed(4)
    model = GCNSampling(Fd, 32, C, 1, Fn.relu, 0.2).to(dev).train()
    smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=np.arange(0, V, 2), seed=6,
                          static=True, defer_transpose=True)
    tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), Adam(model.parameters(), lr=1e-2), cc, smp, labels, dev,
                        need=model.required_inputs(3), keep_losses=True)
    out = []
    state = {}
    def on_step(k, l):
        out.append(l)
        if k == CHECK:
            tr.synchronize(); torch.cuda.synchronize()
            if not bool(torch.isfinite(l)):
                state["bad"] = k
                # which slot computed step k? the one whose graph ran last: find by loss tensor identity is lost (clone); dump all
                for key, s in tr.slots.items():
                    nf = s.nf_cur
                    ids0 = nf._node_mapping.tousertensor()[nf._layer_offsets[0]:nf._layer_offsets[1]].cpu().numpy()
                    rs = s.plan.row_sources.get((0, "features")) if s.plan else None
                    msg = f"  slot {s.slot_index}: n_valid {int(s.n_valid.item())} labels!=-100 {int((s.label != -100).sum())} loss {float(s.loss) if s.loss is not None else None}"
                    if rs is not None:
                        n = rs.shape[0]
                        ip = torch.arange(n + 1, dtype=torch.int32, device=dev); sr = torch.arange(n, dtype=torch.int32, device=dev)
                        got = ops.aggregate_rows(ip, sr, rs, n, "sum").cpu().numpy()
                        valid = ids0 >= 0
                        want = feats.numpy()[np.where(valid, ids0, 0)]
                        bad_rows = np.nonzero(valid & ~(got == want).all(1))[0]
                        sl = rs.slots.cpu().numpy()
                        msg += f" | layer0 rows {int(valid.sum())} wrong rows {len(bad_rows)} (first {bad_rows[:5]}) slots of wrong {sl[bad_rows[:5]]} nonfinite {int((~np.isfinite(got)).sum())}"
                    print(msg, flush=True)
    tr.on_step = on_step
    it = cycle_batches(smp, 40)
    tr.run_steps(it, 14)
    tr.synchronize(); torch.cuda.synchronize()
    ls = torch.stack([l.detach().float().cpu() for l in out])
    if "bad" in state or not bool(torch.isfinite(ls).all()):
        print(f"pass {p}: NON-FINITE", ls.tolist(), flush=True)
        print("miss queue", cc.miss_queue_stats(), "timed_out", cc.misses_timed_out(), flush=True)
        break
    cc.shutdown_miss_queue()
else:
    print("no failure")
