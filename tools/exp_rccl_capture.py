"""What an RCCL process group (torch.distributed 'nccl' on ROCm) tolerates around hipGraph capture. One rank (the boxes have
one GPU): says what the capture machinery and ProcessGroupNCCL's watchdog thread do, not what N > 1 ranks do.
usage: exp_rccl_capture.py <variant>
  nosleep   eager all-reduce on stream S, synchronise, capture a graph WITHOUT a collective on S at once
  sleep     the same with a 1 s pause before the capture (the watchdog has retired the eager work by then)
  other     eager all-reduce on S, capture (no collective) on another stream at once
  inside    pause, then capture mul + all_reduce + add on S, replay twice (GraphedTrainer's in-graph all-reduce)
  flow      GraphedTrainer's pattern since the fix: EVERY eager collective on a communication stream of its own (ordered
            with the compute stream by wait_stream), captures on the compute stream right behind them, with and without a
            collective inside, three rounds"""
import os, sys, time
import torch, torch.distributed as dist
v = sys.argv[1]
if v == "flow":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    torch.cuda.set_device(0)
    t = torch.full((23000,), 3.0, device="cuda")
    comp, comm = torch.cuda.Stream(), torch.cuda.Stream()
    def eager_allreduce():
        comm.wait_stream(comp)
        with torch.cuda.stream(comm):
            dist.all_reduce(t)
        comp.wait_stream(comm)
    for rnd in range(3):
        with torch.cuda.stream(comp):
            t.fill_(3.0)
        eager_allreduce()
        for inside in (False, True):
            eager_allreduce()                      # microseconds before the capture, as in the trainer's set-up steps
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(comp):
                g.capture_begin(capture_error_mode="thread_local")
                try:
                    t.mul_(2.0)
                    if inside:
                        dist.all_reduce(t)
                    t.add_(1.0)
                    time.sleep(0.25)
                finally:
                    g.capture_end()
                t.fill_(3.0); g.replay(); g.replay()
            eager_allreduce()
            comp.synchronize()
            print("flow round", rnd, "collective inside" if inside else "no collective inside", float(t[0]), "(expect 15.0)")
    print("flow ok")
    dist.destroy_process_group()
    sys.exit(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
t = torch.full((23000,), 3.0, device=dev)
st, st2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(st):
    dist.all_reduce(t.clone())
st.synchronize()
if v in ("sleep", "inside"):
    time.sleep(1.0)
cap = st2 if v == "other" else st
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap):
    g.capture_begin(capture_error_mode="thread_local")
    try:
        t.mul_(2.0)
        if v == "inside":
            dist.all_reduce(t)
        t.add_(1.0)
        time.sleep(0.3)          # stay in capture mode for a few watchdog periods
    finally:
        g.capture_end()
    t.fill_(3.0); g.replay(); cap.synchronize()
    print(v, "replay 1:", float(t[0]), "(expect 7.0)")
    g.replay(); cap.synchronize()
    print(v, "replay 2:", float(t[0]), "(expect 15.0)")
time.sleep(0.5)
with torch.cuda.stream(st):
    dist.all_reduce(t)           # an eager collective after the capture still works
st.synchronize()
print(v, "ok")
dist.destroy_process_group()
