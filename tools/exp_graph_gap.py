"""what a stream pays between two graph replays: N replays of a graph of K short kernels back to back, with and without an
event record / a (satisfied) event wait between them, against the same kernels launched one by one.
usage: python tools/exp_graph_gap.py [K]"""
import sys, time
import torch
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = 2000
x = torch.zeros(1 << 22, device=dev)            # 16 MB: ~8 us per add_
st = torch.cuda.Stream(device=dev)
other = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    for _ in range(3):
        x.add_(1.0)
    st.synchronize()
    graphs = []
    for _ in range(5):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(K):
                x.add_(1.0)
        graphs.append(g)
    st.synchronize()
    # kernel time alone
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    graphs[0].replay(); st.synchronize()

    def run(label, body):
        for i in range(50):
            body(i)
        st.synchronize()
        t0 = time.perf_counter()
        s.record(st)
        for i in range(N):
            body(i)
        t_issue = time.perf_counter() - t0
        e.record(st)
        st.synchronize()
        print(f"{label:58s} {s.elapsed_time(e) / N * 1e3:7.1f} us per replay on the stream, host issue {t_issue / N * 1e6:6.1f} us")

    evs = [torch.cuda.Event() for _ in range(5)]
    done = torch.cuda.Event(); done.record(other); other.synchronize()
    run(f"one graph of {K} kernels, same exec", lambda i: graphs[0].replay())
    run(f"5 execs round robin", lambda i: graphs[i % 5].replay())
    run(f"5 execs + event record after each", lambda i: (graphs[i % 5].replay(), evs[i % 5].record(st)))
    run(f"5 execs + 2 event records after each", lambda i: (graphs[i % 5].replay(), evs[i % 5].record(st), evs[(i + 1) % 5].record(st)))
    run(f"5 execs + wait on a complete event + record", lambda i: (st.wait_event(done), graphs[i % 5].replay(), evs[i % 5].record(st)))

    def with_other(i):
        # the record is waited for by another stream (as the sampler stream waits for 'slot free')
        graphs[i % 5].replay(); evs[i % 5].record(st); other.wait_event(evs[i % 5])
    run(f"5 execs + record that another stream waits for", with_other)
    run(f"{K} eager launches", lambda i: [x.add_(1.0) for _ in range(K)])
    run(f"{K} eager launches + event record", lambda i: ([x.add_(1.0) for _ in range(K)], evs[i % 5].record(st)))
