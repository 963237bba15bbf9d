// A captured step replayed as PLAIN launches (round 5).
// The training step is captured once per ring slot into a hipGraph (GraphedTrainer): five kernels for the sampled GCN. Replaying
// it with hipGraphLaunch costs the stream ~12 us between two replays (profiles/r03 graph_replay_gap.txt; 92 us of kernels become
// a 104 us step with the table cached), while two dependent kernels in ONE stream follow each other in ~3.4 us. pg_tape_from_graph
// walks the captured graph once — it must be a linear chain of kernel and memset nodes, which a single-stream capture of this
// library's launches is — and keeps each node's launch parameters (the graph, which owns the parameter storage and the memory pool
// the pointers refer to, must outlive the tape); pg_tape_launch issues them in order on a stream: the same kernels with the same
// arguments in the same order, without the graph launch. Anything that is not such a launch — a kernel launched through the
// module API or with an `extra` buffer (RCCL's captured collectives), a cooperative launch, a copy, a host node, a child graph —
// makes pg_tape_from_graph return PG_ERR_UNSUPPORTED and the caller keeps hipGraphLaunch. The caller must also rule out
// captured torch RNG kernels, whose seed / offset torch refreshes in CUDAGraph.replay() only (trainer.GraphedTrainer._tape_of).
#include <cstring>
#include <vector>

#include "pg_common.h"

using namespace pg;

struct pg_tape {
  struct Op {
    int kind;                      // 0 kernel, 1 memset
    hipKernelNodeParams k;
    hipMemsetParams m;
  };
  std::vector<Op> ops;
};

extern "C" {

int pg_tape_from_graph(void* hip_graph, pg_tape_t** out, int32_t* n_kernels, int32_t* n_other) {
  if (!hip_graph || !out) return PG_ERR_INVALID;
  hipGraph_t g = reinterpret_cast<hipGraph_t>(hip_graph);
  size_t n = 0;
  PG_HIP(hipGraphGetNodes(g, nullptr, &n));
  std::vector<hipGraphNode_t> nodes(n);
  if (n) PG_HIP(hipGraphGetNodes(g, nodes.data(), &n));
  // topological order of a linear chain: follow the single dependency edges
  std::vector<hipGraphNode_t> order;
  {
    std::vector<hipGraphNode_t> roots(n);
    size_t nr = n;
    PG_HIP(hipGraphGetRootNodes(g, roots.data(), &nr));
    if (n && nr != 1) return PG_ERR_UNSUPPORTED;
    hipGraphNode_t cur = n ? roots[0] : nullptr;
    while (cur) {
      order.push_back(cur);
      size_t nd = 0;
      PG_HIP(hipGraphNodeGetDependentNodes(cur, nullptr, &nd));
      if (nd == 0) break;
      if (nd != 1) return PG_ERR_UNSUPPORTED;
      hipGraphNode_t nxt = nullptr;
      PG_HIP(hipGraphNodeGetDependentNodes(cur, &nxt, &nd));
      size_t ndep = 0;
      PG_HIP(hipGraphNodeGetDependencies(nxt, nullptr, &ndep));
      if (ndep != 1) return PG_ERR_UNSUPPORTED;
      cur = nxt;
    }
    if (order.size() != n) return PG_ERR_UNSUPPORTED;
  }
  pg_tape* t = new (std::nothrow) pg_tape;
  if (!t) return PG_ERR_NOMEM;
  int nk = 0, no = 0;
  for (hipGraphNode_t nd : order) {
    hipGraphNodeType ty;
    if (hipGraphNodeGetType(nd, &ty) != hipSuccess) { delete t; return PG_ERR_HIP; }
    pg_tape::Op op{};
    if (ty == hipGraphNodeTypeKernel) {
      op.kind = 0;
      if (hipGraphKernelNodeGetParams(nd, &op.k) != hipSuccess) { delete t; return PG_ERR_HIP; }
      if (op.k.extra || !op.k.func) { delete t; return PG_ERR_UNSUPPORTED; }
      // pg_tape_launch issues hipLaunchKernel(func, ...): `func` must be a HOST function the runtime has a device kernel
      // registered for. A node captured from hipModuleLaunchKernel (JIT-compiled kernels: jiterator, a user's Triton op)
      // carries a hipFunction_t there instead — the launch would fail half-way through the step.
      {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, op.k.func) != hipSuccess) {
          (void)hipGetLastError();
          delete t;
          return PG_ERR_UNSUPPORTED;
        }
      }
      // attributes a plain launch would drop: a cooperative launch keeps hipGraphLaunch (the query failing = none set)
      {
        hipKernelNodeAttrValue av;
        memset(&av, 0, sizeof(av));
        if (hipGraphKernelNodeGetAttribute(nd, hipKernelNodeAttributeCooperative, &av) == hipSuccess) {
          if (av.cooperative) { delete t; return PG_ERR_UNSUPPORTED; }
        } else {
          (void)hipGetLastError();
        }
      }
      ++nk;
    } else if (ty == hipGraphNodeTypeMemset) {
      op.kind = 1;
      if (hipGraphMemsetNodeGetParams(nd, &op.m) != hipSuccess) { delete t; return PG_ERR_HIP; }
      if (op.m.height > 1) { delete t; return PG_ERR_UNSUPPORTED; }
      ++no;
    } else if (ty == hipGraphNodeTypeEmpty) {
      continue;
    } else {
      delete t;
      return PG_ERR_UNSUPPORTED;     // copies, host nodes, child graphs, events: the caller keeps replaying the graph
    }
    t->ops.push_back(op);
  }
  if (n_kernels) *n_kernels = nk;
  if (n_other) *n_other = no;
  *out = t;
  return PG_OK;
}

int pg_tape_launch(const pg_tape_t* t, pg_stream_t stream) {
  if (!t) return PG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  for (const auto& op : t->ops) {
    if (op.kind == 0) {
      PG_HIP(hipLaunchKernel(op.k.func, op.k.gridDim, op.k.blockDim, op.k.kernelParams, op.k.sharedMemBytes, st));
    } else {
      const size_t bytes = (size_t)op.m.width * op.m.elementSize;
      if (op.m.elementSize == 1) PG_HIP(hipMemsetAsync(op.m.dst, (int)op.m.value, bytes, st));
      else if (op.m.elementSize == 4) PG_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(op.m.dst), (int)op.m.value, op.m.width, st));
      else if (op.m.elementSize == 2) PG_HIP(hipMemsetD16Async(reinterpret_cast<hipDeviceptr_t>(op.m.dst), (unsigned short)op.m.value, op.m.width, st));
      else return PG_ERR_UNSUPPORTED;
    }
  }
  return PG_OK;
}

int pg_tape_destroy(pg_tape_t* t) {
  delete t;
  return PG_OK;
}

}  // extern "C"
