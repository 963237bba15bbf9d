// The load-stream half of one minibatch as ONE call (round 5) — examples/profile/pa_gcn.py:86-91's 'gpu-load' range for a
// table that is resident in HBM: cacher.fetch_data(nf) (storage.py:207-216 fetch_from_cache, reduced to the slot look-up
// because the rows are read in place by the aggregation), the source-major copies of the sampled blocks, the aggregations
// of raw feature rows that run ahead of their step, and labels[nf.layer_parent_nid(-1)] (pa_gcn.py:89-90).
// Nothing here is new device work: it is the sequence GraphedTrainer.prepare() issued as seven ctypes / torch calls —
// wait for the sample, pg_slots_full, pg_sampler_transpose, pg_spmm_fwd_rows (one per early block),
// pg_gather_labels_sc, event record — with every argument that does not change from batch to batch resolved once per ring
// slot (pg_batch_plan_t). Alone it changed nothing (round 5: with the table cached the compute stream is the bound); together with the
// captured step replayed as plain launches (pg_tape.hip: five hipLaunchKernel calls instead of one hipGraphLaunch on the launch
// thread) it is what keeps the launch thread off the critical path: each Python -> C transition costs 3-5 us on top of the HIP
// call behind it.
#include "pg_common.h"

using namespace pg;

extern "C" {

int pg_batch_prepare(const pg_batch_plan_t* p, uint64_t drop_step_value) {
  if (!p || !p->load_stream || p->n_early < 0 || p->n_early > PG_MAX_LAYERS) return PG_ERR_INVALID;
  hipStream_t ls = as_stream(p->load_stream);
  if (p->ev_sampled) PG_HIP(hipStreamWaitEvent(ls, reinterpret_cast<hipEvent_t>(p->ev_sampled), 0));
  int rc = PG_OK;
  if (p->rows > 0) {
    rc = pg_slots_full(p->ids, p->rows, p->slot_map, p->slots_out, p->stats, p->load_stream);
    if (rc != PG_OK) return rc;
  }
  if (p->sampler && p->transpose) {
    rc = pg_sampler_transpose(p->sampler, &p->desc, p->load_stream);
    if (rc != PG_OK) return rc;
  }
  for (int i = 0; i < p->n_early; ++i) {
    const pg_batch_early_t& e = p->early[i];
    pg_dropout_t d = e.drop;
    d.step_value = drop_step_value;
    rc = pg_spmm_fwd_rows(e.indptr, e.src, &e.rows, e.n_dst, e.dim, e.reduce, e.out, e.out_stride, e.has_drop ? &d : nullptr,
                          e.prof, e.prof_ring, p->load_stream);
    if (rc != PG_OK) return rc;
  }
  if (p->n_label_rows > 0) {
    rc = pg_gather_labels_sc(p->label_ids, p->n_label_rows, p->labels, p->labels_len, p->label_fill, p->label_out, p->n_valid,
                             p->label_scratch, p->load_stream);
    if (rc != PG_OK) return rc;
  }
  if (p->ev_ready) PG_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(p->ev_ready), ls));
  return PG_OK;
}

}  // extern "C"
