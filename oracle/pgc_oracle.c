/* pgc_oracle.c — CPU restatement of the PaGraph minibatch hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker /
 * the timed CPU baseline.  Nothing under pagraph_amd/ links, imports or calls
 * it; the product path fails loudly when libpagraph_hip.so is missing.
 *
 * Plain C (gcc -O2 -fopenmp), written independently of pagraph_amd/csrc.
 * Each function cites the reference lines it restates (paths relative to the
 * reference checkout).
 *
 * Pinning status
 *   gather  (pgc_fetch_rows)      pinned by tests/golden/g1_*, g3_* — outputs of the
 *                                 reference's own GraphCacheServer.fetch_data.
 *   sampler (pgc_sample_nodeflow) PARITY UNPINNED: the arithmetic lives in DGL 0.4.1
 *                                 (README.md:14), absent from the checkout and not
 *                                 installable; the reference does not seed it
 *                                 (examples/profile/pa_gcn.py:18-24) and has no tests.
 *                                 This restates the build-defined spec.
 *                                 The RNG (Philox4x32-10) is pinned by Random123's
 *                                 published known-answer vectors.
 *   spmm    (pgc_spmm_*)          PARITY UNPINNED against DGL; semantics = DGL's
 *                                 copy_src + mean/sum as used at
 *                                 PaGraph/model/gcn_nssc.py:71-74,139-142.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- Philox -- */
/* Philox4x32-10, Salmon/Moraes/Dror/Shaw, "Parallel random numbers: as easy as
 * 1, 2, 3" (SC'11); constants from the paper / Random123 v1.14 philox.h.      */
void pgc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

/* ---------------------------------------------------------------- gather -- */
/* PaGraph/storage/storage.py:176-204 for ONE field and ONE id list:
 *   gpu_mask = gpu_flag[tnid]                               (:179)
 *   frame[gpu_mask] = cache[localid2cacheid[tnid[gpu_mask]]]  (:191-193)
 *   frame[~gpu_mask] = table[nid_map[tnid[~gpu_mask]]]        (:117,128,199-200)
 * Returns the number of misses (what log_miss_rate receives, :203-204).        */
int64_t pgc_fetch_rows(const int64_t* tnid, int64_t n, const uint8_t* gpu_flag, const int64_t* localid2cacheid,
                       const int64_t* nid_map, const float* cache, const float* table, int32_t dim, float* out) {
  int64_t miss = 0;
#pragma omp parallel for reduction(+ : miss) schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    const int64_t id = tnid[r];
    const float* src;
    if (gpu_flag[id]) {
      src = cache + localid2cacheid[id] * (int64_t)dim;
    } else {
      src = table + nid_map[id] * (int64_t)dim;
      ++miss;
    }
    memcpy(out + r * (int64_t)dim, src, (size_t)dim * sizeof(float));
  }
  return miss;
}

/* --------------------------------------------------------------- sampler -- */
/* One destination vertex: DESIGN.md "Sampler spec" rule (3).
 * deg <= k: all in-neighbours in adjacency order. Otherwise Floyd's uniform
 * k-subset of positions {0..deg-1}: for j=0..k-1, m=deg-k+j, t=U{0..m}; take t
 * unless already taken, else m.  Draw j = 64-bit word pair (j&1) of Philox call
 * (j>>1) with counter (v, epoch, batch, layer<<24 | j>>1), key = seed.          */
static int sample_vertex(const int64_t* indptr, const int32_t* indices, int64_t v, int k, uint64_t seed,
                         uint32_t epoch, uint32_t batch, uint32_t layer, int32_t* out) {
  const int64_t beg = indptr[v];
  const int64_t deg = indptr[v + 1] - beg;
  if (deg <= k) {
    for (int64_t j = 0; j < deg; ++j) out[j] = indices[beg + j];
    return (int)deg;
  }
  uint64_t sel_small[64];
  uint64_t* sel = k <= 64 ? sel_small : (uint64_t*)malloc(sizeof(uint64_t) * (size_t)k);   /* any fan-out (pa_gcn.py:146-147) */
  if (!sel) return -1;
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t w[4] = {0, 0, 0, 0};
  for (int j = 0; j < k; ++j) {
    if ((j & 1) == 0) {
      const uint32_t ctr[4] = {(uint32_t)v, epoch, batch, (layer << 24) | (uint32_t)(j >> 1)};
      pgc_philox4x32_10(ctr, key, w);
    }
    const uint64_t r64 = (j & 1) ? (((uint64_t)w[3] << 32) | w[2]) : (((uint64_t)w[1] << 32) | w[0]);
    const uint64_t m = (uint64_t)(deg - k + j);
    uint64_t t = mulhi64(r64, m + 1);
    for (int i = 0; i < j; ++i)
      if (sel[i] == t) { t = m; break; }
    sel[j] = t;
    out[j] = indices[beg + (int64_t)t];
  }
  if (sel != sel_small) free(sel);
  return k;
}

static int cmp_i64(const void* a, const void* b) {
  const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

static int64_t lower_bound_i64(const int64_t* a, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* Whole NodeFlow for one batch (spec rules 2-7).  Outputs (caller-allocated,
 * worst case): node_mapping[sum caps] layer 0 first; layer_offsets[hops+2];
 * per block b: blk_indptr + indptr_off[b] (|layer b+1|+1 entries),
 * blk_src + src_off[b].  edges_out[b] = edges of block b.  Returns 0, or -1 on
 * allocation failure.                                                          */
int pgc_sample_nodeflow(const int64_t* indptr, const int32_t* indices, const int64_t* seeds, int32_t n_seeds,
                        int32_t k, int32_t hops, uint64_t seed, uint32_t epoch, uint32_t batch,
                        int64_t* node_mapping, int32_t* layer_offsets, int32_t* blk_indptr,
                        const int64_t* indptr_off, int32_t* blk_src, const int64_t* src_off, int32_t* edges_out) {
  if (k < 1 || hops < 1 || hops > 7) return -1;
  int64_t* layer[8] = {0};
  int64_t lsize[8] = {0};
  layer[hops] = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_seeds > 0 ? n_seeds : 1));
  if (!layer[hops]) return -1;
  memcpy(layer[hops], seeds, sizeof(int64_t) * (size_t)n_seeds);
  lsize[hops] = n_seeds;
  int rc = 0;
  for (int b = hops - 1; b >= 0 && rc == 0; --b) {
    const int64_t nd = lsize[b + 1];
    int32_t* picks = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nd * k + 1));
    int32_t* ip = blk_indptr + indptr_off[b];
    if (!picks) { rc = -1; break; }
    int64_t ne = 0;
    ip[0] = 0;
    for (int64_t p = 0; p < nd; ++p) {
      ne += sample_vertex(indptr, indices, layer[b + 1][p], k, seed, epoch, batch, (uint32_t)b, picks + ne);
      ip[p + 1] = (int32_t)ne;
    }
    /* rule (4),(5): per-layer dedup, ascending ids */
    int64_t* u = (int64_t*)malloc(sizeof(int64_t) * (size_t)(ne + 1));
    if (!u) { free(picks); rc = -1; break; }
    for (int64_t e = 0; e < ne; ++e) u[e] = picks[e];
    qsort(u, (size_t)ne, sizeof(int64_t), cmp_i64);
    int64_t nu = 0;
    for (int64_t e = 0; e < ne; ++e)
      if (e == 0 || u[e] != u[e - 1]) u[nu++] = u[e];
    layer[b] = u;
    lsize[b] = nu;
    /* rule (6): edge sources as positions inside layer b */
    int32_t* sp = blk_src + src_off[b];
    for (int64_t e = 0; e < ne; ++e) sp[e] = (int32_t)lower_bound_i64(u, nu, picks[e]);
    edges_out[b] = (int32_t)ne;
    free(picks);
  }
  if (rc == 0) {
    /* rule (7) */
    int64_t off = 0;
    for (int l = 0; l <= hops; ++l) {
      layer_offsets[l] = (int32_t)off;
      memcpy(node_mapping + off, layer[l], sizeof(int64_t) * (size_t)lsize[l]);
      off += lsize[l];
    }
    layer_offsets[hops + 1] = (int32_t)off;
  }
  for (int l = 0; l <= hops; ++l) free(layer[l]);
  return rc;
}

/* CPU baseline leg: sample `n_batches` consecutive batches of `batch_size`
 * seeds (OpenMP over batches, like DGL's num_workers batches in flight,
 * examples/profile/pa_gcn.py:71-76) and return the total number of NodeFlow
 * rows produced; node ids of batch i are written to rows_out + i*cap if non-NULL. */
int64_t pgc_sample_epoch(const int64_t* indptr, const int32_t* indices, const int64_t* seeds, int64_t n_seeds_total,
                         int32_t batch_size, int32_t k, int32_t hops, uint64_t seed, uint32_t epoch,
                         int64_t first_batch, int64_t n_batches, int64_t* rows_out, int64_t cap) {
  int64_t total = 0;
  int64_t capn = 0, c = batch_size;
  for (int l = hops; l >= 0; --l) { capn += c; c *= k; }
#pragma omp parallel for reduction(+ : total) schedule(dynamic, 1)
  for (int64_t bi = 0; bi < n_batches; ++bi) {
    const int64_t b = first_batch + bi;
    const int64_t s0 = b * batch_size;
    if (s0 >= n_seeds_total) continue;
    const int32_t ns = (int32_t)((n_seeds_total - s0) < batch_size ? (n_seeds_total - s0) : batch_size);
    int64_t* nm = (int64_t*)malloc(sizeof(int64_t) * (size_t)capn);
    int32_t lo[9];
    int64_t ioff[8], soff[8];
    int64_t ic = 0, sc = 0, d = batch_size;
    for (int bb = hops - 1; bb >= 0; --bb) { ioff[bb] = ic; soff[bb] = sc; ic += d + 1; sc += d * k; d *= k; }
    int32_t* bip = (int32_t*)malloc(sizeof(int32_t) * (size_t)ic);
    int32_t* bsr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(sc + 1));
    int32_t eo[8];
    if (nm && bip && bsr &&
        pgc_sample_nodeflow(indptr, indices, seeds + s0, ns, k, hops, seed, epoch, (uint32_t)b, nm, lo, bip, ioff, bsr,
                            soff, eo) == 0) {
      total += lo[hops + 1];
      if (rows_out) {
        const int64_t m = lo[hops + 1] < cap ? lo[hops + 1] : cap;
        memcpy(rows_out + bi * cap, nm, sizeof(int64_t) * (size_t)m);
      }
    }
    free(nm); free(bip); free(bsr);
  }
  return total;
}

/* ------------------------------------------------------------------ spmm -- */
/* DGL copy_src + mean|sum over one NodeFlow block (gcn_nssc.py:71-74,139-142):
 * out[v,:] = (1/deg) * sum_{e} h[src[e],:] accumulated in edge order; a
 * destination without in-edges gets zeros.                                     */
void pgc_spmm_fwd(const int32_t* indptr, const int32_t* src, const float* h, int64_t n_dst, int32_t dim, int mean,
                  float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < n_dst; ++v) {
    float* o = out + v * (int64_t)dim;
    for (int c = 0; c < dim; ++c) o[c] = 0.f;
    const int32_t b = indptr[v], e = indptr[v + 1];
    for (int32_t i = b; i < e; ++i) {
      const float* hr = h + (int64_t)src[i] * dim;
      for (int c = 0; c < dim; ++c) o[c] += hr[c];
    }
    if (mean && e > b) {
      const float d = (float)(e - b);
      for (int c = 0; c < dim; ++c) o[c] /= d;
    }
  }
}

/* DGL copy_src + max (graphsage_nssc.py:106-110, the 'pool' aggregator): out[v,:] = element-wise maximum of the
 * in-edge messages; zeros for a destination without in-edges (this build's reading of DGL 0.4.1 — unpinned).  */
void pgc_spmm_fwd_max(const int32_t* indptr, const int32_t* src, const float* h, int64_t n_dst, int32_t dim,
                      float* out) {
  for (int64_t v = 0; v < n_dst; ++v) {
    float* o = out + v * (int64_t)dim;
    const int32_t b = indptr[v], e = indptr[v + 1];
    for (int c = 0; c < dim; ++c) o[c] = 0.f;
    for (int32_t i = b; i < e; ++i) {
      const float* hr = h + (int64_t)src[i] * dim;
      for (int c = 0; c < dim; ++c)
        if (i == b || hr[c] > o[c]) o[c] = hr[c];
    }
  }
}

/* its backward: DGL 0.4.1's ReduceMax functor differentiates as (val == accum) [recollection, unpinned]: every in-edge
 * whose message equals the maximum receives the destination's whole gradient. x = the messages the forward saw.   */
void pgc_spmm_bwd_max(const int32_t* indptr, const int32_t* src, const float* go, const float* x, const float* out,
                      int64_t n_dst, int64_t n_src, int32_t dim, float* gh) {
  memset(gh, 0, sizeof(float) * (size_t)(n_src * dim));
  for (int64_t v = 0; v < n_dst; ++v) {
    const float* gr = go + v * (int64_t)dim;
    const float* o = out + v * (int64_t)dim;
    for (int32_t i = indptr[v]; i < indptr[v + 1]; ++i) {
      float* g = gh + (int64_t)src[i] * dim;
      const float* xr = x + (int64_t)src[i] * dim;
      for (int c = 0; c < dim; ++c)
        if (xr[c] == o[c]) g[c] += gr[c];
    }
  }
}

/* adjoint of the above: grad_h (zeroed here) [n_src, dim] */
void pgc_spmm_bwd(const int32_t* indptr, const int32_t* src, const float* go, int64_t n_dst, int64_t n_src,
                  int32_t dim, int mean, float* gh) {
  memset(gh, 0, sizeof(float) * (size_t)(n_src * dim));
  for (int64_t v = 0; v < n_dst; ++v) {
    const int32_t b = indptr[v], e = indptr[v + 1];
    if (e == b) continue;
    const float d = (float)(e - b);
    for (int32_t i = b; i < e; ++i) {
      float* g = gh + (int64_t)src[i] * dim;
      const float* gr = go + v * (int64_t)dim;
      for (int c = 0; c < dim; ++c) g[c] += mean ? gr[c] / d : gr[c];
    }
  }
}

/* ------------------------------------------------------- synthetic inputs -- */
/* RMAT candidate edge i: `scale` quadrant choices, one Philox word each
 * (call lvl>>2, word lvl&3, counter (i_lo, i_hi, lvl>>2, 'RMAT')), compared
 * against Q32 thresholds a, a+b, a+b+c.  Stands in for the PaRMAT binary of
 * README.md:36-41.                                                            */
void pgc_rmat_edges(uint64_t seed, int32_t scale, uint32_t a, uint32_t b, uint32_t c, int64_t first, int64_t n,
                    int64_t* src, int64_t* dst) {
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  const uint64_t ta = a, tab = (uint64_t)a + b, tabc = (uint64_t)a + b + c;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t e = (uint64_t)(first + i);
    uint64_t s = 0, d = 0;
    uint32_t w[4] = {0, 0, 0, 0};
    for (int lvl = 0; lvl < scale; ++lvl) {
      if ((lvl & 3) == 0) {
        const uint32_t ctr[4] = {(uint32_t)e, (uint32_t)(e >> 32), (uint32_t)(lvl >> 2), 0x524D4154u};
        pgc_philox4x32_10(ctr, key, w);
      }
      const uint64_t x = w[lvl & 3];
      int sb, db;
      if (x < ta) { sb = 0; db = 0; }
      else if (x < tab) { sb = 0; db = 1; }
      else if (x < tabc) { sb = 1; db = 0; }
      else { sb = 1; db = 1; }
      s = (s << 1) | (uint64_t)sb;
      d = (d << 1) | (uint64_t)db;
    }
    src[i] = (int64_t)s;
    dst[i] = (int64_t)d;
  }
}

/* U[0,1) fp32 features (PaGraph/data/preprocess.py:50-63): element (row, c) =
 * (word (c&3) of Philox(counter (row_lo,row_hi,c>>2,'FEAT'), key seed) >> 8) * 2^-24 */
void pgc_random_features(uint64_t seed, int64_t row0, int64_t rows, int32_t dim, float* out, int64_t stride) {
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r) {
    const uint64_t row = (uint64_t)(row0 + r);
    uint32_t w[4];
    for (int c = 0; c < dim; ++c) {
      if ((c & 3) == 0) {
        const uint32_t ctr[4] = {(uint32_t)row, (uint32_t)(row >> 32), (uint32_t)(c >> 2), 0x46454154u};
        pgc_philox4x32_10(ctr, key, w);
      }
      out[r * stride + c] = (float)(w[c & 3] >> 8) * (1.0f / 16777216.0f);
    }
  }
}
