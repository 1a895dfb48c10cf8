/*
 * grx_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see grx_oracle.h).
 *
 * Plain C restatement of the reference's algorithms for the frontier
 * advance/filter/uniquify path and its three clients.  Citations are
 * relative to /root/reference/.
 */
#define _GNU_SOURCE
#include "grx_oracle.h"

#include <ctype.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

void orc_free(void* p) { free(p); }

/* ------------------------------------------------------------------------
 * Matrix Market loader.
 * Follows include/gunrock/io/matrix_market.hxx:99-240:
 *   - banner decides pattern / real / integer and general / symmetric
 *     (mmio banner parsing, include/gunrock/io/detail/mmio.cpp mm_read_banner);
 *   - pattern entries get value 1.0f (:146-163), real/integer read a double and
 *     narrow to float (:164-187), indices are 1-based in the file (:155-160);
 *   - symmetric storage: every off-diagonal (i,j) is emitted as (i,j) directly
 *     followed by (j,i); diagonal entries once (:194-235).
 * complex / hermitian / skew / array files are rejected as the reference does
 * ("Unrecognized matrix market format type", :188-191).
 * ---------------------------------------------------------------------- */
static void lower(char* s) {
  for (; *s; ++s) *s = (char)tolower((unsigned char)*s);
}

int orc_mtx_load(const char* path, int32_t* n_rows, int32_t* n_cols, int32_t* nnz_out,
                 int32_t** row_idx, int32_t** col_idx, float** values) {
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  char line[1100], banner[64], obj[64], fmt[64], field[64], sym[64];
  if (!fgets(line, sizeof line, f)) { fclose(f); return -2; }
  if (sscanf(line, "%63s %63s %63s %63s %63s", banner, obj, fmt, field, sym) != 5) {
    fclose(f); return -2;
  }
  lower(obj); lower(fmt); lower(field); lower(sym);
  if (strcmp(banner, "%%MatrixMarket") != 0 || strcmp(obj, "matrix") != 0) { fclose(f); return -2; }
  if (strcmp(fmt, "coordinate") != 0) { fclose(f); return -3; } /* array => not sparse */
  int is_pattern = !strcmp(field, "pattern");
  int is_num = !strcmp(field, "real") || !strcmp(field, "integer");
  if (!is_pattern && !is_num) { fclose(f); return -4; }
  int is_symmetric = !strcmp(sym, "symmetric");

  /* skip comments, read the size line */
  size_t M = 0, N = 0, NZ = 0;
  for (;;) {
    if (!fgets(line, sizeof line, f)) { fclose(f); return -5; }
    if (line[0] == '%') continue;
    if (sscanf(line, "%zu %zu %zu", &M, &N, &NZ) == 3) break;
  }
  if (M >= INT_MAX || N >= INT_MAX || NZ >= INT_MAX) { fclose(f); return -6; }

  int32_t* I = (int32_t*)malloc(sizeof(int32_t) * (NZ ? NZ : 1));
  int32_t* J = (int32_t*)malloc(sizeof(int32_t) * (NZ ? NZ : 1));
  float* V = (float*)malloc(sizeof(float) * (NZ ? NZ : 1));
  for (size_t i = 0; i < NZ; ++i) {
    size_t r = 0, c = 0;
    double w = 1.0;
    int got = is_pattern ? fscanf(f, " %zu %zu \n", &r, &c)
                         : fscanf(f, " %zu %zu %lf \n", &r, &c, &w);
    if (got != (is_pattern ? 2 : 3) || r == 0 || c == 0) {
      free(I); free(J); free(V); fclose(f); return -7;
    }
    I[i] = (int32_t)r - 1;
    J[i] = (int32_t)c - 1;
    V[i] = is_pattern ? 1.0f : (float)w;
  }
  fclose(f);

  if (is_symmetric) {
    size_t off = 0;
    for (size_t i = 0; i < NZ; ++i) off += (I[i] != J[i]);
    size_t nz2 = 2 * off + (NZ - off);
    int32_t* I2 = (int32_t*)malloc(sizeof(int32_t) * (nz2 ? nz2 : 1));
    int32_t* J2 = (int32_t*)malloc(sizeof(int32_t) * (nz2 ? nz2 : 1));
    float* V2 = (float*)malloc(sizeof(float) * (nz2 ? nz2 : 1));
    size_t p = 0;
    for (size_t i = 0; i < NZ; ++i) {
      I2[p] = I[i]; J2[p] = J[i]; V2[p] = V[i]; ++p;
      if (I[i] != J[i]) { I2[p] = J[i]; J2[p] = I[i]; V2[p] = V[i]; ++p; }
    }
    free(I); free(J); free(V);
    I = I2; J = J2; V = V2; NZ = nz2;
  }
  *n_rows = (int32_t)M; *n_cols = (int32_t)N; *nnz_out = (int32_t)NZ;
  *row_idx = I; *col_idx = J; *values = V;
  return 0;
}

/* ------------------------------------------------------------------------
 * COO -> CSR.  Follows include/gunrock/formats/csr.hxx:119-147: per-row
 * histogram, exclusive prefix, scatter in input order (stable, duplicates
 * kept), offsets shifted back.  (Defect q5 of SURVEY 8a': the reference's host
 * branch counts into an unsized vector; the intended behaviour is restated.)
 * ---------------------------------------------------------------------- */
void orc_coo_to_csr(int32_t n_rows, int32_t nnz, const int32_t* row_idx,
                    const int32_t* col_idx, const float* values, int32_t* Ap,
                    int32_t* Aj, float* Ax) {
  memset(Ap, 0, sizeof(int32_t) * ((size_t)n_rows + 1));
  for (int32_t n = 0; n < nnz; ++n) ++Ap[row_idx[n]];
  int32_t sum = 0;
  for (int32_t i = 0; i < n_rows; ++i) { int32_t t = Ap[i]; Ap[i] = sum; sum += t; }
  Ap[n_rows] = nnz;
  for (int32_t n = 0; n < nnz; ++n) {
    int32_t dest = Ap[row_idx[n]]++;
    Aj[dest] = col_idx[n];
    if (Ax) Ax[dest] = values ? values[n] : 1.0f;
  }
  int32_t last = 0;
  for (int32_t i = 0; i <= n_rows; ++i) { int32_t t = Ap[i]; Ap[i] = last; last = t; }
}

/* ------------------------------------------------------------------------
 * ".csr" binary cache.  Layout of include/gunrock/formats/csr.hxx:159-236:
 * {rows:int32, cols:int32, nnz:int32} then row_offsets[rows+1], column
 * indices[nnz], values[nnz], all raw little-endian.
 * ---------------------------------------------------------------------- */
int orc_csr_write_binary(const char* path, int32_t n_rows, int32_t n_cols, int32_t nnz,
                         const int32_t* Ap, const int32_t* Aj, const float* Ax) {
  FILE* f = fopen(path, "wb");
  if (!f) return -1;
  fwrite(&n_rows, sizeof n_rows, 1, f);
  fwrite(&n_cols, sizeof n_cols, 1, f);
  fwrite(&nnz, sizeof nnz, 1, f);
  fwrite(Ap, sizeof(int32_t), (size_t)n_rows + 1, f);
  fwrite(Aj, sizeof(int32_t), (size_t)nnz, f);
  fwrite(Ax, sizeof(float), (size_t)nnz, f);
  fclose(f);
  return 0;
}

int orc_csr_read_binary(const char* path, int32_t* n_rows, int32_t* n_cols, int32_t* nnz,
                        int32_t** Ap, int32_t** Aj, float** Ax) {
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  if (fread(n_rows, 4, 1, f) != 1 || fread(n_cols, 4, 1, f) != 1 || fread(nnz, 4, 1, f) != 1) {
    fclose(f); return -2;
  }
  *Ap = (int32_t*)malloc(4 * ((size_t)*n_rows + 1));
  *Aj = (int32_t*)malloc(4 * ((size_t)*nnz + 1));
  *Ax = (float*)malloc(4 * ((size_t)*nnz + 1));
  int ok = fread(*Ap, 4, (size_t)*n_rows + 1, f) == (size_t)*n_rows + 1 &&
           fread(*Aj, 4, (size_t)*nnz, f) == (size_t)*nnz &&
           fread(*Ax, 4, (size_t)*nnz, f) == (size_t)*nnz;
  fclose(f);
  return ok ? 0 : -3;
}

/* ------------------------------------------------------------------------
 * R-MAT generator.  NOT from the reference (it has none): this
 * is the build's own specification, restated here so that the GPU generator
 * (essentials_amd/csrc/rmat.hip) can be checked bit for bit.
 *   mix(x)  = splitmix64 finaliser of x + 0x9E3779B97F4A7C15
 *   base    = mix(seed ^ mix(k))
 *   level l = 0..scale-1 draws r = high 32 bits of mix(base + l) and picks the
 *             quadrant by integer thresholds of (A,B,C,D) = (.57,.19,.19,.05):
 *             r < TA:(0,0)  r < TAB:(0,1)  r < TABC:(1,0)  else (1,1)
 *   no vertex permutation, no noise.
 * Symmetrisation is the Matrix-Market loader's (matrix_market.hxx:194-235):
 * pair k emits (u,v) then (v,u), a self loop once; duplicates are kept; rows
 * are then stably counting-sorted (csr.hxx:119-147).
 * ---------------------------------------------------------------------- */
#define RMAT_TA 2448131358u   /* floor(0.57 * 2^32) */
#define RMAT_TAB 3264175144u  /* floor(0.76 * 2^32) */
#define RMAT_TABC 4080218931u /* floor(0.95 * 2^32) */

static inline uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

void orc_rmat_pair(uint32_t scale, uint64_t seed, uint64_t k, int32_t* u_out, int32_t* v_out) {
  uint64_t base = mix64(seed ^ mix64(k));
  uint32_t u = 0, v = 0;
  for (uint32_t l = 0; l < scale; ++l) {
    uint32_t r = (uint32_t)(mix64(base + l) >> 32);
    uint32_t ub = r >= RMAT_TAB;
    uint32_t vb = (r >= RMAT_TA && r < RMAT_TAB) || (r >= RMAT_TABC);
    u = (u << 1) | ub;
    v = (v << 1) | vb;
  }
  *u_out = (int32_t)u;
  *v_out = (int32_t)v;
}

float orc_rmat_weight(uint64_t weight_seed, uint64_t k) {
  if (weight_seed == 0) return 1.0f;
  return (float)(1 + (mix64(weight_seed ^ mix64(k ^ 0x5bd1e995u)) & 63u));
}

int64_t orc_rmat_count(uint32_t scale, uint32_t ef, uint64_t seed, int symmetrize) {
  uint64_t pairs = (uint64_t)ef << scale;
  int64_t cnt = 0;
  for (uint64_t k = 0; k < pairs; ++k) {
    int32_t u, v;
    orc_rmat_pair(scale, seed, k, &u, &v);
    cnt += (symmetrize && u != v) ? 2 : 1;
  }
  return cnt;
}

void orc_rmat_csr(uint32_t scale, uint32_t ef, uint64_t seed, uint64_t wseed, int symmetrize,
                  int32_t* Ap, int32_t* Aj, float* Ax) {
  uint64_t pairs = (uint64_t)ef << scale;
  int32_t n = (int32_t)1 << scale;
  memset(Ap, 0, sizeof(int32_t) * ((size_t)n + 1));
  for (uint64_t k = 0; k < pairs; ++k) {
    int32_t u, v;
    orc_rmat_pair(scale, seed, k, &u, &v);
    ++Ap[u];
    if (symmetrize && u != v) ++Ap[v];
  }
  int32_t sum = 0;
  for (int32_t i = 0; i < n; ++i) { int32_t t = Ap[i]; Ap[i] = sum; sum += t; }
  Ap[n] = sum;
  for (uint64_t k = 0; k < pairs; ++k) {
    int32_t u, v;
    orc_rmat_pair(scale, seed, k, &u, &v);
    float w = orc_rmat_weight(wseed, k);
    int32_t d = Ap[u]++;
    Aj[d] = v; Ax[d] = w;
    if (symmetrize && u != v) { d = Ap[v]++; Aj[d] = u; Ax[d] = w; }
  }
  int32_t last = 0;
  for (int32_t i = 0; i <= n; ++i) { int32_t t = Ap[i]; Ap[i] = last; last = t; }
}

/* ------------------------------------------------------------------------
 * The reference's CPU checkers.
 * orc_bfs_heap  follows examples/algorithms/bfs/bfs_cpu.hxx:29-67
 * orc_sssp_heap follows examples/algorithms/sssp/sssp_cpu.hxx:33-71
 * Both: labels initialised to numeric max outside the timed region, then a
 * min-priority queue of (vertex, label) pairs seeded with (source, 0); pop,
 * relax every out-edge with strict '<', push on improvement.  No stale-entry
 * skip (the reference has none).  Timer covers the search only (:35,:65-67).
 * A binary heap on the label stands in for std::priority_queue; ties may pop
 * in another order, which cannot change the (unique) fix point.
 * ---------------------------------------------------------------------- */
typedef struct { int32_t v; int32_t key; } ih_t;
typedef struct { int32_t v; float key; } fh_t;

#define HEAP_IMPL(NAME, T, KT)                                                  \
  typedef struct { T* a; size_t n, cap; } NAME##_heap;                           \
  static void NAME##_push(NAME##_heap* h, int32_t v, KT key) {                   \
    if (h->n == h->cap) {                                                        \
      h->cap = h->cap ? h->cap * 2 : 1024;                                       \
      h->a = (T*)realloc(h->a, h->cap * sizeof(T));                              \
    }                                                                            \
    size_t i = h->n++;                                                           \
    while (i) {                                                                  \
      size_t p = (i - 1) >> 1;                                                   \
      if (!(h->a[p].key > key)) break;                                           \
      h->a[i] = h->a[p];                                                         \
      i = p;                                                                     \
    }                                                                            \
    h->a[i].v = v; h->a[i].key = key;                                            \
  }                                                                              \
  static T NAME##_pop(NAME##_heap* h) {                                          \
    T top = h->a[0];                                                             \
    T last = h->a[--h->n];                                                       \
    size_t i = 0, n = h->n;                                                      \
    for (;;) {                                                                   \
      size_t c = 2 * i + 1;                                                      \
      if (c >= n) break;                                                         \
      if (c + 1 < n && h->a[c + 1].key < h->a[c].key) ++c;                       \
      if (!(last.key > h->a[c].key)) break;                                      \
      h->a[i] = h->a[c];                                                         \
      i = c;                                                                     \
    }                                                                            \
    if (n) h->a[i] = last;                                                       \
    return top;                                                                  \
  }

HEAP_IMPL(ih, ih_t, int32_t)
HEAP_IMPL(fh, fh_t, float)

float orc_bfs_heap(int32_t n, const int32_t* Ap, const int32_t* Aj, int32_t source,
                   int32_t* depth) {
  for (int32_t i = 0; i < n; ++i) depth[i] = INT32_MAX;
  double t0 = now_ms();
  depth[source] = 0;
  ih_heap h = {0, 0, 0};
  ih_push(&h, source, 0);
  while (h.n) {
    ih_t cur = ih_pop(&h);
    int32_t nd = cur.key + 1;
    for (int32_t e = Ap[cur.v]; e < Ap[cur.v + 1]; ++e) {
      int32_t nb = Aj[e];
      if (nd < depth[nb]) { depth[nb] = nd; ih_push(&h, nb, nd); }
    }
  }
  double t1 = now_ms();
  free(h.a);
  return (float)(t1 - t0);
}

float orc_sssp_heap(int32_t n, const int32_t* Ap, const int32_t* Aj, const float* Ax,
                    int32_t source, float* dist) {
  for (int32_t i = 0; i < n; ++i) dist[i] = 3.402823466e+38f; /* numeric_limits<float>::max() */
  double t0 = now_ms();
  dist[source] = 0;
  fh_heap h = {0, 0, 0};
  fh_push(&h, source, 0.0f);
  while (h.n) {
    fh_t cur = fh_pop(&h);
    for (int32_t e = Ap[cur.v]; e < Ap[cur.v + 1]; ++e) {
      int32_t nb = Aj[e];
      float nd = cur.key + Ax[e];
      if (nd < dist[nb]) { dist[nb] = nd; fh_push(&h, nb, nd); }
    }
  }
  double t1 = now_ms();
  free(h.a);
  return (float)(t1 - t0);
}

/* ------------------------------------------------------------------------
 * advance, block_mapped semantics.
 * Follows include/gunrock/framework/operators/advance/block_mapped.hxx:67-146
 * and advance/helpers.hxx:112-146: invalid input slots (-1) contribute no
 * work; every (valid slot, out-edge) calls op(src, dst, edge, weight) exactly
 * once; the output holds one slot per traversed edge, the neighbour where the
 * op returned true and -1 otherwise.  (The reference's slot ORDER across blocks
 * is unspecified -- global cursor, :94-101 -- so tests compare outputs as
 * multisets; this restatement writes them in input order.)
 * input == NULL restates advance_io_type_t::graph (vertex i for slot i, :67-70).
 * ---------------------------------------------------------------------- */
int64_t orc_advance(int32_t n, const int32_t* Ap, const int32_t* Aj, const float* Ax,
                    const int32_t* input, int64_t n_in, orc_edge_op op, void* ctx,
                    int32_t* output) {
  (void)n;
  int64_t out = 0;
  for (int64_t i = 0; i < n_in; ++i) {
    int32_t v = input ? input[i] : (int32_t)i;
    if (v == -1) continue;
    for (int32_t e = Ap[v]; e < Ap[v + 1]; ++e) {
      int32_t nb = Aj[e];
      int cond = op(v, nb, e, Ax ? Ax[e] : 1.0f, ctx);
      if (output) output[out] = cond ? nb : -1;
      ++out;
    }
  }
  return out;
}

/* filter::bypass -- filter/bypass.hxx:29-45: same length, invalid stays
 * invalid without calling the predicate, rejected entries become -1. */
int64_t orc_filter_bypass(const int32_t* in, int64_t n_in, orc_vertex_op op, void* ctx,
                          int32_t* out) {
  for (int64_t i = 0; i < n_in; ++i) {
    int32_t v = in[i];
    out[i] = (v == -1) ? -1 : (op(v, ctx) ? v : -1);
  }
  return n_in;
}

/* filter::predicated (predicated.hxx:24-38), filter::remove (remove.hxx:23-37)
 * and filter::compact (compact.hxx:23-36) all keep, in order, exactly the
 * valid entries for which the predicate is true. */
int64_t orc_filter_keep(const int32_t* in, int64_t n_in, orc_vertex_op op, void* ctx,
                        int32_t* out) {
  int64_t m = 0;
  for (int64_t i = 0; i < n_in; ++i) {
    int32_t v = in[i];
    if (v != -1 && op(v, ctx)) out[m++] = v;
  }
  return m;
}

/* uniquify -- uniquify/uniquify.hxx:24-34 (sort unless best-effort) then
 * uniquify/unique.hxx:26-30 (drop consecutive duplicates). */
static int cmp_i32(const void* a, const void* b) {
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return (x > y) - (x < y);
}
int64_t orc_uniquify(int32_t* d, int64_t n, int do_sort) {
  if (n == 0) return 0;
  if (do_sort) qsort(d, (size_t)n, sizeof(int32_t), cmp_i32);
  int64_t m = 1;
  for (int64_t i = 1; i < n; ++i)
    if (d[i] != d[m - 1]) d[m++] = d[i];
  return m;
}

/* ------------------------------------------------------------------------
 * The three clients driven through the operator restatements, so that
 * iteration counts and per-level frontier sizes can be pinned as well.
 * ---------------------------------------------------------------------- */
typedef struct { int32_t* depth; int32_t iteration; } bfs_ctx;
/* include/gunrock/algorithms/bfs.hxx:92-114 */
static int bfs_op(int32_t src, int32_t dst, int32_t e, float w, void* c) {
  (void)src; (void)e; (void)w;
  bfs_ctx* x = (bfs_ctx*)c;
  int32_t old = x->depth[dst];
  int32_t nv = x->iteration + 1;
  if (nv < old) x->depth[dst] = nv; /* atomic::min, intended semantics */
  return nv < old;
}

static void trace_level(orc_trace* t, int32_t it, const int32_t* f, int64_t n) {
  if (!t || it >= 64) return;
  int64_t valid = 0;
  for (int64_t i = 0; i < n; ++i) valid += (f[i] != -1);
  t->frontier_slots[it] = n;
  t->frontier_valid[it] = valid;
}

static int64_t frontier_out_len(const int32_t* Ap, const int32_t* f, int64_t n) {
  int64_t s = 0;
  for (int64_t i = 0; i < n; ++i)
    if (f[i] != -1) s += Ap[f[i] + 1] - Ap[f[i]];
  return s;
}

/* bfs.hxx:51-60 (reset), :74-78 (prepare_frontier), :80-132 (loop) under
 * framework/enactor.hxx:243-254 (enact) and :294-296 (is_converged). */
void orc_bfs_frontier(int32_t n, const int32_t* Ap, const int32_t* Aj, const float* Ax,
                      int32_t source, int32_t* depth, orc_trace* trace) {
  for (int32_t i = 0; i < n; ++i) depth[i] = INT32_MAX;
  depth[source] = 0;
  if (trace) memset(trace, 0, sizeof *trace);
  int64_t n_in = 1;
  int32_t* in = (int32_t*)malloc(sizeof(int32_t));
  in[0] = source;
  bfs_ctx ctx = {depth, 0};
  while (n_in != 0) {
    trace_level(trace, ctx.iteration, in, n_in);
    int64_t len = frontier_out_len(Ap, in, n_in);
    int32_t* out = (int32_t*)malloc(sizeof(int32_t) * (size_t)(len ? len : 1));
    orc_advance(n, Ap, Aj, Ax, in, n_in, bfs_op, &ctx, out);
    if (trace) trace->edges_traversed += len;
    free(in);
    in = out; n_in = len;
    ++ctx.iteration;
  }
  free(in);
  if (trace) trace->iterations = ctx.iteration;
}

typedef struct { float* dist; int32_t* visited; int32_t iteration; } sssp_ctx;
/* include/gunrock/algorithms/sssp.hxx:110-124 */
static int sssp_edge_op(int32_t src, int32_t dst, int32_t e, float w, void* c) {
  (void)e;
  sssp_ctx* x = (sssp_ctx*)c;
  float nd = x->dist[src] + w;
  float old = x->dist[dst];
  if (nd < old) x->dist[dst] = nd;
  return nd < old;
}
/* include/gunrock/algorithms/sssp.hxx:126-136 */
static int sssp_vertex_op(int32_t v, void* c) {
  sssp_ctx* x = (sssp_ctx*)c;
  if (x->visited[v] == x->iteration) return 0;
  x->visited[v] = x->iteration;
  return 1;
}

/* sssp.hxx:52-78 (init/reset), :92-96, :98-151 (advance, then bypass filter). */
void orc_sssp_frontier(int32_t n, const int32_t* Ap, const int32_t* Aj, const float* Ax,
                       int32_t source, float* dist, orc_trace* trace) {
  int32_t* visited = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
  for (int32_t i = 0; i < n; ++i) { dist[i] = 3.402823466e+38f; visited[i] = -1; }
  dist[source] = 0;
  if (trace) memset(trace, 0, sizeof *trace);
  int64_t n_in = 1;
  int32_t* in = (int32_t*)malloc(sizeof(int32_t));
  in[0] = source;
  sssp_ctx ctx = {dist, visited, 0};
  while (n_in != 0) {
    trace_level(trace, ctx.iteration, in, n_in);
    int64_t len = frontier_out_len(Ap, in, n_in);
    int32_t* out = (int32_t*)malloc(sizeof(int32_t) * (size_t)(len ? len : 1));
    orc_advance(n, Ap, Aj, Ax, in, n_in, sssp_edge_op, &ctx, out);
    if (trace) trace->edges_traversed += len;
    free(in);
    in = (int32_t*)malloc(sizeof(int32_t) * (size_t)(len ? len : 1));
    orc_filter_bypass(out, len, sssp_vertex_op, &ctx, in);
    free(out);
    n_in = len;
    ++ctx.iteration;
  }
  free(in);
  free(visited);
  if (trace) trace->iterations = ctx.iteration;
}

/* ------------------------------------------------------------------------
 * PageRank.  Follows include/gunrock/algorithms/pr.hxx literally, in float32:
 *   reset :64-92   p = 1/n, plast = 0, iweights[v] = alpha / (sum of out
 *                  weights) or 0 for a dangling vertex
 *   loop  :106-153 plast = p; dsum = sum over dangling v of alpha*p[v];
 *                  p[:] = (1 - alpha + dsum)/n; scatter p[dst] +=
 *                  plast[src]*iweights[src]*w over every edge
 *   stop  :155-178 after >= 1 iteration, when max|p - plast| < tol
 * The reference has no CPU checker and no test for PageRank: PARITY UNPINNED.
 * Float sums here run in index order; the device result depends on the order
 * in which the atomics land, so comparisons use a tolerance.
 * ---------------------------------------------------------------------- */
int32_t orc_pagerank(int32_t n, const int32_t* Ap, const int32_t* Aj, const float* Ax,
                     float alpha, float tol, int32_t max_iter, float* p) {
  float* plast = (float*)calloc((size_t)(n ? n : 1), sizeof(float));
  float* iw = (float*)malloc(sizeof(float) * (size_t)(n ? n : 1));
  float p0 = (float)(1.0 / n);
  for (int32_t v = 0; v < n; ++v) {
    p[v] = p0;
    float s = 0;
    for (int32_t e = Ap[v]; e < Ap[v + 1]; ++e) s += Ax[e];
    iw[v] = s != 0 ? alpha / s : 0;
  }
  int32_t it = 0;
  for (;;) {
    if (it != 0) {
      float err = 0;
      for (int32_t v = 0; v < n; ++v) {
        float d = fabsf(p[v] - plast[v]);
        if (d > err) err = d;
      }
      if (err < tol) break;
    }
    if (max_iter > 0 && it >= max_iter) break;
    memcpy(plast, p, sizeof(float) * (size_t)n);
    /* the reference reduces with thrust::transform_reduce (a tree), not a running float: a
     * sequential float32 sum of 1e7 terms of 1e-8 stagnates, so accumulate wide and round once */
    double dacc = 0;
    for (int32_t v = 0; v < n; ++v) dacc += (iw[v] == 0) ? (double)(alpha * p[v]) : 0.0;
    float dsum = (float)dacc;
    float fill = (1 - alpha + dsum) / n;
    for (int32_t v = 0; v < n; ++v) p[v] = fill;
    for (int32_t v = 0; v < n; ++v) {
      float base = plast[v] * iw[v];
      for (int32_t e = Ap[v]; e < Ap[v + 1]; ++e) p[Aj[e]] += base * Ax[e];
    }
    ++it;
  }
  free(plast);
  free(iw);
  return it;
}

/* ------------------------------------------------------------------------
 * Strong CPU baseline (not in the reference):
 * OpenMP level-synchronous top-down BFS with a CAS-free benign-race claim
 * (depth written only when it improves; all writers of a level write the same
 * value).
 * ---------------------------------------------------------------------- */
float orc_bfs_levelsync_omp(int32_t n, const int32_t* Ap, const int32_t* Aj, int32_t source,
                            int32_t* depth, int32_t* threads) {
  for (int32_t i = 0; i < n; ++i) depth[i] = INT32_MAX;
  int nt = 1;
#ifdef _OPENMP
  nt = omp_get_max_threads();
#endif
  if (threads) *threads = nt;
  int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
  int32_t* nxt = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
  double t0 = now_ms();
  depth[source] = 0;
  cur[0] = source;
  int64_t ncur = 1;
  int32_t level = 0;
  while (ncur) {
    int64_t nnext = 0;
#pragma omp parallel
    {
      int32_t local[1024];
      int nl = 0;
#pragma omp for schedule(dynamic, 64) nowait
      for (int64_t i = 0; i < ncur; ++i) {
        int32_t v = cur[i];
        for (int32_t e = Ap[v]; e < Ap[v + 1]; ++e) {
          int32_t nb = Aj[e];
          if (__atomic_load_n(&depth[nb], __ATOMIC_RELAXED) == INT32_MAX) {
            int32_t expect = INT32_MAX;
            if (__atomic_compare_exchange_n(&depth[nb], &expect, level + 1, 0,
                                            __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
              local[nl++] = nb;
              if (nl == 1024) {
                int64_t at = __atomic_fetch_add(&nnext, nl, __ATOMIC_RELAXED);
                memcpy(nxt + at, local, sizeof(int32_t) * (size_t)nl);
                nl = 0;
              }
            }
          }
        }
      }
      if (nl) {
        int64_t at = __atomic_fetch_add(&nnext, nl, __ATOMIC_RELAXED);
        memcpy(nxt + at, local, sizeof(int32_t) * (size_t)nl);
      }
    }
    int32_t* t = cur; cur = nxt; nxt = t;
    ncur = nnext;
    ++level;
  }
  double t1 = now_ms();
  free(cur);
  free(nxt);
  return (float)(t1 - t0);
}
