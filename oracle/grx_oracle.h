/*
 * grx_oracle.h -- CPU oracle for the frontier advance / filter / uniquify path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and there only as the checker.  The product (libessentials_amd.so)
 * never links, loads or calls it.
 *
 * Every function is a plain-C restatement of an algorithm of the reference
 * (jdwapman/essentials, mounted at /root/reference while the oracle is written);
 * the reference file:line each one follows is cited at its definition in
 * grx_oracle.c.  Parity pinning: the restatement is checked against (i) the
 * reference's own CPU checkers bfs_cpu.hxx / sssp_cpu.hxx compiled in place into
 * oracle/_ref (see oracle/ref_build.sh), (ii) the known answers the reference's
 * tests hold (chesapeake.mtx, io::sample::csr, the tc.cuh 4-vertex graphs).
 * PageRank has no CPU checker and no test in the reference: "parity unpinned".
 *
 * Types follow the reference harnesses: vertex_t = edge_t = int32, weight_t = float.
 */
#ifndef GRX_ORACLE_H
#define GRX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- loaders / formats ------------------------------------------------- */

/* Matrix-Market coordinate file -> COO (0-based), symmetric files expanded.
 * Arrays are malloc'ed; free with orc_free().  Returns 0 on success. */
int orc_mtx_load(const char* path, int32_t* n_rows, int32_t* n_cols, int32_t* nnz,
                 int32_t** row_idx, int32_t** col_idx, float** values);

/* COO -> CSR by stable counting sort on the row (duplicates kept). */
void orc_coo_to_csr(int32_t n_rows, int32_t nnz, const int32_t* row_idx,
                    const int32_t* col_idx, const float* values,
                    int32_t* row_offsets /* n_rows+1 */, int32_t* col_out, float* val_out);

/* The reference's ".csr" binary cache. */
int orc_csr_write_binary(const char* path, int32_t n_rows, int32_t n_cols, int32_t nnz,
                         const int32_t* row_offsets, const int32_t* col, const float* val);
int orc_csr_read_binary(const char* path, int32_t* n_rows, int32_t* n_cols, int32_t* nnz,
                        int32_t** row_offsets, int32_t** col, float** val);
void orc_free(void* p);

/* ---- synthetic input: R-MAT (the build's own spec, not from the reference) -- */

/* Pair k of a Graph500-style Kronecker graph; integer-exact so that the GPU
 * generator reproduces it bit for bit.  See DESIGN.md "RMAT specification". */
void orc_rmat_pair(uint32_t scale, uint64_t seed, uint64_t k, int32_t* u, int32_t* v);
/* Weight of pair k: integer valued float in [1, 64] (0 => every weight 1.0f). */
float orc_rmat_weight(uint64_t weight_seed, uint64_t k);
/* Number of directed edges after the loader-style symmetric expansion. */
int64_t orc_rmat_count(uint32_t scale, uint32_t edge_factor, uint64_t seed, int symmetrize);
/* Full CSR of the graph (arrays caller-allocated: n+1, nnz, nnz). */
void orc_rmat_csr(uint32_t scale, uint32_t edge_factor, uint64_t seed, uint64_t weight_seed,
                  int symmetrize, int32_t* row_offsets, int32_t* col, float* val);

/* ---- the reference's CPU checkers -------------------------------------- */

/* Heap-driven label-setting search; returns the milliseconds spent in the
 * search loop only (timer placement of the reference). */
float orc_bfs_heap(int32_t n, const int32_t* row_offsets, const int32_t* col,
                   int32_t source, int32_t* depth);
float orc_sssp_heap(int32_t n, const int32_t* row_offsets, const int32_t* col,
                    const float* val, int32_t source, float* dist);

/* ---- operator semantics ------------------------------------------------ */

typedef int (*orc_edge_op)(int32_t src, int32_t dst, int32_t edge, float w, void* ctx);
typedef int (*orc_vertex_op)(int32_t v, void* ctx);

/* block_mapped advance, "holes" layout: one output slot per traversed edge,
 * -1 where the op returned false.  input == NULL means advance_io_type_t::graph
 * (all vertices 0..n_in-1).  output == NULL means advance_io_type_t::none.
 * Returns the number of output slots (= sum of degrees of the valid inputs). */
int64_t orc_advance(int32_t n, const int32_t* row_offsets, const int32_t* col,
                    const float* val, const int32_t* input, int64_t n_in,
                    orc_edge_op op, void* ctx, int32_t* output);

/* filters: return the new number of elements */
int64_t orc_filter_bypass(const int32_t* in, int64_t n_in, orc_vertex_op op, void* ctx,
                          int32_t* out);
int64_t orc_filter_keep(const int32_t* in, int64_t n_in, orc_vertex_op op, void* ctx,
                        int32_t* out); /* predicated == remove == compact: stable */
/* uniquify: optional ascending sort, then drop consecutive duplicates */
int64_t orc_uniquify(int32_t* data, int64_t n, int do_sort);

/* ---- the three clients, driven through the operators above ------------- */

typedef struct {
  int32_t iterations;       /* number of loop() calls                          */
  int64_t edges_traversed;  /* sum over iterations of advance output slots     */
  int64_t frontier_slots[64];  /* input slots per iteration (first 64)         */
  int64_t frontier_valid[64];  /* valid input entries per iteration (first 64) */
} orc_trace;

void orc_bfs_frontier(int32_t n, const int32_t* row_offsets, const int32_t* col,
                      const float* val, int32_t source, int32_t* depth, orc_trace* trace);
void orc_sssp_frontier(int32_t n, const int32_t* row_offsets, const int32_t* col,
                       const float* val, int32_t source, float* dist, orc_trace* trace);
/* PageRank: returns iterations executed.  "parity unpinned" in the reference. */
int32_t orc_pagerank(int32_t n, const int32_t* row_offsets, const int32_t* col,
                     const float* val, float alpha, float tol, int32_t max_iter, float* p);

/* ---- strong CPU baseline (not in the reference) ------------------------ */
/* OpenMP level-synchronous top-down BFS; returns ms, *threads = threads used. */
float orc_bfs_levelsync_omp(int32_t n, const int32_t* row_offsets, const int32_t* col,
                            int32_t source, int32_t* depth, int32_t* threads);

#ifdef __cplusplus
}
#endif
#endif
