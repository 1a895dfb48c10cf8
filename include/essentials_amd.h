/*
 * essentials_amd.h -- C ABI of the MI355X-native frontier advance/filter engine.
 *
 * The reference (jdwapman/essentials) is a header-only C++ template library with
 * NO C ABI, FFI or plugin interface: its drop-in boundary is the template surface
 * under include/gunrock/ (this repository ships its own implementation of that
 * surface in include/gunrock/, against which the reference's bfs.hxx / sssp.hxx /
 * pr.hxx compile unchanged).  This header is therefore an ADDITION: the boundary
 * a non-C++ host (ctypes, cgo, JNI, ...) binds, with the engine's templates
 * pre-instantiated for the reference harnesses' types
 *     vertex_t = edge_t = int32_t, weight_t = float
 * (examples/algorithms/bfs/bfs.cu:16-18).  Each entry point names the reference
 * interface it stands for.  Conventions:
 *   - plain pointers and sizes only; every "d_" pointer is DEVICE memory owned by
 *     the caller (hipMalloc / a torch tensor's data_ptr), every "h_" pointer host;
 *   - return value 0 on success, a negative grx_status otherwise;
 *     grx_last_error() returns the message of the calling thread's last failure
 *     (the C++ surface throws gunrock::error::exception_t, reference error.hxx:21-46);
 *   - calls on one context are synchronous with respect to the host, like the
 *     reference's operators (advance/block_mapped.hxx:204);
 *   - one context per host thread; independent contexts do not share state
 *     (reference operators/batch/batch.hxx:69-74 relies on that).
 */
#ifndef ESSENTIALS_AMD_H
#define ESSENTIALS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRX_ABI_VERSION 1

typedef enum grx_status {
  GRX_OK = 0,
  GRX_ERR_INVALID_ARGUMENT = -1,
  GRX_ERR_RUNTIME = -2,      /* HIP / RCCL failure or an engine exception */
  GRX_ERR_UNSUPPORTED = -3,  /* variant not implemented (reference: "... not supported") */
  GRX_ERR_PEER = -4,         /* partitioned run: ANOTHER rank's superstep failed; every rank of the
                                job returns an error from the same superstep (the failing rank its own) */
  GRX_ERR_TIMEOUT = -5       /* partitioned run: the gathered counts of a superstep did not arrive
                                within GRX_PARTITIONED_TIMEOUT_MS (default 30 s) -- a peer is gone or
                                stuck.  The stream still holds the unfinished collective: the caller
                                should exit (not re-exec) without synchronising this context. */
} grx_status;

/* operators::load_balance_t (reference framework/operators/configs.hxx:31-39), same order */
typedef enum grx_load_balance {
  GRX_LB_THREAD_MAPPED = 0,
  GRX_LB_WARP_MAPPED = 1,
  GRX_LB_BLOCK_MAPPED = 2,
  GRX_LB_BUCKETING = 3,
  GRX_LB_MERGE_PATH = 4,
  GRX_LB_MERGE_PATH_V2 = 5,
  GRX_LB_WORK_STEALING = 6
} grx_load_balance;

/* operators::filter_algorithm_t (configs.hxx:73-78), same order */
typedef enum grx_filter_algorithm {
  GRX_FILTER_REMOVE = 0,
  GRX_FILTER_PREDICATED = 1,
  GRX_FILTER_COMPACT = 2,
  GRX_FILTER_BYPASS = 3
} grx_filter_algorithm;

/* operators::uniquify_algorithm_t (configs.hxx:80-85) */
typedef enum grx_uniquify_algorithm { GRX_UNIQUE = 0, GRX_UNIQUE_COPY = 1 } grx_uniquify_algorithm;

/* Built-in per-edge functors for grx_advance (device lambdas cannot cross a C ABI).
 * `state` / `iparam` are the functor's captured values. */
typedef enum grx_edge_op {
  GRX_OP_ALL = 0,        /* return true                                                     */
  GRX_OP_BFS = 1,        /* state: int32 depth[V]; it+1 < atomic::min(&depth[dst], it+1), it=iparam
                            (reference algorithms/bfs.hxx:92-114)                           */
  GRX_OP_SSSP = 2,       /* state: float dist[V]; d = dist[src]+w; d < atomic::min(&dist[dst], d)
                            (reference algorithms/sssp.hxx:110-124)                         */
  GRX_OP_COUNT_EDGE = 3, /* state: int32 calls[E]; atomic::add(&calls[edge],1); return (src+dst)%3==0 */
  GRX_OP_SUM_WEIGHT = 4  /* state: float acc[V]; atomic::add(&acc[dst], w); return false
                            (shape of reference algorithms/pr.hxx:140-146)                  */
} grx_edge_op;

/* Built-in predicates for grx_filter. */
typedef enum grx_vertex_op {
  GRX_PRED_ALL = 0,      /* return true                                                     */
  GRX_PRED_ODD = 1,      /* return v & 1                                                    */
  GRX_PRED_ONCE = 2,     /* state: int32 stamp[V]; stamp[v]==iparam ? false : (stamp[v]=iparam, true)
                            (reference algorithms/sssp.hxx:126-136)                         */
  GRX_PRED_COUNT = 3     /* state: int32 calls[V]; atomic::add(&calls[v],1); return v % 3 != 0 */
} grx_vertex_op;

typedef struct grx_context_s* grx_context_t; /* gcuda::multi_context_t, cuda/context.hxx:136-206 */
typedef struct grx_graph_s* grx_graph_t;     /* graph::graph_t over CSR, graph/build.hxx:26-36   */

/* enactor_properties_t (framework/enactor.hxx:31-54) + the engine's run-time switches */
typedef struct grx_options {
  int32_t load_balance;     /* grx_load_balance; default GRX_LB_BLOCK_MAPPED (what bfs.hxx spells) */
  int32_t holes_layout;     /* 1: one output slot per traversed edge, -1 holes (reference layout) */
  int32_t hub_threshold;    /* 0: default (256): longer lists are cut into chunks                */
  int32_t max_iterations;   /* 0: run to convergence                                              */
  float frontier_sizing_factor; /* 0: default 1.5 (enactor.hxx:36)                                */
  int32_t collect_kernel_time;  /* 1: event-time the advance kernels (adds two events per launch)  */
  int32_t chunk_edges;          /* 0: default (1024): edges per chunk of a hub list                */
  int32_t direction_optimized;  /* grx_bfs: 0 push every level (what bfs.hxx does); 1 pull the wide
                                   levels (advance_direction_t::backward).  grx_pagerank: 0 the push
                                   scatter of pr.hxx (one float atomic per edge); 1 the pull form (sums
                                   per destination over in-edges, no atomics).  Pulling needs in-edges:
                                   an undirected graph or grx_graph_build_in_edges                  */
  float do_alpha;               /* 0: default 4: pull when frontier edges > unexplored edges/alpha   */
  float do_beta;                /* 0: default 24: push again when frontier vertices < |V| / beta     */
  int32_t chunk_queue_limit;    /* test hook, 0: none. Caps the hub chunk queue to force the overflow
                                   path (hubs expanded in place)                                    */
  int32_t sssp_two_pass;        /* grx_sssp: 0 one pass -- the relax functor keeps exactly one copy of an
                                   improved vertex per round (one 64-bit atomic min on distance | round
                                   while the search runs; d_distances is written when it ends); 1 the
                                   reference client's formulation, advance + bypass filter with its racy
                                   stamp test (algorithms/sssp.hxx:110-144).  Same distances either way */
  int32_t call_every_edge;      /* grx_bfs push: 0 the search names its settled destinations (vertices that
                                   have a depth) and the engine skips the functor call for an edge into one
                                   (gunrock/framework/operators/settled.hxx; wide block_mapped levels);
                                   1 the functor is called for every edge, which is all the engine can do
                                   for the unchanged bfs.hxx.  Same depths either way                  */
} grx_options;

/* What enact() reports (framework/enactor.hxx:243-254 returns ms only; the rest is the
 * stats block util/info.hxx:36-71 names but never implemented). */
typedef struct grx_stats {
  float elapsed_ms;            /* device-event time around the BSP loop only           */
  float advance_kernel_ms;     /* sum of advance kernel durations (collect_kernel_time) */
  int32_t iterations;          /* loop() calls                                         */
  int32_t advance_launches;    /* advance kernel launches timed                        */
  int64_t vertices_reached;    /* BFS/SSSP: labels != unreached                        */
  int64_t edges_traversed;     /* BFS/SSSP: sum of out-degrees of reached vertices     */
  int32_t levels_recorded;     /* min(iterations, 64)                                  */
  int32_t pull_iterations;     /* direction-optimised BFS: levels expanded by pulling   */
  int64_t frontier_slots[64];  /* input-frontier length of each iteration              */
  int64_t edges_expanded;      /* BFS/SSSP push: functor calls = sum over iterations of the input
                                  frontier's degrees (SSSP relaxes a vertex again whenever its
                                  distance improved: > edges_traversed)                   */
} grx_stats;

/* ---- library ------------------------------------------------------------- */
int grx_abi_version(void);
const char* grx_last_error(void);

/* ---- context: gcuda::multi_context_t(device[, stream]) ------------------- */
/* stream: a hipStream_t to run on (e.g. torch.cuda.current_stream().cuda_stream) or NULL
 * for a private non-blocking stream (reference cuda/context.hxx:75-87,163-177). */
int grx_context_create(int device, void* stream, grx_context_t* out);
int grx_context_destroy(grx_context_t ctx);
int grx_context_synchronize(grx_context_t ctx);
/* Order everything already enqueued on `other_stream` (a hipStream_t; NULL = the null stream)
 * before the context's NEXT work, on the device (event record + stream wait, no host wait).  A
 * context created without a stream runs on a private NON-BLOCKING stream, which is not ordered
 * after the null stream or after any other: a host that fills operands on its own stream (torch
 * tensor ops) calls this before handing them to an engine entry point.  Results need no such call:
 * the entry points return after the context's stream has drained (except the enqueue-only
 * grx_partitioned_step / grx_partitioned_level_bitmap, which are meant to share the host's stream). */
int grx_context_wait_stream(grx_context_t ctx, void* other_stream);
/* Plain copies between host and device memory ON THE CONTEXT'S STREAM, complete when the call
 * returns (the stream is drained).  For bindings without a HIP runtime of their own -- e.g. a
 * host-staged collective callback (grx_context_attach_collectives). */
int grx_copy_to_host(grx_context_t ctx, void* h_dst, const void* d_src, size_t bytes);
int grx_copy_to_device(grx_context_t ctx, void* d_dst, const void* h_src, size_t bytes);
/* Return every device block the engine parked for reuse (frontier buffers of finished runs, up to
 * 64 GiB per process) to the device.  The engine does so by itself before it reports an
 * out-of-memory; a host that shares the device with another allocator calls this when THAT one
 * runs short. */
int grx_trim_cache(void);
int grx_context_device_info(grx_context_t ctx, int32_t* compute_units, int32_t* wavefront_size,
                            int64_t* total_memory_bytes, char* name, size_t name_len);

/* ---- graph: graph::build::from_csr<device, view_t::csr> ------------------ */
/* Non-owning view over caller-owned device CSR arrays (graph/graph.hxx:159-168).  The arrays must
 * be complete when the handle is first used (synchronise the stream that produced them) and must
 * not change while it lives: the handle reduces and keeps the graph's largest degree. */
int grx_graph_from_device_csr(int32_t n_rows, int32_t n_cols, int32_t nnz,
                              const int32_t* d_row_offsets, const int32_t* d_col,
                              const float* d_val, grx_graph_t* out);
/* Owning device copy of host CSR arrays (format::csr_t<device>, formats/csr.hxx:25-77). */
int grx_graph_from_host_csr(int32_t n_rows, int32_t n_cols, int32_t nnz,
                            const int32_t* h_row_offsets, const int32_t* h_col,
                            const float* h_val, grx_graph_t* out);
/* Matrix Market file -> owning device CSR (io/matrix_market.hxx:99-240 + csr.hxx:79-157). */
int grx_graph_from_mtx(const char* path, grx_graph_t* out);
/* ".csr" binary cache (formats/csr.hxx:159-236). */
int grx_graph_from_csr_file(const char* path, grx_graph_t* out);
int grx_graph_write_csr_file(grx_graph_t g, const char* path);
/* Synthetic R-MAT (no reference counterpart; spec in DESIGN.md, oracle in oracle/grx_oracle.c):
 * 2^scale vertices, edge_factor*2^scale generated pairs, loader-style symmetrisation when
 * symmetrize != 0, weights 1.0f when weight_seed == 0 else integers in [1,64]. Built on the GPU. */
int grx_graph_rmat(grx_context_t ctx, uint32_t scale, uint32_t edge_factor, uint64_t seed,
                   uint64_t weight_seed, int symmetrize, grx_graph_t* out);
/* Copy of `g` whose neighbour lists are sorted by column id (duplicates adjacent, weights carried
 * along): the layout a column-major-sorted Matrix Market file (SuiteSparse convention) gets from
 * the reference's stable row sort (formats/csr.hxx:119-147).  The R-MAT generator above keeps the
 * EMISSION order inside a row (an unsorted edge-list file); this utility produces the other
 * common layout of the same graph. */
int grx_graph_sorted_rows(grx_context_t ctx, grx_graph_t g, grx_graph_t* out);
/* Build and attach the in-edge (transpose / csc) arrays of a DIRECTED graph so that pull advances
 * (grx_options.direction_optimized) can walk in-neighbours; graphs without them are taken to be
 * undirected (symmetric CSR = its own transpose).  Reference counterpart: the csc view of
 * graph::build::from_csr (graph/detail/build.hxx:96-113), which it cannot combine with csr. */
int grx_graph_build_in_edges(grx_context_t ctx, grx_graph_t g);
/* Hot-first numbering (engine data layout, no reference counterpart; the reference's views keep
 * the loader's numbering, graph/build.hxx:26-52).  grx_bfs / grx_sssp run on a copy of the graph
 * whose vertices are renumbered in descending order of out-degree
 * (include/gunrock/graph/reorder.hxx) and deliver their labels in the CALLER's numbering: sources
 * in, labels out, nothing else changes at this boundary.  The copy (a second CSR + 8 bytes per
 * vertex) is built on the first traversal of a square graph of >= 2^16 vertices and >= 2^20 edges
 * without attached in-edges (GRX_HOT_FIRST=0/1 overrides the size rule); enable = 1 builds it now
 * for any square graph, enable = 0 drops it and keeps this handle on its own numbering.  The
 * options that stand for the unchanged reference clients (call_every_edge, sssp_two_pass) and the
 * holes layout always run on the caller's numbering. */
int grx_graph_hot_first(grx_context_t ctx, grx_graph_t g, int enable);
int grx_graph_destroy(grx_graph_t g);
int grx_graph_info(grx_graph_t g, int32_t* n_rows, int32_t* n_cols, int64_t* nnz,
                   const int32_t** d_row_offsets, const int32_t** d_col, const float** d_val);
int grx_graph_copy_to_host(grx_graph_t g, int32_t* h_row_offsets, int32_t* h_col, float* h_val);

/* ---- algorithms: gunrock::{bfs,sssp,pr}::run ------------------------------ */
void grx_default_options(grx_options* opt);
/* bfs::run(G, source, distances, predecessors, context)  algorithms/bfs.hxx:151-176.
 * d_distances: int32[V] out (INT32_MAX = unreached). d_predecessors: unused by the reference
 * ("@todo", bfs.hxx:28), may be NULL. */
int grx_bfs(grx_context_t ctx, grx_graph_t g, int32_t source, int32_t* d_distances,
            int32_t* d_predecessors, const grx_options* opt, grx_stats* stats);
/* sssp::run(G, source, distances, predecessors, context)  algorithms/sssp.hxx:155-185.
 * d_distances: float[V] out (FLT_MAX = unreached). */
int grx_sssp(grx_context_t ctx, grx_graph_t g, int32_t source, float* d_distances,
             int32_t* d_predecessors, const grx_options* opt, grx_stats* stats);
/* pr::run(G, alpha, tol, p, context)  algorithms/pr.hxx:182-216.  d_p: float[V] out. */
int grx_pagerank(grx_context_t ctx, grx_graph_t g, float alpha, float tol, float* d_p,
                 const grx_options* opt, grx_stats* stats);

/* ---- operators (frontier-level overloads) -------------------------------- */
/* operators::advance::execute<lb, forward, in, out>(G, op, input, output, segments, context)
 * (framework/operators/advance/advance.hxx:91-129).
 *   d_input == NULL  -> advance_io_type_t::graph (all n_rows vertices)
 *   d_output == NULL -> advance_io_type_t::none
 * *n_output receives the new number of elements; the output holds them in unspecified
 * order (block_mapped.hxx:94-101). Fails with GRX_ERR_RUNTIME if output_capacity is too small. */
int grx_advance(grx_context_t ctx, grx_graph_t g, const grx_options* opt, int32_t edge_op,
                void* d_state, int32_t iparam, const int32_t* d_input, int64_t n_input,
                int32_t* d_output, int64_t output_capacity, int64_t* n_output);
/* operators::filter::execute<alg>(G, op, input, output, context)  filter/filter.hxx:59-86 */
int grx_filter(grx_context_t ctx, grx_graph_t g, int32_t algorithm, int32_t vertex_op,
               void* d_state, int32_t iparam, const int32_t* d_input, int64_t n_input,
               int32_t* d_output, int64_t output_capacity, int64_t* n_output);
/* operators::uniquify::execute<alg>(input, output, context, best_effort)  uniquify.hxx:15-42.
 * GRX_UNIQUE leaves the result in d_input (d_output is scratch of >= n_input elements). */
int grx_uniquify(grx_context_t ctx, int32_t algorithm, int32_t best_effort, int32_t* d_input,
                 int64_t n_input, int32_t* d_output, int64_t output_capacity, int64_t* n_output);

/* ---- multi-GPU: one process per GPU, frontier all-gather over RCCL/xGMI ------------------ */
/* The reference declares multi_context_t for several devices but every operator throws for
 * size() != 1 (advance.hxx:125-128); this is new functionality.
 *
 * Model: 1-D vertex partition.  Rank r owns rows [row_begin,row_end); its local CSR keeps GLOBAL
 * vertex ids and has all V rows, the rows it does not own being empty.  Every rank holds a
 * replica of the label array (BFS depth / SSSP distance).  One BSP superstep =
 *   grx_partitioned_expand   local advance over the owned input frontier (any schedule) with the
 *                            BFS / SSSP functor on the replica; the vertices it improved are
 *                            packed as [count | (vertex,label) ...] into the rank's send slot
 *   all-gather of the send slots over RCCL (done by the host: torch.distributed / ncclAllGather)
 *   grx_partitioned_admit    min-combine every other rank's pairs into the replica and append
 *                            the owned, improved vertices to the next input frontier
 * Dense BFS supersteps (more finds than V/64) exchange per-rank LEVEL BITMAPS instead of pairs
 * (grx_partitioned_level_bitmap + recv_format GRX_RECV_LEVEL_BITMAP).
 * grx_partitioned_run (below) is the whole loop in C++ with the collectives issued directly on
 * RCCL; the step-level entry points stay public for hosts that bring their own loop. */

/* Rank-local slice of a replicated graph; split points balance EDGES (prefix of row offsets). */
int grx_graph_partition(grx_graph_t full, int rank, int world_size, grx_graph_t* out,
                        int32_t* row_begin, int32_t* row_end);
/* The same slice of the graph's HOT-FIRST renumbered copy (grx_graph_hot_first; built here if the
 * handle has none yet): [row_begin, row_end) is then a range of RENUMBERED vertices (edge-balanced
 * like above: the first ranks own few, heavy vertices), and the slice remembers both permutations.
 * grx_partitioned_run on a plan made from it takes the source and delivers the labels in the
 * CALLER's numbering (sources in, labels out, as for grx_bfs): the supersteps run on renumbered
 * replicas, one scatter pass at the end of the run writes d_labels.  The single-superstep entry
 * points below (grx_partitioned_expand / _admit / _step) see such a slice as the plain graph it is --
 * renumbered ids in and out. */
int grx_graph_partition_hot_first(grx_context_t ctx, grx_graph_t full, int rank, int world_size,
                                  grx_graph_t* out, int32_t* row_begin, int32_t* row_end);
/* d_labels: replica [V] (int32 depth for GRX_OP_BFS, float distance for GRX_OP_SSSP).
 * round: superstep number (the BFS level).  d_frontier / n_frontier: owned input frontier
 * (global ids).  d_scratch: int32[scratch_capacity] workspace for the raw output frontier
 * (>= sum of degrees of the frontier; local nnz + V is enough for a duplicate-free frontier).
 * d_sent_stamp: int32[V], -1 initially: a vertex improved several times in one superstep is
 * packed once.  d_send: int64[send_capacity >= V + 1]; word 0 receives the number of pairs,
 * words 1.. the pairs (low 32 bits vertex, high 32 bits label bits).  *n_found = that count. */
int grx_partitioned_expand(grx_context_t ctx, grx_graph_t local, const grx_options* opt,
                           int32_t edge_op, void* d_labels, int32_t round,
                           const int32_t* d_frontier, int64_t n_frontier, int32_t* d_scratch,
                           int64_t scratch_capacity, int32_t* d_sent_stamp, int64_t* d_send,
                           int64_t send_capacity, int64_t* n_found);
/* What a gathered buffer holds. */
typedef enum grx_recv_format {
  GRX_RECV_PAIRS = 0,        /* per rank: [count | (vertex,label) ...], `slot` words each           */
  GRX_RECV_LEVEL_BITMAP = 1, /* BFS only, dense supersteps: per rank ceil(V/64) words, bit v = "this
                                rank discovered v in the superstep"; labels are implied (the level) */
  GRX_RECV_REPLICA_MIN = 2   /* dense SSSP supersteps: nothing was gathered -- the host all-reduced
                                (MIN) the label replicas in place; d_recv is the label SNAPSHOT
                                (one entry per vertex) grx_partitioned_step took before the advance:
                                owned vertices whose label fell below it are admitted             */
} grx_recv_format;

/* Dense BFS exchange, enqueue-only: d_words[ceil(V/64)] <- bit v = (d_depth[v] == level).  Called
 * after a superstep whose finds (level = round + 1) are too many for the pair list: V/8 bytes per
 * rank travel instead of 8 bytes per discovery. */
int grx_partitioned_level_bitmap(grx_context_t ctx, const int32_t* d_depth, int64_t n_vertices,
                                 int32_t level, int64_t* d_words, int64_t word_capacity);

/* d_recv: int64[world_size * slot] gathered send slots (recv_format says what they hold).
 * GRX_RECV_PAIRS: pairs of other ranks are min-combined into d_labels; a vertex is appended to
 * d_next (capacity next_capacity) when it is owned (row_begin <= v < row_end), its label improved
 * (or it is one of this rank's own discoveries) and d_stamp[v] != round (then d_stamp[v] = round:
 * one copy per superstep, like the bypass filter of reference algorithms/sssp.hxx:126-136).
 * GRX_RECV_LEVEL_BITMAP: every vertex in the union of the bitmaps gets depth round + 1; the owned
 * ones are appended (ascending).  `round` = the superstep whose finds were gathered. */
int grx_partitioned_admit(grx_context_t ctx, int32_t edge_op, void* d_labels, int64_t n_vertices,
                          int32_t* d_stamp, int32_t round, const int64_t* d_recv,
                          int32_t recv_format, int32_t world_size, int64_t slot, int32_t my_rank,
                          int32_t row_begin, int32_t row_end, int32_t* d_next,
                          int64_t next_capacity, int64_t* n_next);

/* The same superstep FUSED and enqueue-only (block_mapped schedule): [admit d_recv, the gather of
 * superstep round - 1, into d_frontier / *d_frontier_count ->] advance over d_frontier (its length
 * is read on the DEVICE from *d_frontier_count) -> pack into d_send.  Nothing is awaited: the
 * caller issues the collective on the SAME stream (create the context on that stream) and
 * synchronises once per superstep, on the gathered counts.  d_recv == NULL on the first superstep
 * (the caller preset d_frontier / *d_frontier_count).  Buffer overflows of a step are reported by
 * the next call.  d_snapshot (optional, 4 bytes per vertex): receives the labels of the owned range
 * as they are before the advance, for a GRX_RECV_REPLICA_MIN admission by the next call. */
int grx_partitioned_step(grx_context_t ctx, grx_graph_t local, const grx_options* opt,
                         int32_t edge_op, void* d_labels, int32_t* d_stamp, int32_t* d_sent_stamp,
                         int32_t round, const int64_t* d_recv, int32_t recv_format,
                         int32_t world_size, int64_t slot, int32_t my_rank, int32_t row_begin,
                         int32_t row_end, int32_t* d_frontier, int64_t frontier_capacity,
                         uint64_t* d_frontier_count, int32_t* d_scratch, int64_t scratch_capacity,
                         int64_t* d_send, int64_t send_capacity, void* d_snapshot);

/* PageRank on the same partition (replicas of the rank vector p): one iteration's
 * local half.  d_partial[V + 1] <- contributions of the rows this rank owns,
 *   d_partial[dst] += d_rank[src] * d_scale[src] * w   over the local edges   (pr.hxx:140-146)
 *   d_partial[V]    = alpha * (sum of d_rank over the OWNED vertices without out-edges)
 * d_scale[V] (alpha / sum of out-weights, 0 for rows without edges; pr.hxx:77-91) is computed when
 * compute_scale != 0 and reused otherwise.  The host all-reduces d_partial (SUM) and sets
 *   p[v] = (1 - alpha + partial[V]) / V + partial[v];  done when max |p - p_previous| < tol.
 * Synchronous. */
int grx_pagerank_partitioned_scatter(grx_context_t ctx, grx_graph_t local, float alpha,
                                     const float* d_rank, float* d_scale, int32_t compute_scale,
                                     float* d_partial, int32_t row_begin, int32_t row_end,
                                     const grx_options* opt);

/* ---- multi-GPU: the job, and the whole traversal as ONE call -------------------------------- */
/* A context joins a job of `world_size` ranks (one process per GPU; this IS
 * gcuda::multi_context_t::attach_job).  Two transports:
 *   grx_context_attach_rccl         RCCL over xGMI (production): ncclCommInitRank with the 128-byte
 *                                   id rank 0 obtained from grx_job_unique_id and handed to the
 *                                   other ranks by any side channel; collective over all ranks.
 *                                   ncclAllGather / ncclAllReduce are then issued by the C++
 *                                   superstep loop on the context's own stream.
 *   grx_context_attach_collectives  host callbacks with the same meaning (device pointers in,
 *                                   device pointers out; the callback may stage through the host
 *                                   but must have completed when it returns).  For transports
 *                                   other than RCCL and for test rigs (ranks sharing one GPU). */
#define GRX_UNIQUE_ID_BYTES 128
typedef enum grx_collective_dtype { GRX_INT32 = 0, GRX_FLOAT32 = 1, GRX_INT64 = 2 } grx_collective_dtype;
typedef enum grx_collective_op { GRX_MIN = 0, GRX_SUM = 1, GRX_MAX = 2 } grx_collective_op;
typedef int (*grx_all_gather_fn)(void* user, const void* d_send, void* d_recv,
                                 uint64_t bytes_per_rank, void* stream);
typedef int (*grx_all_reduce_fn)(void* user, void* d_buffer, uint64_t count, int32_t dtype,
                                 int32_t op, void* stream);
int grx_job_unique_id(void* id128);
int grx_context_attach_rccl(grx_context_t ctx, int rank, int world_size, const void* id128);
int grx_context_attach_collectives(grx_context_t ctx, int rank, int world_size,
                                   grx_all_gather_fn all_gather, grx_all_reduce_fn all_reduce,
                                   void* user);
int grx_context_detach(grx_context_t ctx);
/* backend: "single" (no job), "rccl" or "hooks". */
int grx_context_job_info(grx_context_t ctx, int32_t* rank, int32_t* world_size, char* backend,
                         size_t backend_len);

/* A partitioned traversal's persistent state: the rank's slice, its owned range and every buffer
 * the supersteps need (frontier, raw output, send / receive slots, level bitmaps, stamps), all
 * allocated once.  small_slot: int64 words per rank in the first all-gather of a superstep
 * (0 = 32768, i.e. 256 KiB); dense_threshold / replica_threshold: finds on the busiest rank
 * above which BFS exchanges level bitmaps / SSSP all-reduces the replicas (0 = V/64 and
 * V/world_size; negative = never). */
typedef struct grx_partitioned_s* grx_partitioned_t;
typedef struct grx_partitioned_stats {
  float elapsed_ms;             /* host wall time of the superstep loop (device drained)        */
  int32_t supersteps;
  int32_t collectives;          /* all-gathers + all-reduces issued by this rank                 */
  int32_t bitmap_supersteps;    /* BFS supersteps that exchanged level bitmaps                   */
  int32_t allreduce_supersteps; /* SSSP supersteps that all-reduced (MIN) the replicas           */
  int32_t iterations;           /* PageRank: iterations                                          */
  float last_error;             /* PageRank: max |p - p_previous| of the last iteration          */
  int64_t pairs_exchanged;      /* finds of all ranks over the run                               */
  int64_t bytes_sent;           /* payload bytes this rank contributed to collectives            */
  int32_t large_gather_supersteps; /* supersteps whose pairs outgrew the first (small) slot: one
                                      more all-gather; collectives == supersteps + bitmap_ +
                                      allreduce_ + large_gather_supersteps for a traversal        */
  int32_t reserved;
} grx_partitioned_stats;
int grx_partitioned_create(grx_context_t ctx, grx_graph_t local, int32_t row_begin, int32_t row_end,
                           const grx_options* opt, int64_t small_slot, int64_t dense_threshold,
                           int64_t replica_threshold, grx_partitioned_t* out);
int grx_partitioned_destroy(grx_partitioned_t plan);
/* The whole BSP loop in C++ (what essentials_amd/distributed.py drove from Python in round 1):
 * reset the replica and the stamps, then per superstep ONE enqueue-only grx_partitioned_step, the
 * all-gather of the send slots on the SAME stream, one host wait on the gathered counts, and --
 * by the busiest rank's count -- the level-bitmap all-gather (BFS), the replica all-reduce (SSSP)
 * or a second, larger all-gather.  edge_op: GRX_OP_BFS (d_labels int32[V]) or GRX_OP_SSSP
 * (float[V]); every rank passes the same source and gets the full label array.  Collective over
 * the job the context is attached to (a context without a job runs it as a job of one). */
int grx_partitioned_run(grx_partitioned_t plan, int32_t edge_op, int32_t source, void* d_labels,
                        grx_partitioned_stats* stats);
/* PageRank on the same plan: per iteration the local scatter (grx_pagerank_partitioned_scatter),
 * ONE all-reduce (SUM) of V + 1 floats, the update and the stop test
 * max |p - p_previous| < tol after >= 1 iteration (pr.hxx:155-178); max_iterations 0 = none. */
int grx_partitioned_pagerank(grx_partitioned_t plan, float alpha, float tol, int32_t max_iterations,
                             float* d_p, grx_partitioned_stats* stats);

/* ---- measurement helpers ------------------------------------------------- */
/* Streaming copy of `bytes` (16 B per lane) timed with events on the context stream:
 * the achievable-HBM roof quoted beside the 8 TB/s vendor peak. Returns GB/s. */
int grx_measure_copy_bandwidth(grx_context_t ctx, size_t bytes, int repeats, double* gbps);

/* Random-gather ceiling of a graph: acc += table[column_indices[i]] over ALL its edges (4-B table
 * entries, one per vertex) -- the access every label-testing advance functor shares (reference
 * bfs.hxx:92-107, sssp.hxx:95-113), without frontier logic, atomics or output.  mode 0 = plain
 * (L1-cached) loads, 1 = the agent-scope loads math::atomic::min pre-tests with.  Best of
 * `repeats`; returns lookups (= edges) per second: the roof MTEPS of a push traversal is
 * measured against. */
int grx_measure_gather_rate(grx_context_t ctx, grx_graph_t g, int mode, int repeats,
                            double* lookups_per_second);

#ifdef __cplusplus
}
#endif
#endif /* ESSENTIALS_AMD_H */
