/**
 * @file partitioned.hip
 * @brief C ABI: the device side of vertex-partitioned (multi-GPU) traversals --
 * graph slicing, the local superstep (advance + pack) and the admission kernel.
 * The collective between the two is issued by the host (torch.distributed /
 * RCCL all-gather), see essentials_amd/distributed.py and include/essentials_amd.h.
 *
 * No reference counterpart: the reference's operators throw for more than one
 * context (advance.hxx:125-128); design per SURVEY.md 8(e).
 */
#include "capi_internal.hxx"

#include <gunrock/hip/kernels/exchange_kernels.hxx>

#include <algorithm>
#include <vector>

using namespace essentials_amd;

namespace {

__global__ void __launch_bounds__(256)
    slice_offsets_kernel(const int32_t* full, int32_t n, int32_t lo, int32_t hi, int32_t* local) {
  const int32_t base = full[lo];
  const int32_t top = full[hi];
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i <= n; i += (int64_t)gridDim.x * 256) {
    int32_t o = full[i];
    o = o < base ? base : (o > top ? top : o);
    local[i] = o - base;  // rows outside [lo,hi) become empty
  }
}

using hip::kernels::APPEND_ITEMS;
using hip::kernels::APPEND_TILE;
using hip::kernels::tile_append_t;
using hip::kernels::append_tile;
using hip::kernels::pack_pairs_kernel;
using hip::kernels::admit_kernel;

/// Dense BFS exchange: bit v of `words` = (depth[v] == level), i.e. "this rank discovered v in
/// the superstep that just ran".  One coalesced ballot pass, no atomics; V/8 bytes per rank
/// travel instead of 8 bytes per discovery.
__global__ void __launch_bounds__(256)
    level_bitmap_kernel(const int32_t* depth, int64_t n, int32_t level, unsigned long long* words) {
  const int64_t padded = (n + 63) / 64 * 64;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < padded; i += (int64_t)gridDim.x * 256) {
    const bool in = i < n && depth[i] == level;
    const unsigned long long m = __ballot(in);
    if ((threadIdx.x & 63) == 0)
      words[i / 64] = m;
  }
}

/// Admission from the gathered level bitmaps (world x words_per_rank): the union of all ranks'
/// discoveries gets depth = level in this replica; the owned ones form the next frontier
/// (ascending).  A vertex reached in an earlier level is in nobody's bitmap (replicas agree after
/// every superstep), so every owned set bit is appended exactly once.
__global__ void __launch_bounds__(256)
    admit_bitmap_kernel(int32_t* depth, int64_t n, int32_t level, const unsigned long long* recv,
                        int32_t world, int64_t words_per_rank, int32_t lo, int32_t hi, int32_t* next,
                        unsigned long long next_capacity, unsigned long long* next_count,
                        unsigned long long* overflow) {
  __shared__ unsigned wave_totals[256 / hip::wave_size + 1];
  __shared__ unsigned long long s_base;
  const int64_t n_words = (n + 63) / 64;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t rounds = (n_words + stride - 1) / stride;
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t w = r * stride + blockIdx.x * 256ll + threadIdx.x;
    unsigned long long m = 0;
    if (w < n_words)
      for (int32_t p = 0; p < world; ++p)
        m |= recv[(int64_t)p * words_per_rank + w];
    // owned part of this word: vertices [lo, hi)
    unsigned long long own = m;
    const int64_t v0 = w * 64;
    if (v0 + 64 <= lo || v0 >= hi) {
      own = 0;
    } else {
      if (v0 < lo)
        own &= ~0ull << (lo - v0);
      if (v0 + 64 > hi)
        own &= ~0ull >> (v0 + 64 - hi);
    }
    unsigned total = 0;
    unsigned at = hip::block_exclusive_sum<256>((unsigned)__popcll(own), total, wave_totals);
    if (total) {  // workgroup-uniform
      if (threadIdx.x == 0)
        s_base = atomicAdd(next_count, (unsigned long long)total);
      __syncthreads();
    }
    unsigned long long bits = m;
    while (bits) {
      const int b = __ffsll((long long)bits) - 1;
      bits &= bits - 1;
      const int64_t v = v0 + b;
      if (depth[v] > level)
        depth[v] = level;
    }
    if (total) {
      unsigned long long pos = s_base + at;
      while (own) {
        const int b = __ffsll((long long)own) - 1;
        own &= own - 1;
        if (pos < next_capacity)
          next[pos] = (int32_t)(v0 + b);
        else
          *overflow = 1ull;
        ++pos;
      }
      __syncthreads();  // s_base is rewritten by the next round
    }
  }
}

/// Admission after an all-reduce(MIN) of the label replicas (dense SSSP supersteps): an owned
/// vertex enters the next frontier when its label is now below the snapshot taken before the
/// superstep's advance -- whoever lowered it.  One pass over the owned range, exactly once each.
template <typename label_t>
__global__ void __launch_bounds__(256)
    admit_replica_kernel(const label_t* labels, const label_t* snapshot, int32_t lo, int32_t hi,
                         int32_t* next, unsigned long long next_capacity,
                         unsigned long long* next_count, unsigned long long* overflow) {
  __shared__ tile_append_t<int32_t> lds;
  const int64_t total = (int64_t)hi - lo;
  const int64_t stride = (int64_t)gridDim.x * APPEND_TILE;
  const int64_t rounds = (total + stride - 1) / stride;
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t tile = r * stride + (int64_t)blockIdx.x * APPEND_TILE;
    int32_t admitted[APPEND_ITEMS];
    unsigned keep = 0;
#pragma unroll
    for (int k = 0; k < APPEND_ITEMS; ++k) {
      const int64_t t = tile + k * 256 + threadIdx.x;
      admitted[k] = -1;
      if (t < total) {
        const int32_t v = lo + (int32_t)t;
        if (labels[v] < snapshot[v]) {
          admitted[k] = v;
          keep |= 1u << k;
        }
      }
    }
    append_tile(lds, admitted, keep, next, next_capacity, next_count, overflow);
  }
}

inline unsigned bitmap_grid(int64_t n_items, int cus) {
  const int64_t g = (n_items + 255) / 256;
  const int64_t cap = (int64_t)cus * 8;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

template <typename label_t>
int expand_as(grx_context_t ctx, grx_graph_t local, const grx_options& o, int32_t edge_op,
              label_t* labels, int32_t iparam, const int32_t* d_frontier, int64_t n_frontier,
              int32_t* d_scratch, int64_t scratch_capacity, int32_t* d_sent, int64_t* d_send,
              int64_t send_capacity, int64_t* n_found) {
  return with_load_balance(o.load_balance, [&](auto lb_tag) -> int {
    constexpr auto lb = decltype(lb_tag)::value;
    using operators::advance_direction_t;
    using operators::advance_io_type_t;
    auto& sc = ctx->single();
    scoped_options scope(sc, &o);
    sc.options().holes_layout = false;  // the exchange needs packed discoveries
    graph_type G = local->view();
    auto fin = frontier_type::wrap(const_cast<int32_t*>(d_frontier), (std::size_t)n_frontier,
                                   (std::size_t)(n_frontier ? n_frontier : 1));
    // the owned frontier is duplicate-free (admit_kernel), so its work is bounded by the rank's
    // edge count: no sizing pass
    fin.set_work_hint((unsigned long long)local->nnz);
    auto fout = frontier_type::wrap(d_scratch, 0, (std::size_t)scratch_capacity);
    hip::device_array_t<edge_t> segments;
    if (edge_op == GRX_OP_BFS) {
      int* depth = reinterpret_cast<int*>(labels);
      const int next_level = iparam + 1;
      auto visit = [depth, next_level] __host__ __device__(vertex_t const& src, vertex_t const& dst,
                                                           edge_t const& e, weight_t const& w) -> bool {
        return next_level < math::atomic::min(&depth[dst], next_level);
      };
      operators::advance::execute<lb, advance_direction_t::forward, advance_io_type_t::vertices,
                                  advance_io_type_t::vertices>(G, visit, &fin, &fout, segments,
                                                               *ctx->mc);
    } else {
      float* dist = reinterpret_cast<float*>(labels);
      // one copy of an improved vertex per superstep (the first improver wins the exchange on
      // its `sent` stamp): the raw output is duplicate-free; later improvements still lower
      // dist[dst] and the pack reads the latest value
      int32_t* sent = d_sent;
      const int32_t rnd = iparam;
      auto relax = [dist, sent, rnd] __host__ __device__(vertex_t const& src, vertex_t const& dst,
                                                         edge_t const& e, weight_t const& w) -> bool {
        float through = thread::load(&dist[src]) + w;
        if (!(through < math::atomic::min(&dist[dst], through)))
          return false;
        return math::atomic::exch(&sent[dst], rnd) != rnd;
      };
      operators::advance::execute<lb, advance_direction_t::forward, advance_io_type_t::vertices,
                                  advance_io_type_t::vertices>(G, relax, &fin, &fout, segments,
                                                               *ctx->mc);
    }
    const int64_t count = (int64_t)fout.get_number_of_elements();
    auto& ws = sc.workspace();
    unsigned long long* counters = ws.counters();
    // counters[C_SELECT] is zero here (the advance's hand-off cleared every counter)
    if (count) {
      const unsigned grid = (unsigned)std::min<int64_t>((count + APPEND_TILE - 1) / APPEND_TILE,
                                                        (int64_t)sc.compute_units() * 8);
      pack_pairs_kernel<label_t><<<grid, 256, 0, sc.stream()>>>(d_scratch, count, labels, d_send,
                                                                 send_capacity, counters);
      GRX_HIP_CHECK(hipGetLastError());
    }
    // the hand-off also leaves the pair count in the slot's header word
    unsigned long long* m = operators::advance::detail::await_counters(
        sc, operators::advance::detail::publish_counters(sc, reinterpret_cast<long long*>(d_send),
                                                         hip::kernels::C_SELECT));
    error::throw_if_exception(m[hip::kernels::C_OVERFLOW] != 0,
                              "grx_partitioned_expand: send buffer too small (needs V + 1 words)");
    *n_found = (int64_t)m[hip::kernels::C_SELECT];
    return (int)GRX_OK;
  });
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
/// Rank `rank`'s slice of `full` (rows outside [lo, hi) empty, global ids).
std::unique_ptr<grx_graph_s> slice_of(grx_graph_s* full, int rank, int world, int32_t& lo, int32_t& hi) {
  const int32_t n = full->n_rows;
  std::vector<int32_t> ap((std::size_t)n + 1);
  GRX_HIP_CHECK(hipMemcpy(ap.data(), full->d_ap, ap.size() * 4, hipMemcpyDeviceToHost));
  // edge-balanced split points: first row whose offset reaches k * E / world
  auto split = [&](int k) -> int32_t {
    if (k <= 0) return 0;
    if (k >= world) return n;
    const int64_t target = (int64_t)ap[n] * k / world;
    return (int32_t)(std::lower_bound(ap.begin(), ap.end(), (int32_t)target) - ap.begin());
  };
  lo = split(rank);
  hi = split(rank + 1);
  if (lo > n) lo = n;
  if (hi > n) hi = n;
  if (hi < lo) hi = lo;
  auto g = std::make_unique<grx_graph_s>();
  g->n_rows = n;
  g->n_cols = full->n_cols;
  g->nnz = (int64_t)ap[hi] - ap[lo];
  g->ap.resize((std::size_t)n + 1);
  g->aj.resize((std::size_t)std::max<int64_t>(g->nnz, 1));
  g->ax.resize((std::size_t)std::max<int64_t>(g->nnz, 1));
  slice_offsets_kernel<<<1024, 256>>>(full->d_ap, n, lo, hi, g->ap.data());
  GRX_HIP_CHECK(hipGetLastError());
  if (g->nnz) {
    GRX_HIP_CHECK(hipMemcpy(g->aj.data(), full->d_aj + ap[lo], (std::size_t)g->nnz * 4,
                            hipMemcpyDeviceToDevice));
    GRX_HIP_CHECK(hipMemcpy(g->ax.data(), full->d_ax + ap[lo], (std::size_t)g->nnz * 4,
                            hipMemcpyDeviceToDevice));
  }
  GRX_HIP_CHECK(hipDeviceSynchronize());
  g->adopt();
  g->hot_first = 0;  // a slice is traversed as it is
  return g;
}
}  // namespace

extern "C" {

int grx_graph_partition(grx_graph_t full, int rank, int world, grx_graph_t* out,
                        int32_t* row_begin, int32_t* row_end) {
  if (!full || !out || world < 1 || rank < 0 || rank >= world)
    return invalid("grx_graph_partition: bad arguments");
  return guarded([&] {
    int32_t lo = 0, hi = 0;
    auto g = slice_of(full, rank, world, lo, hi);
    if (row_begin) *row_begin = lo;
    if (row_end) *row_end = hi;
    *out = g.release();
    return (int)GRX_OK;
  });
}

int grx_graph_partition_hot_first(grx_context_t ctx, grx_graph_t full, int rank, int world,
                                  grx_graph_t* out, int32_t* row_begin, int32_t* row_end) {
  if (!ctx || !full || !out || world < 1 || rank < 0 || rank >= world)
    return invalid("grx_graph_partition_hot_first: bad arguments");
  return guarded([&] {
    if (full->n_rows != full->n_cols || full->in_edges)
      return unsupported("grx_graph_partition_hot_first: needs a square graph without attached in-edges");
    const int before = full->hot_first;
    if (!full->hot) {
      full->hot_first = 1;
      hot_copy(ctx, full);
      full->hot_first = before;  // the handle's own traversals keep their rule
    }
    error::throw_if_exception(!full->hot, "grx_graph_partition_hot_first: no renumbered copy");
    int32_t lo = 0, hi = 0;
    auto g = slice_of(full->hot.get(), rank, world, lo, hi);
    // the slice carries both permutations: grx_partitioned_run translates at its boundary
    g->hot_rank_of = full->hot_rank_of;
    g->hot_vertex_of.resize((std::size_t)full->n_rows);
    GRX_HIP_CHECK(hipMemcpy(g->hot_vertex_of.data(), full->hot_vertex_of.data(), (std::size_t)full->n_rows * 4,
                            hipMemcpyDeviceToDevice));
    GRX_HIP_CHECK(hipDeviceSynchronize());
    g->renumbered_slice = true;
    if (row_begin) *row_begin = lo;
    if (row_end) *row_end = hi;
    *out = g.release();
    return (int)GRX_OK;
  });
}

int grx_partitioned_expand(grx_context_t ctx, grx_graph_t local, const grx_options* opt,
                           int32_t edge_op, void* d_labels, int32_t iparam,
                           const int32_t* d_frontier, int64_t n_frontier, int32_t* d_scratch,
                           int64_t scratch_capacity, int32_t* d_sent_stamp, int64_t* d_send,
                           int64_t send_capacity, int64_t* n_found) {
  if (!ctx || !local || !d_labels || !d_scratch || !d_send || !d_sent_stamp || !n_found ||
      send_capacity < 2 ||
      scratch_capacity < 1 || n_frontier < 0 || (n_frontier && !d_frontier))
    return invalid("grx_partitioned_expand: bad arguments");
  if (edge_op != GRX_OP_BFS && edge_op != GRX_OP_SSSP)
    return unsupported("grx_partitioned_expand: edge_op must be GRX_OP_BFS or GRX_OP_SSSP");
  grx_options o;
  grx_default_options(&o);
  if (opt)
    o = *opt;
  return guarded([&] {
    if (edge_op == GRX_OP_BFS)
      return expand_as<int32_t>(ctx, local, o, edge_op, (int32_t*)d_labels, iparam, d_frontier,
                                n_frontier, d_scratch, scratch_capacity, d_sent_stamp, d_send,
                                send_capacity, n_found);
    return expand_as<float>(ctx, local, o, edge_op, (float*)d_labels, iparam, d_frontier,
                            n_frontier, d_scratch, scratch_capacity, d_sent_stamp, d_send,
                            send_capacity, n_found);
  });
}

int grx_partitioned_level_bitmap(grx_context_t ctx, const int32_t* d_depth, int64_t n_vertices,
                                 int32_t level, int64_t* d_words, int64_t word_capacity) {
  if (!ctx || !d_depth || !d_words || n_vertices < 1 || word_capacity < (n_vertices + 63) / 64)
    return invalid("grx_partitioned_level_bitmap: bad arguments");
  return guarded([&] {
    auto& sc = ctx->single();
    level_bitmap_kernel<<<bitmap_grid(n_vertices, sc.compute_units()), 256, 0, sc.stream()>>>(
        d_depth, n_vertices, level, reinterpret_cast<unsigned long long*>(d_words));
    GRX_HIP_CHECK(hipGetLastError());
    return (int)GRX_OK;
  });
}

namespace {
/// Enqueue the admission of one gather (either format) into next / *next_count.
void enqueue_admit(gcuda::standard_context_t& sc, int32_t edge_op, int32_t recv_format, void* d_labels,
                   int64_t n_vertices, int32_t* d_stamp, int32_t round, const int64_t* d_recv,
                   int32_t world, int64_t slot, int32_t me, int32_t lo, int32_t hi, int32_t* d_next,
                   int64_t next_capacity, unsigned long long* next_count,
                   unsigned long long* overflow) {
  if (recv_format == GRX_RECV_REPLICA_MIN) {
    const int64_t owned = std::max<int64_t>((int64_t)hi - lo, 1);
    const unsigned grid = (unsigned)std::min<int64_t>((owned + APPEND_TILE - 1) / APPEND_TILE,
                                                      (int64_t)sc.compute_units() * 8);
    if (edge_op == GRX_OP_BFS)
      admit_replica_kernel<int32_t><<<grid, 256, 0, sc.stream()>>>(
          (const int32_t*)d_labels, reinterpret_cast<const int32_t*>(d_recv), lo, hi, d_next,
          (unsigned long long)next_capacity, next_count, overflow);
    else
      admit_replica_kernel<float><<<grid, 256, 0, sc.stream()>>>(
          (const float*)d_labels, reinterpret_cast<const float*>(d_recv), lo, hi, d_next,
          (unsigned long long)next_capacity, next_count, overflow);
  } else if (recv_format == GRX_RECV_LEVEL_BITMAP) {
    const int64_t n_words = (n_vertices + 63) / 64;
    admit_bitmap_kernel<<<bitmap_grid(n_words, sc.compute_units()), 256, 0, sc.stream()>>>(
        (int32_t*)d_labels, n_vertices, round + 1,
        reinterpret_cast<const unsigned long long*>(d_recv), world, slot, lo, hi, d_next,
        (unsigned long long)next_capacity, next_count, overflow);
  } else {
    const int64_t total = (int64_t)world * (slot - 1);
    const unsigned grid = (unsigned)std::min<int64_t>(
        std::max<int64_t>((total + APPEND_TILE - 1) / APPEND_TILE, 1), (int64_t)sc.compute_units() * 8);
    if (edge_op == GRX_OP_BFS)
      admit_kernel<int32_t, false><<<grid, 256, 0, sc.stream()>>>(
          (int32_t*)d_labels, d_stamp, round, d_recv, world, slot, me, lo, hi, d_next,
          (unsigned long long)next_capacity, next_count, overflow);
    else
      admit_kernel<float, true><<<grid, 256, 0, sc.stream()>>>(
          (float*)d_labels, d_stamp, round, d_recv, world, slot, me, lo, hi, d_next,
          (unsigned long long)next_capacity, next_count, overflow);
  }
  GRX_HIP_CHECK(hipGetLastError());
}

int check_recv(int32_t edge_op, int32_t recv_format, int64_t n_vertices, int64_t slot) {
  if (recv_format == GRX_RECV_REPLICA_MIN)
    return GRX_OK;  // d_recv is the label snapshot, one entry per vertex
  if (recv_format != GRX_RECV_PAIRS && recv_format != GRX_RECV_LEVEL_BITMAP)
    return invalid("partitioned: unknown recv_format");
  if (recv_format == GRX_RECV_LEVEL_BITMAP) {
    if (edge_op != GRX_OP_BFS)
      return unsupported("partitioned: the level bitmap exchange carries no labels (BFS only)");
    if (slot < (n_vertices + 63) / 64)
      return invalid("partitioned: bitmap slot shorter than ceil(V / 64) words");
  } else if (slot < 2) {
    return invalid("partitioned: pair slot needs at least 2 words");
  }
  return GRX_OK;
}
}  // namespace

int grx_partitioned_admit(grx_context_t ctx, int32_t edge_op, void* d_labels, int64_t n_vertices,
                          int32_t* d_stamp, int32_t round, const int64_t* d_recv,
                          int32_t recv_format, int32_t world, int64_t slot, int32_t me, int32_t lo,
                          int32_t hi, int32_t* d_next, int64_t next_capacity, int64_t* n_next) {
  if (!ctx || !d_labels || !d_stamp || !d_recv || !d_next || !n_next || world < 1 || me < 0 ||
      me >= world || n_vertices < 1)
    return invalid("grx_partitioned_admit: bad arguments");
  if (edge_op != GRX_OP_BFS && edge_op != GRX_OP_SSSP)
    return unsupported("grx_partitioned_admit: edge_op must be GRX_OP_BFS or GRX_OP_SSSP");
  if (int rc = check_recv(edge_op, recv_format, n_vertices, slot))
    return rc;
  return guarded([&] {
    auto& sc = ctx->single();
    auto& ws = sc.workspace();
    unsigned long long* counters = ws.counters();  // zero between operators (see fetch_counters)
    enqueue_admit(sc, edge_op, recv_format, d_labels, n_vertices, d_stamp, round, d_recv, world, slot,
                  me, lo, hi, d_next, next_capacity, counters + hip::kernels::C_OUT,
                  counters + hip::kernels::C_OVERFLOW);
    unsigned long long* m = operators::advance::detail::fetch_counters(sc);
    error::throw_if_exception(m[hip::kernels::C_OVERFLOW] != 0,
                              "grx_partitioned_admit: next frontier capacity exceeded");
    *n_next = (int64_t)m[hip::kernels::C_OUT];
    return (int)GRX_OK;
  });
}

/* One fused superstep, ENQUEUE ONLY (no host wait): [admit the previous gather ->] advance over the
 * owned frontier (length read on the device) -> pack the finds into the send slot.  The host then
 * issues the collective on the same stream and synchronises once, on the gathered counts. */
int grx_partitioned_step(grx_context_t ctx, grx_graph_t local, const grx_options* opt,
                         int32_t edge_op, void* d_labels, int32_t* d_stamp, int32_t* d_sent,
                         int32_t round, const int64_t* d_recv, int32_t recv_format, int32_t world,
                         int64_t slot, int32_t me, int32_t lo, int32_t hi, int32_t* d_frontier,
                         int64_t frontier_capacity, uint64_t* d_frontier_count, int32_t* d_scratch,
                         int64_t scratch_capacity, int64_t* d_send, int64_t send_capacity,
                         void* d_snapshot) {
  if (!ctx || !local || !d_labels || !d_stamp || !d_sent || !d_frontier || !d_frontier_count ||
      !d_scratch || !d_send || send_capacity < 2 || frontier_capacity < 1 || scratch_capacity < 1 ||
      world < 1 || me < 0 || me >= world)
    return invalid("grx_partitioned_step: bad arguments");
  if (edge_op != GRX_OP_BFS && edge_op != GRX_OP_SSSP)
    return unsupported("grx_partitioned_step: edge_op must be GRX_OP_BFS or GRX_OP_SSSP");
  if (d_recv)
    if (int rc = check_recv(edge_op, recv_format, local->n_rows, slot))
      return rc;
  grx_options o;
  grx_default_options(&o);
  if (opt)
    o = *opt;
  return guarded([&] {
    auto& sc = ctx->single();
    auto& ws = sc.workspace();
    scoped_options scope(sc, &o);
    sc.options().holes_layout = false;
    unsigned long long* counters = ws.counters();
    unsigned long long* count_dev = reinterpret_cast<unsigned long long*>(d_frontier_count);
    // the previous step's hand-off has long landed (the host synchronised on the gather since):
    // check that nothing overflowed
    if (ctx->pending_sequence) {
      unsigned long long* m = operators::advance::detail::await_counters(sc, ctx->pending_sequence);
      ctx->pending_sequence = 0;
      error::throw_if_exception(m[hip::kernels::C_OVERFLOW] != 0,
                                "grx_partitioned_step: a buffer of the previous superstep overflowed");
    }
    // 1. admit what the other ranks found last superstep -> this superstep's owned frontier
    if (d_recv) {
      // *count_dev is zero: the previous step's hand-off cleared it.
      // The gather carries the finds of superstep round - 1
      enqueue_admit(sc, edge_op, recv_format, d_labels, local->n_rows, d_stamp, round - 1, d_recv,
                    world, slot, me, lo, hi, d_frontier, frontier_capacity, count_dev,
                    counters + hip::kernels::C_OVERFLOW);
    }
    // the labels of the owned range as they stand before this superstep's advance: what a
    // GRX_RECV_REPLICA_MIN admission of this superstep compares against
    if (d_snapshot && hi > lo)
      GRX_HIP_CHECK(hipMemcpyAsync((char*)d_snapshot + (std::size_t)lo * 4,
                                   (const char*)d_labels + (std::size_t)lo * 4,
                                   (std::size_t)(hi - lo) * 4, hipMemcpyDeviceToDevice, sc.stream()));
    // 2. local advance over the owned frontier (duplicate-free: work bounded by the rank's edges)
    graph_type G = local->view();
    const std::size_t bound = (std::size_t)std::min<int64_t>(frontier_capacity,
                                                              std::max<int32_t>(hi - lo, 1));
    if (edge_op == GRX_OP_BFS) {
      int* depth = reinterpret_cast<int*>(d_labels);
      const int next_level = round + 1;
      auto visit = [depth, next_level] __host__ __device__(vertex_t const& src, vertex_t const& dst,
                                                           edge_t const& e, weight_t const& w) -> bool {
        return next_level < math::atomic::min(&depth[dst], next_level);
      };
      // a wide superstep (the job found many vertices in the previous one): the single-GPU search's
      // wide-level form -- settled bitmap of the low ids in LDS, batched depth look-ups, the functor
      // on packed survivors (operators/settled.hxx); the frontier length stays on the device
      bool wide = false;
      if (sc.options().settled_filter && ctx->superstep_finds_hint >= (long long)sc.options().fused_min_slots) {
        auto has_depth = [depth] __device__(vertex_t const& v) -> bool {
          return depth[v] != std::numeric_limits<vertex_t>::max();
        };
        ctx->superstep_settled.rebuild((std::size_t)local->n_rows, has_depth, sc);
        wide = operators::advance::block_mapped::enqueue_packed_settled(
            G, operators::advance::with_settled(visit, ctx->superstep_settled.view(), has_depth), d_frontier,
            bound, count_dev, (unsigned long long)local->nnz, d_scratch, (std::size_t)scratch_capacity, sc);
      }
      if (!wide)
        operators::advance::block_mapped::enqueue_packed(G, visit, d_frontier, bound, count_dev,
                                                         (unsigned long long)local->nnz, d_scratch,
                                                         (std::size_t)scratch_capacity, sc);
    } else {
      float* dist = reinterpret_cast<float*>(d_labels);
      // one copy of an improved vertex per superstep (the first improver wins the exchange on
      // its `sent` stamp): the raw output is duplicate-free; later improvements still lower
      // dist[dst] and the pack reads the latest value
      int32_t* sent = d_sent;
      const int32_t rnd = round;
      auto relax = [dist, sent, rnd] __host__ __device__(vertex_t const& src, vertex_t const& dst,
                                                         edge_t const& e, weight_t const& w) -> bool {
        float through = thread::load(&dist[src]) + w;
        if (!(through < math::atomic::min(&dist[dst], through)))
          return false;
        return math::atomic::exch(&sent[dst], rnd) != rnd;
      };
      // wide supersteps once a quarter of the vertices have been found (the hubs are reached): the
      // single-GPU search's 2-byte bound image in front of the distances (operators::advance::
      // with_bounds; clients.hxx::sssp_enactor_t) -- bounds of the replica as it stands now
      bool wide = false;
      if (sc.options().settled_filter && ctx->superstep_finds_hint >= (long long)sc.options().fused_min_slots &&
          4 * ctx->superstep_found_so_far >= (long long)local->n_rows) {
        auto ordered = [] __host__ __device__(float x) -> unsigned {
          unsigned b;
          memcpy(&b, &x, 4);
          return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
        };
        if (ctx->superstep_bound16.size() < (std::size_t)local->n_rows)
          ctx->superstep_bound16.resize((std::size_t)local->n_rows);
        unsigned short* bound16 = ctx->superstep_bound16.data();
        const float* dist_now = dist;
        hip::for_each_index(
            (std::size_t)local->n_rows,
            [dist_now, bound16, ordered] __device__(std::size_t i) {
              const unsigned bits = ordered(dist_now[i]);
              const unsigned up = (bits >> 16) + ((bits & 0xffffu) ? 1u : 0u);
              bound16[i] = (unsigned short)(up > 0xffffu ? 0xffffu : up);
            },
            sc.stream());
        auto cached = [dist, ordered] __device__(vertex_t const& src, vertex_t const& dst, edge_t const& e,
                                                 weight_t const& w, unsigned short const& b) -> bool {
          const unsigned limit = (unsigned)b;
          return limit != 0xffffu && ordered(dist[src] + w) >= (limit << 16);
        };
        const unsigned short* bounds = bound16;
        auto not_shorter = [cached, bounds] __device__(vertex_t const& src, vertex_t const& dst, edge_t const& e,
                                                       weight_t const& w) -> bool {
          return cached(src, dst, e, w, bounds[dst]);
        };
        wide = operators::advance::block_mapped::enqueue_packed_settled(
            G, operators::advance::with_bounds<vertex_t>(relax, not_shorter, cached, bounds, (std::size_t)local->n_rows),
            d_frontier, bound, count_dev, (unsigned long long)local->nnz, d_scratch, (std::size_t)scratch_capacity,
            sc);
      }
      if (!wide)
        operators::advance::block_mapped::enqueue_packed(G, relax, d_frontier, bound, count_dev,
                                                         (unsigned long long)local->nnz, d_scratch,
                                                         (std::size_t)scratch_capacity, sc);
    }
    // 3. pack the finds (their number is counters[C_OUT], still on the device; C_SELECT is zero:
    //    every hand-off clears it)
    const unsigned pgrid = (unsigned)sc.compute_units() * 8;
    if (edge_op == GRX_OP_BFS)
      pack_pairs_kernel<int32_t><<<pgrid, 256, 0, sc.stream()>>>(
          d_scratch, 0, (int32_t*)d_labels, d_send, send_capacity, counters,
          counters + hip::kernels::C_OUT);
    else
      pack_pairs_kernel<float><<<pgrid, 256, 0, sc.stream()>>>(
          d_scratch, 0, (float*)d_labels, d_send, send_capacity, counters,
          counters + hip::kernels::C_OUT);
    GRX_HIP_CHECK(hipGetLastError());
    // 4. hand the counters over (overflow flag), write the pair count into the slot's header, clear
    //    the counters and the frontier length (the next admit accumulates into it) -- one launch,
    //    nobody waits here
    ctx->pending_sequence = operators::advance::detail::publish_counters(
        sc, reinterpret_cast<long long*>(d_send), hip::kernels::C_SELECT, count_dev);
    return (int)GRX_OK;
  });
}

}  // extern "C"
