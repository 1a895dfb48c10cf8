/**
 * @file advance_kernels.hxx
 * @brief Hand-written gfx950 kernels of the advance operator: neighbour-list
 * expansion of a frontier over CSR with an opaque per-edge functor.
 *
 * The contract is the reference's (framework/operators/advance/block_mapped.hxx:
 * 30-147, thread_mapped.hxx:25-96, merge_path.hxx:28-114): for every VALID input
 * slot and every out-edge e=(v,n) call op(v, n, e, w) exactly once; when an output
 * frontier is requested, emit n where op returned true.  Everything else --
 * schedules, LDS staging, output packing -- is designed for CDNA4:
 *
 *  - 256-thread workgroups = 4 wavefronts of 64 lanes; a lane reads
 *    column_indices[e] for CONSECUTIVE e across the wavefront (coalesced 256-B
 *    requests out of HBM), never a private list unless the schedule is
 *    thread_mapped;
 *  - workgroups are PERSISTENT (grid = a few per CU) and walk tiles / chunks /
 *    edge ranges with a grid stride, so per-workgroup state survives across work
 *    units;
 *  - the per-tile (vertex, first edge, scanned degree) triples live in LDS and
 *    the owner of edge i is found by binary search there;
 *  - accepted neighbours are packed with ballot + mbcnt into a PER-WAVEFRONT LDS
 *    queue whose fill level is a wave-uniform register: no LDS atomics and no
 *    workgroup barrier while expanding.  A wavefront claims output space with one
 *    global atomic per ~450 accepted neighbours, a workgroup with one more when it
 *    retires -- the single output cursor saturates near 90 atomics/us on MI355X,
 *    so it must see O(outputs/450 + workgroups) atomics, not O(tiles);
 *  - lists of >= hub_threshold edges are not expanded in place: they are cut into
 *    fixed chunks appended to a device queue (one reservation atomic per tile)
 *    that a second persistent kernel spreads over all 256 CUs (an RMAT-22 hub has
 *    3e5 edges);
 *  - while packing, the kernel also sums the degrees of the neighbours it emits:
 *    the next advance over that frontier then knows an upper bound of its work
 *    without the reference's extra reduction pass (advance/helpers.hxx:112-146);
 *  - "holes" layout (one output slot per traversed edge, invalid where the op
 *    said no) reproduces the reference's output exactly and is kept as a mode;
 *    the default "packed" layout writes only accepted neighbours.
 *
 * Counter slots of workspace_t::counters() used here: see counter_slot.
 */
#pragma once

#include <gunrock/framework/operators/configs.hxx>
#include <gunrock/framework/operators/settled.hxx>
#include <gunrock/hip/primitives.hxx>
#include <gunrock/util/type_limits.hxx>

namespace gunrock {
namespace hip {
namespace kernels {

using operators::advance_io_type_t;

constexpr int ADV_BLOCK = 256;                    // threads per workgroup
constexpr int ADV_WAVES = ADV_BLOCK / wave_size;  // 4
constexpr int ADV_WQCAP = 512;                    // entries of one wavefront's output queue
#ifndef GRX_ADV_UNROLL
#define GRX_ADV_UNROLL 4
#endif
constexpr int ADV_UNROLL = GRX_ADV_UNROLL;        // independent edges in flight per lane

enum counter_slot : int {
  C_OUT = 0,        ///< output cursor (elements)
  C_CHUNKS = 1,     ///< hub chunk queue cursor
  C_WORK = 2,       ///< degree sum / total work of the input frontier
  C_OVERFLOW = 3,   ///< set when an output write was dropped for lack of capacity
  // 4: free
  C_NEXT_WORK = 5,  ///< sum of degrees of the emitted neighbours (work of the next advance)
  C_SELECT = 6,     ///< packers (partitioned supersteps): number of selected elements
  C_BUCKET0 = 8,    ///< bucketing: small / medium queue cursors (8, 9)
  C_MAXDEG = 12,    ///< max degree reduction
  C_TILE_POOL = 16  ///< 16..23: dynamic tile cursors, one per pool of workgroups (work_stealing)
};

template <typename vertex_t, typename edge_t>
struct chunk_t {
  vertex_t source;
  int count;
  edge_t first;
};

// ---------------------------------------------------------------------------
// Per-wavefront output queue (LDS), fill level in a wave-uniform register.
// ---------------------------------------------------------------------------
template <typename vertex_t>
struct wave_queue_t {
  vertex_t* q;              // this wavefront's ADV_WQCAP entries in LDS
  unsigned fill;            // wave-uniform
  unsigned long long work;  // per lane: sum of degrees of the neighbours this lane emitted
  int cursor = C_OUT;       // counter slot that hands out positions of the destination list
  unsigned cap = ADV_WQCAP; // entries behind q

  __device__ __forceinline__ void flush(vertex_t* out, std::size_t capacity,
                                        unsigned long long* counters) {
    if (fill == 0)
      return;
    const int lane = lane_id();
    unsigned long long base = 0;
    if (lane == 0)
      base = atomicAdd(&counters[cursor], (unsigned long long)fill);
    base = __shfl(base, 0, wave_size);
    for (unsigned j = lane; j < fill; j += wave_size) {
      if (base + j < capacity)
        out[base + j] = q[j];
      else
        counters[C_OVERFLOW] = 1ull;
    }
    fill = 0;
  }

  /// flush() that also sums degree_of(entry) into `work`: the degrees of a whole queue are
  /// looked up here, 8 independent loads per lane, instead of one dependent lookup behind each
  /// accepted edge inside the expansion loop.
  template <typename degree_f>
  __device__ __forceinline__ void flush_summing(vertex_t* out, std::size_t capacity,
                                                unsigned long long* counters, degree_f degree_of) {
    if (fill == 0)
      return;
    const int lane = lane_id();
    unsigned long long base = 0;
    if (lane == 0)
      base = atomicAdd(&counters[cursor], (unsigned long long)fill);
    base = __shfl(base, 0, wave_size);
    for (unsigned j = lane; j < fill; j += wave_size) {
      const vertex_t x = q[j];
      work += degree_of(x);
      if (base + j < capacity)
        out[base + j] = x;
      else
        counters[C_OVERFLOW] = 1ull;
    }
    fill = 0;
  }

  /// push() without a degree: pair with flush_summing / drain_block_summing.
  template <typename degree_f>
  __device__ __forceinline__ void push_deferred(bool keep, vertex_t value, vertex_t* out,
                                                std::size_t capacity, unsigned long long* counters,
                                                degree_f degree_of) {
    unsigned long long m = __ballot(keep);
    if (m == 0)
      return;
    if (fill + wave_size > cap)
      flush_summing(out, capacity, counters, degree_of);
    if (keep)
      q[fill + rank_in_mask(m)] = value;
    fill += (unsigned)__popcll(m);
  }

  /// All 64 lanes must call (keep=false for idle lanes).
  __device__ __forceinline__ void push(bool keep, vertex_t value, unsigned degree, vertex_t* out,
                                       std::size_t capacity, unsigned long long* counters) {
    unsigned long long m = __ballot(keep);
    if (m == 0)
      return;
    if (fill + wave_size > cap)
      flush(out, capacity, counters);
    if (keep) {
      q[fill + rank_in_mask(m)] = value;
      work += degree;
    }
    fill += (unsigned)__popcll(m);
  }
};

/// Retire a workgroup: drain its four wavefront queues with ONE cursor atomic and
/// publish the degree sum of everything it emitted.  Contains barriers.
template <int WAVES = ADV_WAVES, typename vertex_t>
__device__ __forceinline__ void drain_block(wave_queue_t<vertex_t>& wq, unsigned* s_counts,
                                            unsigned long long* s_base, vertex_t* out,
                                            std::size_t capacity, unsigned long long* counters) {
  const int lane = lane_id();
  const int wave = threadIdx.x / wave_size;
  const unsigned long long wave_work = wave_sum(wq.work);
  if (lane == 0) {
    s_counts[wave] = wq.fill;
    if (wave_work && wq.cursor == C_OUT)
      atomicAdd(&counters[C_NEXT_WORK], wave_work);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned total = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w)
      total += s_counts[w];
    *s_base = total ? atomicAdd(&counters[wq.cursor], (unsigned long long)total) : 0ull;
  }
  __syncthreads();
  unsigned long long base = *s_base;
  for (int w = 0; w < wave; ++w)
    base += s_counts[w];
  for (unsigned j = lane; j < wq.fill; j += wave_size) {
    if (base + j < capacity)
      out[base + j] = wq.q[j];
    else
      counters[C_OVERFLOW] = 1ull;
  }
  wq.fill = 0;
  wq.work = 0;
  __syncthreads();
}

/// drain_block for queues filled with push_deferred: the degrees of the leftover entries are
/// looked up first.
template <int WAVES = ADV_WAVES, typename vertex_t, typename degree_f>
__device__ __forceinline__ void drain_block_summing(wave_queue_t<vertex_t>& wq, unsigned* s_counts,
                                                    unsigned long long* s_base, vertex_t* out,
                                                    std::size_t capacity,
                                                    unsigned long long* counters,
                                                    degree_f degree_of) {
  for (unsigned j = lane_id(); j < wq.fill; j += wave_size)
    wq.work += degree_of(wq.q[j]);
  drain_block<WAVES>(wq, s_counts, s_base, out, capacity, counters);
}

// ---------------------------------------------------------------------------
// Sum / max of degrees of the valid input slots.
// Restates advance/helpers.hxx:112-146 (compute_output_length) as one kernel with
// a 64-bit result (the reference sums in edge_t, SURVEY.md 8a' q8).  One atomic
// per workgroup; launch with a few hundred workgroups.
// ---------------------------------------------------------------------------
template <advance_io_type_t IN, typename graph_t, typename vertex_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    degree_sum_kernel(graph_t G, const vertex_t* input, std::size_t n_in,
                      unsigned long long* counters) {
  __shared__ unsigned long long s_part[ADV_WAVES];
  unsigned long long local = 0;
  for (std::size_t i = blockIdx.x * (std::size_t)ADV_BLOCK + threadIdx.x; i < n_in;
       i += (std::size_t)gridDim.x * ADV_BLOCK) {
    vertex_t v = (IN == advance_io_type_t::graph) ? (vertex_t)i : input[i];
    if (util::limits::is_valid(v))
      local += (unsigned long long)G.get_number_of_neighbors(v);
  }
  local = wave_sum(local);
  if (lane_id() == 0)
    s_part[threadIdx.x / wave_size] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < ADV_WAVES; ++w)
      t += s_part[w];
    if (t)
      atomicAdd(&counters[C_WORK], t);
  }
}

template <typename graph_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    max_degree_kernel(graph_t G, unsigned long long* counters) {
  using vertex_t = typename graph_t::vertex_type;
  __shared__ unsigned long long s_part[ADV_WAVES];
  unsigned long long local = 0;
  const std::size_t n = (std::size_t)G.get_number_of_vertices();
  for (std::size_t i = blockIdx.x * (std::size_t)ADV_BLOCK + threadIdx.x; i < n;
       i += (std::size_t)gridDim.x * ADV_BLOCK) {
    unsigned long long d = (unsigned long long)G.get_number_of_neighbors((vertex_t)i);
    local = d > local ? d : local;
  }
  local = wave_max(local);
  if (lane_id() == 0)
    s_part[threadIdx.x / wave_size] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < ADV_WAVES; ++w)
      t = s_part[w] > t ? s_part[w] : t;
    if (t)
      atomicMax(&counters[C_MAXDEG], t);
  }
}

// ---------------------------------------------------------------------------
// block_mapped: persistent workgroups over tiles of ADV_BLOCK input slots.
// ---------------------------------------------------------------------------
template <bool HOLES,
          bool DYNAMIC,
          advance_io_type_t IN,
          advance_io_type_t OUT,
          typename graph_t,
          typename op_t,
          typename vertex_t,
          typename edge_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    block_mapped_kernel(graph_t G,
                        op_t op,
                        const vertex_t* __restrict__ input,
                        std::size_t n_in,
                        vertex_t* __restrict__ output,
                        std::size_t capacity,
                        unsigned long long* counters,
                        chunk_t<vertex_t, edge_t>* chunks,
                        unsigned long long chunk_capacity,
                        unsigned hub_threshold,
                        unsigned chunk_edges,
                        const unsigned long long* n_in_device = nullptr,
                        unsigned tile_width = ADV_BLOCK) {
  using weight_t = typename graph_t::weight_type;
  constexpr bool HAS_OUT = (OUT != advance_io_type_t::none);
  constexpr bool PACKED = HAS_OUT && !HOLES;
  // the frontier length may live on the device (a producer kernel earlier in the stream wrote
  // it): no host round trip between producing and consuming a frontier
  if (n_in_device)
    n_in = (std::size_t)__hip_atomic_load(n_in_device, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  __shared__ vertex_t s_vertex[ADV_BLOCK];
  __shared__ edge_t s_first[ADV_BLOCK];
  __shared__ unsigned s_scan[ADV_BLOCK];
  __shared__ unsigned long long s_wave_totals[ADV_WAVES + 1];
  __shared__ unsigned s_counts[ADV_WAVES];
  __shared__ unsigned long long s_base;
  __shared__ unsigned long long s_tile;
  __shared__ vertex_t s_queue[PACKED ? ADV_WAVES * ADV_WQCAP : 1];

  const int tid = threadIdx.x;
  const int wave = tid / wave_size;
  wave_queue_t<vertex_t> wq{s_queue + (PACKED ? wave * ADV_WQCAP : 0), 0u, 0ull};
  auto degree_of = [&G](vertex_t x) -> unsigned { return (unsigned)G.get_number_of_neighbors(x); };

  // a tile = tile_width (<= ADV_BLOCK) consecutive input slots, one per lane; narrow tiles bound
  // the work of one tile (tile_width * hub_threshold edges) when there are few of them
  const unsigned long long n_tiles = (n_in + tile_width - 1) / tile_width;

  // Tiles are claimed two ahead and their operands fetched one tile early: while this workgroup
  // expands tile t, the row offsets of tile t+1 and the input slots of tile t+2 are in flight.
  // Under load a dependent global access costs a full trip through queues that the expansion's
  // gathers keep deep (~10 us on RMAT-22 level 2), and a tile needs two of them before its first
  // edge: fetched in line they were most of the kernel's time.
  // Dynamic claims (work_stealing): workgroups form up to 8 pools (blockIdx % 8 -- workgroups are
  // dealt round-robin to the 8 XCDs), pool p owns tiles p, p + 8, ... and hands them out through
  // its own cursor: eight cursor addresses instead of one (a single address retires ~90 atomics/us),
  // and the cursor value a claim returns was REQUESTED during the previous claim, a whole tile
  // earlier, so its round trip is off the critical path.
  const unsigned pools = gridDim.x < 8u ? gridDim.x : 8u;
  const unsigned pool = blockIdx.x % pools;
  unsigned long long pending = 0;  // thread 0: cursor value requested but not yet handed out
  if (DYNAMIC && tid == 0)
    pending = atomicAdd(&counters[C_TILE_POOL + pool], 1ull);
  auto claim = [&](unsigned long long after) -> unsigned long long {
    if (DYNAMIC) {
      if (tid == 0) {
        s_tile = pending * pools + pool;
        pending = atomicAdd(&counters[C_TILE_POOL + pool], 1ull);
      }
      __syncthreads();
      const unsigned long long t = s_tile;
      __syncthreads();  // s_tile is rewritten by the next claim
      return t;
    }
    return after + gridDim.x;
  };
  auto slot_vertex = [&](unsigned long long t) -> vertex_t {
    const std::size_t idx = (std::size_t)t * tile_width + tid;
    if (t < n_tiles && (unsigned)tid < tile_width && idx < n_in)
      return (IN == advance_io_type_t::graph) ? (vertex_t)idx : input[idx];
    return gunrock::numeric_limits<vertex_t>::invalid();
  };

#ifdef GRX_TILE_TIMING
  unsigned long long tt_start = wall_clock64(), tt_stage = 0, tt_edges = 0, tt_iters = 0, tt_tiles = 0;
#endif
  unsigned long long tile = DYNAMIC ? claim(0) : (unsigned long long)blockIdx.x;
  unsigned long long tile1 = claim(tile);
  vertex_t v = slot_vertex(tile);
  edge_t first = 0, last = 0;
  if (util::limits::is_valid(v)) {
    first = G.get_starting_edge(v);
    last = G.get_starting_edge(v + 1);
  }
  vertex_t v1 = slot_vertex(tile1);

  while (tile < n_tiles) {
    // ---- 1. this tile's operands arrived during the previous one; start the next ones ------
#ifdef GRX_TILE_TIMING
    const unsigned long long tt0 = wall_clock64();
#endif
    const unsigned deg = (unsigned)(last - first);
    const unsigned long long tile2 = claim(tile1);
    edge_t first1 = 0, last1 = 0;
    if (util::limits::is_valid(v1)) {
      first1 = G.get_starting_edge(v1);
      last1 = G.get_starting_edge(v1 + 1);
    }
    const vertex_t v2 = slot_vertex(tile2);
    // ---- 2. hubs leave the tile as equal chunks; one reservation per tile ----
    //         (degree, chunk count) are scanned together, packed in 64 bits
    unsigned my_chunks = 0;
    if (!HOLES && deg >= hub_threshold)
      my_chunks = (deg + chunk_edges - 1) / chunk_edges;
    unsigned long long packed = ((unsigned long long)my_chunks << 32) | (my_chunks ? 0u : deg);
    unsigned long long packed_total;
    unsigned long long packed_excl =
        block_exclusive_sum<ADV_BLOCK>(packed, packed_total, s_wave_totals);
    unsigned total = (unsigned)packed_total;
    unsigned excl = (unsigned)packed_excl;
    if (!HOLES) {
      const unsigned tile_chunks = (unsigned)(packed_total >> 32);
      if (tile_chunks) {  // workgroup-uniform
        if (tid == 0)
          s_base = atomicAdd(&counters[C_CHUNKS], (unsigned long long)tile_chunks);
        __syncthreads();
        const unsigned long long at = s_base + (packed_excl >> 32);
        const bool fits = s_base + tile_chunks <= chunk_capacity;
        if (my_chunks) {
          for (unsigned c = 0; c < my_chunks && at + c < chunk_capacity; ++c) {
            const unsigned off = c * chunk_edges;
            chunk_t<vertex_t, edge_t> d;
            d.source = v;
            d.first = first + (edge_t)off;
            d.count = fits ? (int)((deg - off < chunk_edges) ? deg - off : chunk_edges) : 0;
            chunks[at + c] = d;
          }
        }
        __syncthreads();  // s_base is reused below
        if (!fits) {
          // the queue is full: this tile expands its hubs in place (slow, correct)
          unsigned t2;
          unsigned long long p2 = block_exclusive_sum<ADV_BLOCK>((unsigned long long)deg,
                                                                 packed_total, s_wave_totals);
          t2 = (unsigned)packed_total;
          total = t2;
          excl = (unsigned)p2;
        }
      }
    }
    s_vertex[tid] = v;
    s_first[tid] = first;
    s_scan[tid] = excl;
    if (HOLES && HAS_OUT && tid == 0)
      s_base = total ? atomicAdd(&counters[C_OUT], (unsigned long long)total) : 0ull;
    __syncthreads();

#ifdef GRX_TILE_TIMING
    const unsigned long long tt1 = wall_clock64();
    tt_stage += tt1 - tt0;
    tt_iters += (total + ADV_BLOCK * ADV_UNROLL - 1) / (ADV_BLOCK * ADV_UNROLL);
    ++tt_tiles;
#endif
    // ---- 3. stride the concatenated neighbour lists -------------------------
    for (unsigned i0 = 0; i0 < total; i0 += ADV_BLOCK * ADV_UNROLL) {
      vertex_t src[ADV_UNROLL], nbr[ADV_UNROLL];
      edge_t eid[ADV_UNROLL];
      weight_t wgt[ADV_UNROLL];
      bool live[ADV_UNROLL];
#pragma unroll
      for (int k = 0; k < ADV_UNROLL; ++k) {
        const unsigned i = i0 + k * ADV_BLOCK + tid;
        live[k] = i < total;
        if (live[k]) {
          const int slot = rightmost_le(s_scan, i, ADV_BLOCK);
          src[k] = s_vertex[slot];
          eid[k] = s_first[slot] + (edge_t)(i - s_scan[slot]);
          nbr[k] = G.get_destination_vertex(eid[k]);
          wgt[k] = G.get_edge_weight(eid[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < ADV_UNROLL; ++k) {
        bool keep = false;
        if (live[k])
          keep = op(src[k], nbr[k], eid[k], wgt[k]);
        if constexpr (HAS_OUT) {
          if constexpr (HOLES) {
            if (live[k]) {
              const unsigned long long at = s_base + i0 + k * ADV_BLOCK + tid;
              if (at < capacity)
                output[at] = keep ? nbr[k] : gunrock::numeric_limits<vertex_t>::invalid();
              else
                counters[C_OVERFLOW] = 1ull;
            }
          } else {
            wq.push_deferred(keep, nbr[k], output, capacity, counters, degree_of);
          }
        }
      }
    }
    __syncthreads();  // the LDS tile arrays are rewritten by the next tile
#ifdef GRX_TILE_TIMING
    tt_edges += wall_clock64() - tt1;
#endif

    tile = tile1;
    tile1 = tile2;
    v = v1;
    first = first1;
    last = last1;
    v1 = v2;
  }
#ifdef GRX_TILE_TIMING
  const unsigned long long tt_loop_end = wall_clock64();
#endif
  if constexpr (PACKED)
    drain_block_summing(wq, s_counts, &s_base, output, capacity, counters, degree_of);
#ifdef GRX_TILE_TIMING
  if (tid == 0) {  // 100 MHz ticks, summed over workgroups (slots 24..30), max total in slot 7
    const unsigned long long tt_end = wall_clock64();
    atomicAdd(&counters[24], tt_stage);
    atomicAdd(&counters[25], tt_edges);
    atomicAdd(&counters[26], tt_iters);
    atomicAdd(&counters[27], tt_tiles);
    atomicAdd(&counters[28], tt_end - tt_start);
    atomicAdd(&counters[29], tt_end - tt_loop_end);
    atomicAdd(&counters[30], 1ull);
    atomicMax(&counters[7], tt_end - tt_start);
  }
#endif
}

// ---------------------------------------------------------------------------
// block_mapped, FUSED form for wide frontiers: classify_hubs_kernel + expand_fused_kernel.
//
// Why: the two-kernel form above (tiles, then hub chunks) leaves the machine half idle while the
// tile kernel's stragglers finish -- rocprofv3 PMC on RMAT-22 BFS level 2: 96 G L2 requests/s in
// the tile kernel against 144 G/s in the chunk kernel, mean workgroup busy 56 % of the kernel --
// and the chunk kernel cannot start before the last tile has queued its hubs.  Here a light
// pre-pass queues every hub list as chunks and marks the slot in a bit mask (one lane per slot,
// ~20 us for 2 M slots); then ONE persistent kernel expands its share of the tiles (hub slots
// skipped) and, as soon as a workgroup has no tile left, claims chunks from eight cursors (one per
// pool of workgroups dealt to an XCD, the claim for the next chunk requested while the current
// one is expanded) until the queue is empty: early finishers eat the chunk queue while the
// heaviest tiles are still running.  Exactly-once is kept by construction: a hub list is reached
// through its chunks only, every other list through its tile only.
// ---------------------------------------------------------------------------
constexpr int CLASSIFY_SLOTS = 8;                          // input slots per thread and super-tile
constexpr int CLASSIFY_TILE = ADV_BLOCK * CLASSIFY_SLOTS;  // 2048 slots per workgroup step
constexpr int CLAIM_LINE = 16;   // 64-bit words between two claim cursors (one 128-B line each)
constexpr int CLAIM_BATCH = 4;   // chunks handed out per claim

template <advance_io_type_t IN, typename graph_t, typename vertex_t, typename edge_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    classify_hubs_kernel(graph_t G,
                         const vertex_t* __restrict__ input,
                         std::size_t n_in,
                         const unsigned long long* n_in_device,
                         chunk_t<vertex_t, edge_t>* chunks,
                         unsigned long long chunk_capacity,
                         unsigned hub_threshold,
                         unsigned chunk_edges,
                         unsigned long long* __restrict__ hub_mask,
                         unsigned long long* __restrict__ claim_cursors,
                         unsigned long long* counters) {
  __shared__ unsigned s_wave_totals[ADV_WAVES + 1];
  __shared__ unsigned long long s_base;
  if (n_in_device)
    n_in = (std::size_t)__hip_atomic_load(n_in_device, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int tid = threadIdx.x;
  const int lane = lane_id();
  if (blockIdx.x == 0 && tid < 8)  // the expansion kernel's claim cursors start from zero
    claim_cursors[tid * CLAIM_LINE] = 0ull;
  // a workgroup takes 2048 slots at a time, eight per thread (eight independent row lookups in
  // flight per lane), and reserves queue space for all their hubs with ONE atomic.  The slots of
  // a wavefront are 64 consecutive ones (one mask word, coalesced loads); consecutive GROUPS of 64
  // go to different workgroups: an ascending frontier of the hot-first copy starts with all its
  // hubs, and 2048 consecutive slots would leave their descriptors to one workgroup (42 us of
  // SSSP's second wide iteration against 8 us for as many slots without hubs)
  const std::size_t n_groups = (n_in + wave_size - 1) / wave_size;
  const std::size_t groups_per_step = (std::size_t)gridDim.x * (CLASSIFY_TILE / wave_size);
  const std::size_t wave_in_block = (std::size_t)(tid / wave_size);
  for (std::size_t step = 0; step < n_groups; step += groups_per_step) {
    vertex_t v[CLASSIFY_SLOTS];
    edge_t first[CLASSIFY_SLOTS];
    unsigned deg[CLASSIFY_SLOTS];
    unsigned mine = 0;
    auto slot_of = [&](int k) -> std::size_t {
      return (step + ((std::size_t)k * ADV_WAVES + wave_in_block) * gridDim.x + blockIdx.x) * wave_size +
             (std::size_t)lane;
    };
#pragma unroll
    for (int k = 0; k < CLASSIFY_SLOTS; ++k) {
      const std::size_t idx = slot_of(k);
      v[k] = gunrock::numeric_limits<vertex_t>::invalid();
      if (idx < n_in)
        v[k] = (IN == advance_io_type_t::graph) ? (vertex_t)idx : input[idx];
    }
#pragma unroll
    for (int k = 0; k < CLASSIFY_SLOTS; ++k) {
      first[k] = 0;
      deg[k] = 0;
      if (util::limits::is_valid(v[k])) {
        first[k] = G.get_starting_edge(v[k]);
        deg[k] = (unsigned)(G.get_starting_edge(v[k] + 1) - first[k]);
      }
      if (deg[k] >= hub_threshold)
        mine += (deg[k] + chunk_edges - 1) / chunk_edges;
    }
    unsigned total = 0;
    const unsigned excl = block_exclusive_sum<ADV_BLOCK>(mine, total, s_wave_totals);
    if (total) {  // workgroup-uniform
      if (tid == 0)
        s_base = atomicAdd(&counters[C_CHUNKS], (unsigned long long)total);
      __syncthreads();
    }
    unsigned long long at = total ? s_base + excl : 0ull;
#pragma unroll
    for (int k = 0; k < CLASSIFY_SLOTS; ++k) {
      const std::size_t idx = slot_of(k);
      const unsigned my_chunks = (deg[k] >= hub_threshold) ? (deg[k] + chunk_edges - 1) / chunk_edges : 0u;
      // a list is queued only when ALL its chunks fit; one that does not stays in its tile (slow,
      // correct) and the slots it reserved below the capacity become empty chunks
      const bool queued = my_chunks && at + my_chunks <= chunk_capacity;
      // a list of a few chunks is written by its lane; a longer one by the whole wavefront, chunk
      // c by lane c % 64 (the 320 K edges of R-MAT-22's largest hub are 313 descriptors: written by
      // one lane they were 40 of the 46-61 us this pass took on the level that holds the hubs)
      constexpr unsigned LANE_CHUNKS = 4;
      if (my_chunks <= LANE_CHUNKS)
        for (unsigned c = 0; c < my_chunks && at + c < chunk_capacity; ++c) {
          const unsigned off = c * chunk_edges;
          chunk_t<vertex_t, edge_t> d;
          d.source = v[k];
          d.first = first[k] + (edge_t)off;
          d.count = queued ? (int)((deg[k] - off < chunk_edges) ? deg[k] - off : chunk_edges) : 0;
          chunks[at + c] = d;
        }
      for (unsigned long long longs = __ballot(my_chunks > LANE_CHUNKS); longs; longs &= longs - 1) {
        const int owner = __ffsll((long long)longs) - 1;
        const vertex_t hub = __shfl(v[k], owner);
        const edge_t hub_first = __shfl(first[k], owner);
        const unsigned hub_degree = __shfl(deg[k], owner);
        const unsigned hub_chunks = __shfl(my_chunks, owner);
        const bool hub_queued = __shfl((int)queued, owner) != 0;
        const unsigned long long hub_at =
            ((unsigned long long)(unsigned)__shfl((int)(unsigned)(at >> 32), owner) << 32) |
            (unsigned)__shfl((int)(unsigned)at, owner);
        for (unsigned c = (unsigned)lane; c < hub_chunks && hub_at + c < chunk_capacity; c += wave_size) {
          const unsigned off = c * chunk_edges;
          chunk_t<vertex_t, edge_t> d;
          d.source = hub;
          d.first = hub_first + (edge_t)off;
          d.count = hub_queued ? (int)((hub_degree - off < chunk_edges) ? hub_degree - off : chunk_edges) : 0;
          chunks[hub_at + c] = d;
        }
      }
      at += my_chunks;
      // 64 consecutive, 64-aligned slots per wavefront and k: one mask word
      const unsigned long long word = __ballot(queued);
      if (lane == 0 && idx < (n_in + wave_size - 1) / wave_size * wave_size)
        hub_mask[idx / wave_size] = word;
    }
    if (total)
      __syncthreads();  // s_base is rewritten by the next super-tile
  }
}

template <advance_io_type_t IN,
          advance_io_type_t OUT,
          typename graph_t,
          typename op_t,
          typename vertex_t,
          typename edge_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    expand_fused_kernel(graph_t G,
                        op_t op,
                        const vertex_t* __restrict__ input,
                        std::size_t n_in,
                        const unsigned long long* n_in_device,
                        vertex_t* __restrict__ output,
                        std::size_t capacity,
                        unsigned long long* counters,
                        const chunk_t<vertex_t, edge_t>* __restrict__ chunks,
                        unsigned long long chunk_capacity,
                        const unsigned long long* __restrict__ hub_mask,
                        unsigned long long* __restrict__ claim_cursors,
                        int dealt = 0) {
  using weight_t = typename graph_t::weight_type;
  constexpr bool HAS_OUT = (OUT != advance_io_type_t::none);
  if (n_in_device)
    n_in = (std::size_t)__hip_atomic_load(n_in_device, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  __shared__ vertex_t s_vertex[ADV_BLOCK];
  __shared__ edge_t s_first[ADV_BLOCK];
  __shared__ unsigned s_scan[ADV_BLOCK];
  __shared__ unsigned s_wave_totals[ADV_WAVES + 1];
  __shared__ unsigned s_counts[ADV_WAVES];
  __shared__ unsigned long long s_base;
  __shared__ unsigned long long s_claim;
  __shared__ vertex_t s_queue[HAS_OUT ? ADV_WAVES * ADV_WQCAP : 1];

  const int tid = threadIdx.x;
  const int wave = tid / wave_size;
  wave_queue_t<vertex_t> wq{s_queue + (HAS_OUT ? wave * ADV_WQCAP : 0), 0u, 0ull};
  auto degree_of = [&G](vertex_t x) -> unsigned { return (unsigned)G.get_number_of_neighbors(x); };

  // ---- phase 1: this workgroup's tiles (static stride), hub slots skipped -------------------------
  const unsigned long long n_tiles = (n_in + ADV_BLOCK - 1) / ADV_BLOCK;
  auto fetch_slot = [&](unsigned long long t, vertex_t& v, edge_t& first, edge_t& last) {
    v = gunrock::numeric_limits<vertex_t>::invalid();
    first = last = 0;
    // an ASCENDING frontier of the hot-first copy is falling degrees: 256 consecutive slots would
    // make the first tiles 50 times the work of the last (one workgroup walking 64 K edges while the
    // others idle: 106 us for SSSP's 77 K-slot iteration).  Dealt, a tile is 16 groups of 16
    // consecutive slots, group g of tile t being group g * tiles + t of the frontier: the hottest 256
    // slots go to 16 tiles, and 16 neighbouring ids still share their row-offset lines (single slots
    // dealt: 47 us there, but 19 -> 41 us on a BFS level of 185 K slots that had no skew to fix)
    constexpr unsigned DEAL = 16;
    const std::size_t idx = dealt ? ((std::size_t)(tid / DEAL) * n_tiles + t) * DEAL + tid % DEAL
                                  : (std::size_t)t * ADV_BLOCK + tid;
    if (t < n_tiles && idx < n_in) {
      const vertex_t x = (IN == advance_io_type_t::graph) ? (vertex_t)idx : input[idx];
      const bool hub = (hub_mask[idx / wave_size] >> (idx % wave_size)) & 1ull;
      if (util::limits::is_valid(x) && !hub) {
        v = x;
        first = G.get_starting_edge(x);
        last = G.get_starting_edge(x + 1);
      }
    }
  };
  // operands of the next tile are fetched while this one expands (two dependent global accesses
  // per tile otherwise sit in front of its first edge)
  unsigned long long tile = blockIdx.x;
  vertex_t v, v1;
  edge_t first, last, first1, last1;
  fetch_slot(tile, v, first, last);
  while (tile < n_tiles) {
    fetch_slot(tile + gridDim.x, v1, first1, last1);
    const unsigned deg = (unsigned)(last - first);
    unsigned total;
    const unsigned excl = block_exclusive_sum<ADV_BLOCK>(deg, total, s_wave_totals);
    s_vertex[tid] = v;
    s_first[tid] = first;
    s_scan[tid] = excl;
    __syncthreads();
    for (unsigned i0 = 0; i0 < total; i0 += ADV_BLOCK * ADV_UNROLL) {
      vertex_t src[ADV_UNROLL], nbr[ADV_UNROLL];
      edge_t eid[ADV_UNROLL];
      weight_t wgt[ADV_UNROLL];
      bool live[ADV_UNROLL];
#pragma unroll
      for (int k = 0; k < ADV_UNROLL; ++k) {
        const unsigned i = i0 + k * ADV_BLOCK + tid;
        live[k] = i < total;
        if (live[k]) {
          const int slot = rightmost_le(s_scan, i, ADV_BLOCK);
          src[k] = s_vertex[slot];
          eid[k] = s_first[slot] + (edge_t)(i - s_scan[slot]);
          nbr[k] = G.get_destination_vertex(eid[k]);
          wgt[k] = G.get_edge_weight(eid[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < ADV_UNROLL; ++k) {
        bool keep = false;
        if (live[k])
          keep = op(src[k], nbr[k], eid[k], wgt[k]);
        if constexpr (HAS_OUT)
          wq.push_deferred(keep, nbr[k], output, capacity, counters, degree_of);
      }
    }
    __syncthreads();  // the LDS tile arrays are rewritten by the next tile
    tile += gridDim.x;
    v = v1;
    first = first1;
    last = last1;
  }

  // ---- phase 2: hub chunks, claimed dynamically ---------------------------------------------------
  unsigned long long n_chunks =
      __hip_atomic_load(&counters[C_CHUNKS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (n_chunks > chunk_capacity)
    n_chunks = chunk_capacity;
  if (n_chunks) {  // grid-uniform
    // Chunks are handed out in batches of CLAIM_BATCH through eight cursors, each on a cache line
    // of its own (a single line retires ~90 atomics/us, and cursors sharing one line share that
    // rate): pool p owns batches p, p + P, p + 2P, ...; workgroups of one XCD (blockIdx % 8) start
    // on one pool and move on to the next when theirs is exhausted.
    const unsigned long long n_batches = (n_chunks + CLAIM_BATCH - 1) / CLAIM_BATCH;
    const unsigned pools = gridDim.x < 8u ? gridDim.x : 8u;
    unsigned pool = blockIdx.x % pools;
    unsigned tried = 0;              // thread 0: pools found empty so far
    unsigned long long pending = 0;  // thread 0: cursor value requested, not yet handed out
    if (tid == 0)
      pending = atomicAdd(&claim_cursors[pool * CLAIM_LINE], 1ull);
    for (;;) {
      if (tid == 0) {
        unsigned long long b = pending * pools + pool;
        while (b >= n_batches && ++tried < pools) {  // own pool exhausted: help the next one
          pool = (pool + 1) % pools;
          b = atomicAdd(&claim_cursors[pool * CLAIM_LINE], 1ull) * pools + pool;
        }
        s_claim = b;
        if (b < n_batches)  // request the NEXT batch now; its round trip hides behind this one
          pending = atomicAdd(&claim_cursors[pool * CLAIM_LINE], 1ull);
      }
      __syncthreads();
      const unsigned long long batch = s_claim;
      __syncthreads();  // s_claim is rewritten by the next claim
      if (batch >= n_batches)
        break;
      const unsigned long long c0 = batch * CLAIM_BATCH;
      const unsigned long long c1 = c0 + CLAIM_BATCH < n_chunks ? c0 + CLAIM_BATCH : n_chunks;
      for (unsigned long long c = c0; c < c1; ++c) {
        const chunk_t<vertex_t, edge_t> d = chunks[c];
        vertex_t source = d.source;
        for (int j0 = 0; j0 < d.count; j0 += ADV_BLOCK * ADV_UNROLL) {
          vertex_t nbr[ADV_UNROLL];
          edge_t eid[ADV_UNROLL];
          weight_t wgt[ADV_UNROLL];
          bool live[ADV_UNROLL];
#pragma unroll
          for (int k = 0; k < ADV_UNROLL; ++k) {
            const int j = j0 + k * ADV_BLOCK + tid;
            live[k] = j < d.count;
            if (live[k]) {
              eid[k] = d.first + (edge_t)j;
              nbr[k] = G.get_destination_vertex(eid[k]);
              wgt[k] = G.get_edge_weight(eid[k]);
            }
          }
#pragma unroll
          for (int k = 0; k < ADV_UNROLL; ++k) {
            bool keep = false;
            if (live[k])
              keep = op(source, nbr[k], eid[k], wgt[k]);
            if constexpr (HAS_OUT)
              wq.push_deferred(keep, nbr[k], output, capacity, counters, degree_of);
          }
        }
      }
    }
  }
  if constexpr (HAS_OUT)
    drain_block_summing(wq, s_counts, &s_base, output, capacity, counters, degree_of);
}

// ---------------------------------------------------------------------------
// Wide levels of a client that named its settled destinations (operators/settled.hxx): the work
// split of expand_fused_kernel (tiles with hub slots masked out, then hub chunks claimed
// dynamically) with
//  - ONE 1024-thread workgroup per CU whose dynamic LDS holds the settled bitmap (96 KB = 768 K
//    ids): an edge whose destination has its bit set costs no memory request at all;
//  - the client's pure predicate evaluated for the other edges, four independent loads in flight
//    per lane, while the column loads of the next four are already under way (inside the functor the same test is a dependent load in front of an atomic, and the
//    four calls of a lane are serialised);
//  - the surviving edges packed per wavefront (LDS, 12 B each) and the functor called on full
//    groups of 64: its memory-side atomics are waited for once per 64 SURVIVORS, not once per 64
//    edges;
//  - every wavefront on its own: 64-slot tiles scanned in the wave, batches of hub chunks claimed
//    per wave; the sixteen waves meet at two barriers only (bitmap loaded, queues drained).
// ---------------------------------------------------------------------------
#ifndef GRX_SET_BLOCK
#define GRX_SET_BLOCK 1024
#endif
constexpr int SET_BLOCK = GRX_SET_BLOCK;
constexpr int SET_WAVES = SET_BLOCK / wave_size;  // 16
constexpr int SET_WQCAP = 256;                    // output queue entries per wavefront
constexpr int SET_PENDING = 2 * wave_size;        // surviving edges a wavefront may hold
constexpr int SET_UNROLL = 4;                     // edges per lane and round

template <typename vertex_t, typename edge_t>
struct pending_edge_t {
  vertex_t source;
  vertex_t neighbor;
  edge_t edge;
};

template <advance_io_type_t IN,
          advance_io_type_t OUT,
          typename graph_t,
          typename op_t,
          typename vertex_t,
          typename edge_t>
__global__ void __launch_bounds__(SET_BLOCK)
    expand_settled_kernel(graph_t G,
                          op_t op,
                          const vertex_t* __restrict__ input,
                          std::size_t n_in,
                          vertex_t* __restrict__ output,
                          std::size_t capacity,
                          unsigned long long* counters,
                          const chunk_t<vertex_t, edge_t>* __restrict__ chunks,
                          unsigned long long chunk_capacity,
                          const unsigned long long* __restrict__ hub_mask,
                          unsigned long long* __restrict__ claim_cursors,
                          const unsigned long long* n_in_device = nullptr, int dealt = 0) {
  constexpr bool HAS_OUT = (OUT != advance_io_type_t::none);
  using weight_t = typename graph_t::weight_type;
  using pending_t = pending_edge_t<vertex_t, edge_t>;
  if (n_in_device)  // fused pipelines: the frontier length was written by a kernel earlier in the stream
    n_in = (std::size_t)__hip_atomic_load(n_in_device, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  extern __shared__ unsigned s_settled[];  // op.settled.limit bits
  __shared__ vertex_t s_vertex[SET_BLOCK];
  __shared__ edge_t s_first[SET_BLOCK];
  __shared__ unsigned s_scan[SET_BLOCK];
  __shared__ unsigned s_counts[SET_WAVES];
  __shared__ unsigned long long s_base;
  __shared__ vertex_t s_queue[HAS_OUT ? SET_WAVES * SET_WQCAP : 1];
  __shared__ pending_t s_pending[SET_WAVES * SET_PENDING];

  const int tid = threadIdx.x;
  const int wave = tid / wave_size;
  const int lane = lane_id();
  const vertex_t limit = op.settled.limit;
  {
    const unsigned groups = (unsigned)(op.lds_bytes() / 16u);  // the image is whole 16-byte groups
    const uint4* src = reinterpret_cast<const uint4*>(op.settled.bits);
    uint4* dst = reinterpret_cast<uint4*>(s_settled);
    for (unsigned w = tid; w < groups; w += SET_BLOCK)
      dst[w] = src[w];
  }
  __syncthreads();
  auto in_bitmap = [&](vertex_t n) -> bool {  // an unconditional LDS read: no branch, no wait per edge
    const unsigned at = (unsigned)n < (unsigned)limit ? (unsigned)n : 0u;
    return (unsigned)n < (unsigned)limit && ((s_settled[at >> 5] >> (at & 31u)) & 1u);
  };

  wave_queue_t<vertex_t> wq{s_queue + (HAS_OUT ? wave * SET_WQCAP : 0), 0u, 0ull};
  wq.cap = SET_WQCAP;
  auto degree_of = [&G](vertex_t x) -> unsigned { return (unsigned)G.get_number_of_neighbors(x); };
  vertex_t* const w_vertex = s_vertex + wave * wave_size;
  edge_t* const w_first = s_first + wave * wave_size;
  unsigned* const w_scan = s_scan + wave * wave_size;
  pending_t* const w_pending = s_pending + wave * SET_PENDING;
  unsigned n_pending = 0;  // wave-uniform

  // the functor, on the top `count` (<= 64) pending edges of this wavefront
#ifdef GRX_SETTLED_STATS
  unsigned long long st_calls = 0, st_survivors = 0, st_tested = 0, st_rounds = 0;
#endif
  auto call = [&](unsigned count) __attribute__((always_inline)) {
#ifdef GRX_SETTLED_STATS
    st_calls += 1;
    st_survivors += count;
#endif
    n_pending -= count;
    bool keep = false;
    vertex_t nbr = 0;
    if ((unsigned)lane < count) {
      const pending_t e = w_pending[n_pending + lane];
      nbr = e.neighbor;
#if defined(GRX_SETTLED_EXP) && GRX_SETTLED_EXP == 5  // timing experiment: packing and calling, no functor
      keep = e.neighbor < 0;
#else
      keep = op(e.source, e.neighbor, e.edge, G.get_edge_weight(e.edge));
#endif
#if defined(GRX_SETTLED_EXP) && GRX_SETTLED_EXP == 3  // timing experiment: functor without the output path
      keep = keep && nbr < 0;
#endif
    }
    if constexpr (HAS_OUT)
      wq.push_deferred(keep, nbr, output, capacity, counters, degree_of);
  };
  // one round = four edges per lane: bitmap, predicate (independent loads), pack, call on full
  // groups
  auto consider = [&](const vertex_t (&src)[SET_UNROLL], const vertex_t (&nbr)[SET_UNROLL],
                      const edge_t (&eid)[SET_UNROLL], const weight_t (&wgt)[SET_UNROLL],
                      const bool (&live)[SET_UNROLL]) __attribute__((always_inline)) {
    bool open[SET_UNROLL];
    bool by_cache[SET_UNROLL];  // 2-byte image: the edge is decided from LDS alone
    if constexpr (op_t::lds_image == 2) {
      const unsigned short* s_values = reinterpret_cast<const unsigned short*>(s_settled);
#pragma unroll
      for (int k = 0; k < SET_UNROLL; ++k) {
        const bool covered = (unsigned)nbr[k] < (unsigned)limit;
        const unsigned short value = s_values[covered ? (unsigned)nbr[k] : 0u];
        // unconditional (its source-label load flies with the other lanes'); masked below
        by_cache[k] = op.cached(src[k], nbr[k], eid[k], wgt[k], value);
        open[k] = !covered;
        by_cache[k] = by_cache[k] && covered;
      }
    } else {
#pragma unroll
      for (int k = 0; k < SET_UNROLL; ++k) {
        open[k] = !in_bitmap(nbr[k]);
        by_cache[k] = false;
      }
    }
#pragma unroll
    for (int k = 0; k < SET_UNROLL; ++k)
      open[k] = open[k] && live[k];
    // The predicate is evaluated by ALL lanes, unconditionally: behind a branch each of its loads
    // would be waited for in turn.  Lanes with nothing to ask look at vertex 0 (one hot line).
    bool done[SET_UNROLL];
#pragma unroll
    for (int k = 0; k < SET_UNROLL; ++k)
#if defined(GRX_SETTLED_EXP) && GRX_SETTLED_EXP == 1  // timing experiment: columns + LDS only
      done[k] = true;
#else
      done[k] = (bool)op.rejects(src[k], open[k] ? nbr[k] : vertex_t(0), eid[k], wgt[k]);
#endif
    __builtin_amdgcn_sched_barrier(0);  // all the loads first, then their consumers
#pragma unroll
    for (int k = 0; k < SET_UNROLL; ++k) {
      if constexpr (op_t::lds_image == 2)  // covered: the cache's verdict; not covered: the predicate's
        done[k] = open[k] ? done[k] : (by_cache[k] || !live[k]);
      else
        done[k] = done[k] || !open[k];
    }
#if defined(GRX_SETTLED_EXP) && GRX_SETTLED_EXP == 2  // timing experiment: lookups, no functor
#pragma unroll
    for (int k = 0; k < SET_UNROLL; ++k)
      done[k] = done[k] || nbr[k] >= 0;
#endif
#ifdef GRX_SETTLED_STATS
    st_rounds += 1;
#pragma unroll
    for (int k = 0; k < SET_UNROLL; ++k)
      st_tested += __popcll(__ballot(open[k]));
#endif
#pragma unroll
    for (int k = 0; k < SET_UNROLL; ++k) {
      const bool survive = !done[k];
      const unsigned long long m = __ballot(survive);
      if (m == 0)
        continue;
      if (survive)
        w_pending[n_pending + rank_in_mask(m)] = pending_t{src[k], nbr[k], eid[k]};
      n_pending += (unsigned)__popcll(m);
      __builtin_amdgcn_wave_barrier();
      if (n_pending >= (unsigned)wave_size)
        call(wave_size);
    }
  };
  // Rounds pass through a delay line of one: the column loads of round r + 1 are issued before
  // round r is considered, so they fly together with r's predicate loads.
  vertex_t p_src[SET_UNROLL], p_nbr[SET_UNROLL];
  edge_t p_eid[SET_UNROLL];
  weight_t p_wgt[SET_UNROLL];  // dead (and gone) when the predicate looks at the destination only
  bool p_live[SET_UNROLL];
  bool have_prev = false;  // wave-uniform
  auto submit = [&](const vertex_t (&src)[SET_UNROLL], const vertex_t (&nbr)[SET_UNROLL],
                    const edge_t (&eid)[SET_UNROLL], const weight_t (&wgt)[SET_UNROLL],
                    const bool (&live)[SET_UNROLL]) __attribute__((always_inline)) {
    if (have_prev)
      consider(p_src, p_nbr, p_eid, p_wgt, p_live);
#pragma unroll
    for (int k = 0; k < SET_UNROLL; ++k) {
      p_src[k] = src[k];
      p_nbr[k] = nbr[k];
      p_eid[k] = eid[k];
      p_wgt[k] = wgt[k];
      p_live[k] = live[k];
    }
    have_prev = true;
  };

  // ---- phase 1: 64-slot tiles, one per wavefront and step; hub slots are left to phase 2 ----------
  const unsigned long long n_tiles = (n_in + wave_size - 1) / wave_size;
  const unsigned long long wave_stride = (unsigned long long)gridDim.x * SET_WAVES;
  auto fetch_slot = [&](unsigned long long t, vertex_t& v, edge_t& first, edge_t& last) {
    v = gunrock::numeric_limits<vertex_t>::invalid();
    first = last = 0;
    // An ASCENDING frontier (operators::filter::select_range) on a hot-first numbered graph is sorted
    // by falling degree: 64 consecutive slots would make tiles of up to 64 x 255 edges next to tiles
    // of 64 (measured on RMAT-22: BFS level 2 246 us, SSSP iteration 3 374 us, with HALF the memory
    // traffic of the 229 / 246 us a frontier in discovery order takes -- one wavefront drags the
    // kernel).  Dealt ACROSS the frontier instead, lane l of tile t takes slot l * tiles + t: every
    // tile holds one row of each of 64 degree strata, and a lane walks its stratum front to back.
    const std::size_t idx = dealt ? (std::size_t)lane * n_tiles + t : (std::size_t)t * wave_size + lane;
    if (t < n_tiles && idx < n_in) {
      const vertex_t x = (IN == advance_io_type_t::graph) ? (vertex_t)idx : input[idx];
      const bool hub = (hub_mask[idx / wave_size] >> (idx % wave_size)) & 1ull;
      if (util::limits::is_valid(x) && !hub) {
        v = x;
        first = G.get_starting_edge(x);
        last = G.get_starting_edge(x + 1);
      }
    }
  };
  // dealt == 2: the sixteen wavefronts of a workgroup take CONSECUTIVE tiles, i.e. consecutive slots of
  // every stratum -- their 4-byte reads of the frontier, the hub mask and the row offsets fall into
  // lines the CU already holds (wave-major numbering spreads neighbouring tiles over all CUs and XCDs)
  unsigned long long tile = dealt == 2 ? (unsigned long long)blockIdx.x * SET_WAVES + wave
                                       : (unsigned long long)wave * gridDim.x + blockIdx.x;
  vertex_t v, v1;
  edge_t first, last, first1, last1;
  fetch_slot(tile, v, first, last);
  while (tile < n_tiles) {  // wave-uniform
    fetch_slot(tile + wave_stride, v1, first1, last1);
    const unsigned deg = (unsigned)(last - first);
    const unsigned incl = wave_inclusive_sum(deg);
    const unsigned total = __shfl(incl, wave_size - 1, wave_size);
    w_vertex[lane] = v;
    w_first[lane] = first;
    w_scan[lane] = incl - deg;
    __builtin_amdgcn_wave_barrier();  // LDS is in order within a wavefront; pin the compiler too
    for (unsigned i0 = 0; i0 < total; i0 += wave_size * SET_UNROLL) {
      vertex_t src[SET_UNROLL], nbr[SET_UNROLL];
      edge_t eid[SET_UNROLL];
      weight_t wgt[SET_UNROLL];
      bool live[SET_UNROLL];
#pragma unroll
      for (int k = 0; k < SET_UNROLL; ++k) {
        const unsigned i = i0 + k * wave_size + lane;
        live[k] = i < total;
        src[k] = nbr[k] = 0;
        eid[k] = 0;
        wgt[k] = weight_t(0);
        if (live[k]) {
          const int slot = rightmost_le(w_scan, i, wave_size);
          src[k] = w_vertex[slot];
          eid[k] = w_first[slot] + (edge_t)(i - w_scan[slot]);
          nbr[k] = G.get_destination_vertex(eid[k]);
          wgt[k] = G.get_edge_weight(eid[k]);
        }
      }
      submit(src, nbr, eid, wgt, live);
    }
    __builtin_amdgcn_wave_barrier();  // the tile arrays are rewritten; rounds carry their sources
    tile += wave_stride;
    v = v1;
    first = first1;
    last = last1;
  }

  // ---- phase 2: hub chunks, a batch per claim and wavefront --------------------------------------
  unsigned long long n_chunks =
      __hip_atomic_load(&counters[C_CHUNKS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (n_chunks > chunk_capacity)
    n_chunks = chunk_capacity;
  if (n_chunks) {  // grid-uniform
    const unsigned long long n_batches = (n_chunks + CLAIM_BATCH - 1) / CLAIM_BATCH;
    const unsigned pools = gridDim.x < 8u ? gridDim.x : 8u;
    unsigned pool = blockIdx.x % pools;  // lane 0's; the other lanes only receive batches
    unsigned tried = 0;
    unsigned long long pending = 0;
    if (lane == 0)
      pending = atomicAdd(&claim_cursors[pool * CLAIM_LINE], 1ull);
    for (;;) {
      unsigned long long batch = 0;
      if (lane == 0) {
        batch = pending * pools + pool;
        while (batch >= n_batches && ++tried < pools) {  // own pool exhausted: help the next one
          pool = (pool + 1) % pools;
          batch = atomicAdd(&claim_cursors[pool * CLAIM_LINE], 1ull) * pools + pool;
        }
        if (batch < n_batches)  // request the NEXT batch now; its round trip hides behind this one
          pending = atomicAdd(&claim_cursors[pool * CLAIM_LINE], 1ull);
      }
      batch = __shfl(batch, 0, wave_size);
      if (batch >= n_batches)
        break;
      const unsigned long long c0 = batch * CLAIM_BATCH;
      const unsigned long long c1 = c0 + CLAIM_BATCH < n_chunks ? c0 + CLAIM_BATCH : n_chunks;
      for (unsigned long long c = c0; c < c1; ++c) {
        const chunk_t<vertex_t, edge_t> d = chunks[c];
        for (int j0 = 0; j0 < d.count; j0 += wave_size * SET_UNROLL) {
          vertex_t src[SET_UNROLL], nbr[SET_UNROLL];
          edge_t eid[SET_UNROLL];
          weight_t wgt[SET_UNROLL];
          bool live[SET_UNROLL];
#pragma unroll
          for (int k = 0; k < SET_UNROLL; ++k) {
            const int j = j0 + k * wave_size + lane;
            live[k] = j < d.count;
            src[k] = d.source;
            eid[k] = live[k] ? d.first + (edge_t)j : edge_t(0);
            nbr[k] = live[k] ? G.get_destination_vertex(eid[k]) : vertex_t(0);
            wgt[k] = live[k] ? G.get_edge_weight(eid[k]) : weight_t(0);
          }
          submit(src, nbr, eid, wgt, live);
        }
      }
    }
  }
  if (have_prev)
    consider(p_src, p_nbr, p_eid, p_wgt, p_live);
#ifdef GRX_SETTLED_STATS
  if (lane == 0) {  // diagnostic build: functor calls, edges they carried, predicate tests, rounds
    atomicAdd(&counters[10], st_calls);
    atomicAdd(&counters[11], st_survivors);
    atomicAdd(&counters[13], st_tested);
    atomicAdd(&counters[14], st_rounds);
  }
#endif
  if (n_pending)  // wave-uniform; fewer than 64 are left
    call(n_pending);
  if constexpr (HAS_OUT)
    drain_block_summing<SET_WAVES>(wq, s_counts, &s_base, output, capacity, counters, degree_of);
}

// ---------------------------------------------------------------------------
// Hub chunks: persistent workgroups, one chunk (consecutive edges of one source)
// at a time, lanes on consecutive edges.
// ---------------------------------------------------------------------------
template <advance_io_type_t OUT, typename graph_t, typename op_t, typename vertex_t, typename edge_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    chunk_kernel(graph_t G,
                 op_t op,
                 const chunk_t<vertex_t, edge_t>* __restrict__ chunks,
                 unsigned long long chunk_capacity,
                 vertex_t* __restrict__ output,
                 std::size_t capacity,
                 unsigned long long* counters,
                 long long known_chunks = -1) {
  using weight_t = typename graph_t::weight_type;
  constexpr bool HAS_OUT = (OUT != advance_io_type_t::none);
  __shared__ unsigned s_counts[ADV_WAVES];
  __shared__ unsigned long long s_base;
  __shared__ vertex_t s_queue[HAS_OUT ? ADV_WAVES * ADV_WQCAP : 1];

  const int tid = threadIdx.x;
  wave_queue_t<vertex_t> wq{s_queue + (HAS_OUT ? (tid / wave_size) * ADV_WQCAP : 0), 0u, 0ull};
  auto degree_of = [&G](vertex_t x) -> unsigned { return (unsigned)G.get_number_of_neighbors(x); };

  // the queue length: still in the device counter when this launch directly follows the
  // producer, or passed by a host that already fetched (and thereby cleared) the counters
  unsigned long long n_chunks =
      known_chunks >= 0 ? (unsigned long long)known_chunks
                        : __hip_atomic_load(&counters[C_CHUNKS], __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
  if (n_chunks > chunk_capacity)
    n_chunks = chunk_capacity;  // the overflowed tail was expanded in place

  for (unsigned long long c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const chunk_t<vertex_t, edge_t> d = chunks[c];
    vertex_t source = d.source;
    for (int j0 = 0; j0 < d.count; j0 += ADV_BLOCK * ADV_UNROLL) {
      vertex_t nbr[ADV_UNROLL];
      edge_t eid[ADV_UNROLL];
      weight_t wgt[ADV_UNROLL];
      bool live[ADV_UNROLL];
#pragma unroll
      for (int k = 0; k < ADV_UNROLL; ++k) {
        const int j = j0 + k * ADV_BLOCK + tid;
        live[k] = j < d.count;
        if (live[k]) {
          eid[k] = d.first + (edge_t)j;
          nbr[k] = G.get_destination_vertex(eid[k]);
          wgt[k] = G.get_edge_weight(eid[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < ADV_UNROLL; ++k) {
        bool keep = false;
        if (live[k])
          keep = op(source, nbr[k], eid[k], wgt[k]);
        if constexpr (HAS_OUT) {
          wq.push_deferred(keep, nbr[k], output, capacity, counters, degree_of);
        }
      }
    }
  }
  if constexpr (HAS_OUT)
    drain_block_summing(wq, s_counts, &s_base, output, capacity, counters, degree_of);
}

// ---------------------------------------------------------------------------
// Hub chunks, wavefront-granular: every wavefront of the persistent grid takes one
// chunk at a time (64 consecutive edges per step, ADV_UNROLL steps in flight), so a
// chunk shorter than a workgroup's 1024-edge step wastes no lanes and the hub
// threshold can drop to a wavefront's width.
// ---------------------------------------------------------------------------
template <advance_io_type_t OUT, typename graph_t, typename op_t, typename vertex_t, typename edge_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    wave_chunk_kernel(graph_t G,
                      op_t op,
                      const chunk_t<vertex_t, edge_t>* __restrict__ chunks,
                      unsigned long long chunk_capacity,
                      vertex_t* __restrict__ output,
                      std::size_t capacity,
                      unsigned long long* counters,
                      long long known_chunks = -1) {
  using weight_t = typename graph_t::weight_type;
  constexpr bool HAS_OUT = (OUT != advance_io_type_t::none);
  __shared__ unsigned s_counts[ADV_WAVES];
  __shared__ unsigned long long s_base;
  __shared__ vertex_t s_queue[HAS_OUT ? ADV_WAVES * ADV_WQCAP : 1];

  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int wave = tid / wave_size;
  wave_queue_t<vertex_t> wq{s_queue + (HAS_OUT ? wave * ADV_WQCAP : 0), 0u, 0ull};

  unsigned long long n_chunks =
      known_chunks >= 0 ? (unsigned long long)known_chunks
                        : __hip_atomic_load(&counters[C_CHUNKS], __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
  if (n_chunks > chunk_capacity)
    n_chunks = chunk_capacity;
  const unsigned long long n_waves = (unsigned long long)gridDim.x * ADV_WAVES;
  for (unsigned long long c = (unsigned long long)blockIdx.x * ADV_WAVES + wave; c < n_chunks;
       c += n_waves) {
    const chunk_t<vertex_t, edge_t> d = chunks[c];
    vertex_t source = d.source;
    for (int j0 = 0; j0 < d.count; j0 += wave_size * ADV_UNROLL) {
      vertex_t nbr[ADV_UNROLL];
      edge_t eid[ADV_UNROLL];
      weight_t wgt[ADV_UNROLL];
      bool live[ADV_UNROLL];
#pragma unroll
      for (int k = 0; k < ADV_UNROLL; ++k) {
        const int j = j0 + k * wave_size + lane;
        live[k] = j < d.count;
        if (live[k]) {
          eid[k] = d.first + (edge_t)j;
          nbr[k] = G.get_destination_vertex(eid[k]);
          wgt[k] = G.get_edge_weight(eid[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < ADV_UNROLL; ++k) {
        bool keep = false;
        if (live[k])
          keep = op(source, nbr[k], eid[k], wgt[k]);
        if constexpr (HAS_OUT) {
          unsigned dn = 0;
          if (keep)
            dn = (unsigned)G.get_number_of_neighbors(nbr[k]);
          wq.push(keep, nbr[k], dn, output, capacity, counters);
        }
      }
    }
  }
  if constexpr (HAS_OUT)
    drain_block(wq, s_counts, &s_base, output, capacity, counters);
}

// ---------------------------------------------------------------------------
// thread_mapped: one lane per input slot (reference thread_mapped.hxx:59-95).
// Packed output through the wavefront queue; holes output at segments[slot]+rank.
// ---------------------------------------------------------------------------
template <bool HOLES,
          advance_io_type_t IN,
          advance_io_type_t OUT,
          typename graph_t,
          typename op_t,
          typename vertex_t,
          typename edge_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    thread_mapped_kernel(graph_t G,
                         op_t op,
                         const vertex_t* __restrict__ input,
                         std::size_t n_in,
                         const edge_t* __restrict__ segments,
                         vertex_t* __restrict__ output,
                         std::size_t capacity,
                         unsigned long long* counters) {
  using weight_t = typename graph_t::weight_type;
  constexpr bool HAS_OUT = (OUT != advance_io_type_t::none);
  constexpr bool PACKED = HAS_OUT && !HOLES;
  __shared__ unsigned s_counts[ADV_WAVES];
  __shared__ unsigned long long s_base;
  __shared__ vertex_t s_queue[PACKED ? ADV_WAVES * ADV_WQCAP : 1];
  const int tid = threadIdx.x;
  wave_queue_t<vertex_t> wq{s_queue + (PACKED ? (tid / wave_size) * ADV_WQCAP : 0), 0u, 0ull};

  const std::size_t n_tiles = (n_in + ADV_BLOCK - 1) / ADV_BLOCK;
  for (std::size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const std::size_t idx = tile * ADV_BLOCK + tid;
    vertex_t v = gunrock::numeric_limits<vertex_t>::invalid();
    edge_t first = 0;
    edge_t deg = 0;
    if (idx < n_in) {
      v = (IN == advance_io_type_t::graph) ? (vertex_t)idx : input[idx];
      if (util::limits::is_valid(v)) {
        first = G.get_starting_edge(v);
        deg = G.get_starting_edge(v + 1) - first;
      }
    }
    // the wavefront iterates to its longest list; shorter lanes idle
    const edge_t longest = wave_max(deg);
    for (edge_t r = 0; r < longest; ++r) {
      bool keep = false;
      vertex_t n = gunrock::numeric_limits<vertex_t>::invalid();
      if (r < deg) {
        edge_t e = first + r;
        n = G.get_destination_vertex(e);
        weight_t w = G.get_edge_weight(e);
        keep = op(v, n, e, w);
        if constexpr (HAS_OUT && HOLES) {
          const unsigned long long at = (unsigned long long)segments[idx] + (unsigned long long)r;
          if (at < capacity)
            output[at] = keep ? n : gunrock::numeric_limits<vertex_t>::invalid();
          else
            counters[C_OVERFLOW] = 1ull;
        }
      }
      if constexpr (PACKED) {
        unsigned dn = 0;
        if (keep)
          dn = (unsigned)G.get_number_of_neighbors(n);
        wq.push(keep, n, dn, output, capacity, counters);
      }
    }
  }
  if constexpr (PACKED)
    drain_block(wq, s_counts, &s_base, output, capacity, counters);
}

// ---------------------------------------------------------------------------
// warp_mapped: one wavefront per input slot, 64 lanes stride the list.
// ---------------------------------------------------------------------------
template <advance_io_type_t IN,
          advance_io_type_t OUT,
          typename graph_t,
          typename op_t,
          typename vertex_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    wave_mapped_kernel(graph_t G,
                       op_t op,
                       const vertex_t* __restrict__ input,
                       std::size_t n_in,
                       vertex_t* __restrict__ output,
                       std::size_t capacity,
                       unsigned long long* counters) {
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;
  constexpr bool HAS_OUT = (OUT != advance_io_type_t::none);
  __shared__ unsigned s_counts[ADV_WAVES];
  __shared__ unsigned long long s_base;
  __shared__ vertex_t s_queue[HAS_OUT ? ADV_WAVES * ADV_WQCAP : 1];
  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int wave = tid / wave_size;
  wave_queue_t<vertex_t> wq{s_queue + (HAS_OUT ? wave * ADV_WQCAP : 0), 0u, 0ull};

  // wavefront w of the grid takes slots w, w + W, w + 2W, ...
  const std::size_t n_waves = (std::size_t)gridDim.x * ADV_WAVES;
  for (std::size_t idx = (std::size_t)blockIdx.x * ADV_WAVES + wave; idx < n_in; idx += n_waves) {
    vertex_t v = (IN == advance_io_type_t::graph) ? (vertex_t)idx : input[idx];
    if (!util::limits::is_valid(v))
      continue;  // wave-uniform
    const edge_t first = G.get_starting_edge(v);
    const edge_t deg = G.get_starting_edge(v + 1) - first;
    for (edge_t r0 = 0; r0 < deg; r0 += wave_size) {
      const edge_t r = r0 + lane;
      bool keep = false;
      vertex_t n = gunrock::numeric_limits<vertex_t>::invalid();
      if (r < deg) {
        edge_t e = first + r;
        n = G.get_destination_vertex(e);
        weight_t w = G.get_edge_weight(e);
        keep = op(v, n, e, w);
      }
      if constexpr (HAS_OUT) {
        unsigned dn = 0;
        if (keep)
          dn = (unsigned)G.get_number_of_neighbors(n);
        wq.push(keep, n, dn, output, capacity, counters);
      }
    }
  }
  if constexpr (HAS_OUT)
    drain_block(wq, s_counts, &s_base, output, capacity, counters);
}

// ---------------------------------------------------------------------------
// merge_path: segments[] = exclusive scan of the input slots' degrees
// (segments[n_in] = total).  Persistent workgroups take MP_TILE consecutive WORK
// ITEMS (edges) at a time, find the slots that overlap them, stage those in LDS
// and expand.  Output position in holes mode is the work-item index itself
// (deterministic, like reference merge_path.hxx:104-105).
// ---------------------------------------------------------------------------
constexpr int MP_EPT = 4;
constexpr int MP_TILE = ADV_BLOCK * MP_EPT;  // 1024 edges per step
constexpr int MP_SLOTS = 1024;               // slots staged in LDS per step

template <typename edge_t>
__device__ __forceinline__ std::size_t global_rightmost_le(const edge_t* seg, std::size_t n,
                                                           unsigned long long key) {
  // largest s in [0, n) with seg[s] <= key  (seg[0] == 0 <= key)
  std::size_t lo = 0, hi = n;
  while (hi - lo > 1) {
    std::size_t mid = lo + ((hi - lo) >> 1);
    if ((unsigned long long)seg[mid] <= key)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

template <bool HOLES,
          advance_io_type_t IN,
          advance_io_type_t OUT,
          typename graph_t,
          typename op_t,
          typename vertex_t,
          typename edge_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    merge_path_kernel(graph_t G,
                      op_t op,
                      const vertex_t* __restrict__ input,
                      std::size_t n_in,
                      const edge_t* __restrict__ segments,
                      unsigned long long total_work,
                      vertex_t* __restrict__ output,
                      std::size_t capacity,
                      unsigned long long* counters) {
  using weight_t = typename graph_t::weight_type;
  constexpr bool HAS_OUT = (OUT != advance_io_type_t::none);
  constexpr bool PACKED = HAS_OUT && !HOLES;
  __shared__ vertex_t s_vertex[MP_SLOTS];
  __shared__ edge_t s_first[MP_SLOTS];
  __shared__ edge_t s_seg[MP_SLOTS];
  __shared__ unsigned s_counts[ADV_WAVES];
  __shared__ unsigned long long s_base;
  __shared__ vertex_t s_queue[PACKED ? ADV_WAVES * ADV_WQCAP : 1];
  const int tid = threadIdx.x;
  wave_queue_t<vertex_t> wq{s_queue + (PACKED ? (tid / wave_size) * ADV_WQCAP : 0), 0u, 0ull};

  const unsigned long long n_steps = (total_work + MP_TILE - 1) / MP_TILE;
  for (unsigned long long step = blockIdx.x; step < n_steps; step += gridDim.x) {
    const unsigned long long w0 = step * MP_TILE;
    const unsigned long long w1 = (w0 + MP_TILE < total_work) ? w0 + MP_TILE : total_work;

    // slots overlapping [w0, w1): every lane runs the same two searches (broadcast loads)
    const std::size_t slot_lo = global_rightmost_le(segments, n_in, w0);
    const std::size_t slot_hi = global_rightmost_le(segments, n_in, w1 - 1);
    const std::size_t n_slots = slot_hi - slot_lo + 1;
    const bool staged = n_slots <= (std::size_t)MP_SLOTS;
    if (staged) {
      for (std::size_t s = tid; s < n_slots; s += ADV_BLOCK) {
        const std::size_t idx = slot_lo + s;
        vertex_t v = (IN == advance_io_type_t::graph) ? (vertex_t)idx : input[idx];
        s_vertex[s] = v;
        s_seg[s] = segments[idx];
        s_first[s] = util::limits::is_valid(v) ? G.get_starting_edge(v) : (edge_t)0;
      }
    }
    __syncthreads();

#pragma unroll
    for (int k = 0; k < MP_EPT; ++k) {
      const unsigned long long i = w0 + (unsigned long long)k * ADV_BLOCK + tid;
      bool keep = false;
      vertex_t n = gunrock::numeric_limits<vertex_t>::invalid();
      if (i < w1) {
        vertex_t v;
        edge_t e;
        if (staged) {
          const int s = rightmost_le(s_seg, (edge_t)i, (int)n_slots);
          v = s_vertex[s];
          e = s_first[s] + ((edge_t)i - s_seg[s]);
        } else {
          const std::size_t idx = slot_lo + global_rightmost_le(segments + slot_lo, n_slots, i);
          v = (IN == advance_io_type_t::graph) ? (vertex_t)idx : input[idx];
          e = G.get_starting_edge(v) + ((edge_t)i - segments[idx]);
        }
        n = G.get_destination_vertex(e);
        weight_t w = G.get_edge_weight(e);
        keep = op(v, n, e, w);
        if constexpr (HAS_OUT && HOLES) {
          if (i < capacity)
            output[i] = keep ? n : gunrock::numeric_limits<vertex_t>::invalid();
          else
            counters[C_OVERFLOW] = 1ull;
        }
      }
      if constexpr (PACKED) {
        unsigned dn = 0;
        if (keep)
          dn = (unsigned)G.get_number_of_neighbors(n);
        wq.push(keep, n, dn, output, capacity, counters);
      }
    }
    __syncthreads();  // LDS staging arrays are rewritten by the next step
  }
  if constexpr (PACKED)
    drain_block(wq, s_counts, &s_base, output, capacity, counters);
}

/// Degree of every input slot (0 for invalid), written for the device-wide scan.
template <advance_io_type_t IN, typename graph_t, typename vertex_t, typename edge_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    slot_degree_kernel(graph_t G, const vertex_t* input, std::size_t n_in, edge_t* degrees) {
  for (std::size_t i = blockIdx.x * (std::size_t)ADV_BLOCK + threadIdx.x; i <= n_in;
       i += (std::size_t)gridDim.x * ADV_BLOCK) {
    edge_t d = 0;
    if (i < n_in) {
      vertex_t v = (IN == advance_io_type_t::graph) ? (vertex_t)i : input[i];
      if (util::limits::is_valid(v))
        d = G.get_number_of_neighbors(v);
    }
    degrees[i] = d;  // entry n_in is the scan's sentinel (reference helpers.hxx:55-57)
  }
}

// ---------------------------------------------------------------------------
// PULL (advance_direction_t::backward).  The input frontier holds CANDIDATE
// destinations u (e.g. the still-unvisited vertices).  For each valid u the
// in-edges of u are walked in order and op(v, u, e, w) is called with v the
// in-neighbour, until the op returns true or the list ends; u is emitted when it
// did.  (For an undirected graph the CSR view is its own transpose.)
//   pull_probe_kernel : one lane per candidate, at most PULL_PROBES edges; a
//                       candidate still undecided with edges left goes to the long
//                       queue as (u, resume offset)
//   pull_long_kernel  : one wavefront per queued candidate, 64 edges per step,
//                       ballot early exit
// On a power-law graph most candidates of a wide BFS level meet a frontier
// neighbour within the first few probes: the label / bitmap lookups drop from one
// per edge of the frontier to a few per unvisited vertex.
// ---------------------------------------------------------------------------
constexpr int PULL_PROBES = 16;

template <typename vertex_t>
struct resume_t {
  vertex_t vertex;
  int offset;
};

/// Append `value` of the lanes with `keep` to a global list: one atomic per wavefront.
template <typename vertex_t>
__device__ __forceinline__ void wave_append(bool keep, vertex_t value, vertex_t* list,
                                            unsigned long long list_capacity,
                                            unsigned long long* cursor,
                                            unsigned long long* overflow) {
  const unsigned long long m = __ballot(keep);
  if (!m)
    return;
  unsigned long long base = 0;
  if (lane_id() == 0)
    base = atomicAdd(cursor, (unsigned long long)__popcll(m));
  base = __shfl(base, 0, wave_size);
  if (keep) {
    const unsigned long long at = base + rank_in_mask(m);
    if (at < list_capacity)
      list[at] = value;
    else
      *overflow = 1ull;
  }
}

template <advance_io_type_t OUT, bool REJECTS, typename graph_t, typename op_t, typename vertex_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    pull_probe_kernel(graph_t G,
                      op_t op,
                      const vertex_t* __restrict__ candidates,
                      std::size_t n_candidates,
                      vertex_t* __restrict__ output,
                      std::size_t capacity,
                      vertex_t* __restrict__ rejected,
                      resume_t<vertex_t>* long_queue,
                      unsigned long long long_capacity,
                      unsigned long long* counters,
                      const typename graph_t::edge_type* __restrict__ forward_offsets = nullptr) {
  // forward_offsets: row offsets of the FORWARD graph when G is a separate in-edge view (a
  // directed graph with an attached transpose).  The work hint left on the output frontier is the
  // sum of OUT-degrees -- what the next push advance expands -- not of the in-degrees walked here.
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;
  constexpr bool HAS_OUT = (OUT != advance_io_type_t::none);
  __shared__ unsigned s_counts[ADV_WAVES];
  __shared__ unsigned long long s_base;
  __shared__ vertex_t s_queue[HAS_OUT ? ADV_WAVES * ADV_WQCAP : 1];
  __shared__ vertex_t s_rejects[REJECTS ? ADV_WAVES * ADV_WQCAP : 1];
  const int tid = threadIdx.x;
  const int lane = lane_id();
  wave_queue_t<vertex_t> wq{s_queue + (HAS_OUT ? (tid / wave_size) * ADV_WQCAP : 0), 0u, 0ull};
  wave_queue_t<vertex_t> rq{s_rejects + (REJECTS ? (tid / wave_size) * ADV_WQCAP : 0), 0u, 0ull,
                            C_BUCKET0};

  const std::size_t n_tiles = (n_candidates + ADV_BLOCK - 1) / ADV_BLOCK;
  for (std::size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const std::size_t idx = tile * ADV_BLOCK + tid;
    vertex_t u = gunrock::numeric_limits<vertex_t>::invalid();
    edge_t first = 0;
    edge_t deg = 0;
    if (idx < n_candidates) {
      u = candidates[idx];
      if (util::limits::is_valid(u)) {
        first = G.get_starting_edge(u);
        deg = G.get_starting_edge(u + 1) - first;
      }
    }
    bool hit = false;
    edge_t r = 0;
    const edge_t stop = deg < (edge_t)PULL_PROBES ? deg : (edge_t)PULL_PROBES;
    for (; r < stop; ++r) {
      edge_t e = first + r;
      vertex_t v = G.get_destination_vertex(e);
      weight_t w = G.get_edge_weight(e);
      if (op(v, u, e, w)) {
        hit = true;
        break;
      }
    }
    // undecided with edges left: hand over to a whole wavefront
    const bool more = !hit && r < deg;
    const unsigned long long mm = __ballot(more);
    if (mm) {
      unsigned long long base = 0;
      if (lane == 0)
        base = atomicAdd(&counters[C_CHUNKS], (unsigned long long)__popcll(mm));
      base = __shfl(base, 0, wave_size);
      if (more) {
        const unsigned long long at = base + rank_in_mask(mm);
        if (at < long_capacity)
          long_queue[at] = resume_t<vertex_t>{u, (int)r};
        else
          counters[C_OVERFLOW] = 1ull;
      }
    }
    if constexpr (HAS_OUT) {
      unsigned dn = 0;
      if (hit)
        dn = forward_offsets ? (unsigned)(forward_offsets[u + 1] - forward_offsets[u]) : (unsigned)deg;
      wq.push(hit, u, dn, output, capacity, counters);
    }
    if constexpr (REJECTS) {
      // candidates whose whole list was walked without a hit stay candidates
      const bool rejected_here = util::limits::is_valid(u) && !hit && !more && deg > 0;
      rq.push(rejected_here, u, 0u, rejected, n_candidates, counters);
    }
  }
  if constexpr (HAS_OUT)
    drain_block(wq, s_counts, &s_base, output, capacity, counters);
  if constexpr (REJECTS)
    drain_block(rq, s_counts, &s_base, rejected, n_candidates, counters);
}

template <advance_io_type_t OUT, bool REJECTS, typename graph_t, typename op_t, typename vertex_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    pull_long_kernel(graph_t G,
                     op_t op,
                     const resume_t<vertex_t>* __restrict__ long_queue,
                     unsigned long long long_capacity,
                     vertex_t* __restrict__ output,
                     std::size_t capacity,
                     vertex_t* __restrict__ rejected,
                     unsigned long long rejected_capacity,
                     unsigned long long* counters,
                     const typename graph_t::edge_type* __restrict__ forward_offsets = nullptr) {
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;
  constexpr bool HAS_OUT = (OUT != advance_io_type_t::none);
  __shared__ unsigned s_counts[ADV_WAVES];
  __shared__ unsigned long long s_base;
  __shared__ vertex_t s_queue[HAS_OUT ? ADV_WAVES * ADV_WQCAP : 1];
  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int wave = tid / wave_size;
  __shared__ vertex_t s_rejects[REJECTS ? ADV_WAVES * ADV_WQCAP : 1];
  wave_queue_t<vertex_t> wq{s_queue + (HAS_OUT ? wave * ADV_WQCAP : 0), 0u, 0ull};
  wave_queue_t<vertex_t> rq{s_rejects + (REJECTS ? wave * ADV_WQCAP : 0), 0u, 0ull, C_BUCKET0};

  unsigned long long n =
      __hip_atomic_load(&counters[C_CHUNKS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (n > long_capacity)
    n = long_capacity;
  const unsigned long long n_waves = (unsigned long long)gridDim.x * ADV_WAVES;
  for (unsigned long long q = (unsigned long long)blockIdx.x * ADV_WAVES + wave; q < n; q += n_waves) {
    const resume_t<vertex_t> item = long_queue[q];
    vertex_t u = item.vertex;
    const edge_t first = G.get_starting_edge(u);
    const edge_t deg = G.get_starting_edge(u + 1) - first;
    bool found = false;
    for (edge_t r0 = (edge_t)item.offset; r0 < deg && !found; r0 += wave_size) {
      const edge_t r = r0 + lane;
      bool hit = false;
      if (r < deg) {
        edge_t e = first + r;
        vertex_t v = G.get_destination_vertex(e);
        weight_t w = G.get_edge_weight(e);
        hit = op(v, u, e, w);
      }
      found = __ballot(hit) != 0ull;  // wave-uniform
    }
    if constexpr (HAS_OUT) {
      // lane 0 speaks for the wavefront
      unsigned dn = 0;
      if (found && lane == 0)
        dn = forward_offsets ? (unsigned)(forward_offsets[u + 1] - forward_offsets[u]) : (unsigned)deg;
      wq.push(found && lane == 0, u, dn, output, capacity, counters);
    }
    if constexpr (REJECTS)
      rq.push(!found && lane == 0, u, 0u, rejected, (std::size_t)rejected_capacity, counters);
  }
  if constexpr (HAS_OUT)
    drain_block(wq, s_counts, &s_base, output, capacity, counters);
  if constexpr (REJECTS)
    drain_block(rq, s_counts, &s_base, rejected, (std::size_t)rejected_capacity, counters);
}

// ---------------------------------------------------------------------------
// bucketing: bin valid input slots by degree into three queues.
//   small  (< BUCKET_SMALL)  -> thread-per-slot        (thread_mapped_kernel)
//   medium (< hub_threshold) -> wavefront-per-slot     (wave_mapped_kernel)
//   large                    -> equal chunks           (chunk_kernel)
// Davidson et al.'s SSSP schedule, which the reference names but leaves empty
// (advance/bucketing.hxx:24-36).
// ---------------------------------------------------------------------------
constexpr unsigned BUCKET_SMALL = 16;

template <advance_io_type_t IN, typename graph_t, typename vertex_t, typename edge_t>
__global__ void __launch_bounds__(ADV_BLOCK)
    bucket_kernel(graph_t G,
                  const vertex_t* __restrict__ input,
                  std::size_t n_in,
                  vertex_t* __restrict__ small_q,
                  vertex_t* __restrict__ medium_q,
                  chunk_t<vertex_t, edge_t>* chunks,
                  unsigned long long chunk_capacity,
                  unsigned hub_threshold,
                  unsigned chunk_edges,
                  unsigned long long* counters) {
  // two wavefront queues: small and medium ids
  __shared__ vertex_t s_small[ADV_WAVES * ADV_WQCAP];
  __shared__ vertex_t s_medium[ADV_WAVES * ADV_WQCAP];
  const int tid = threadIdx.x;
  const int wave = tid / wave_size;
  const int lane = lane_id();
  unsigned n_small = 0, n_medium = 0;  // wave-uniform
  vertex_t* qs = s_small + wave * ADV_WQCAP;
  vertex_t* qm = s_medium + wave * ADV_WQCAP;

  auto flush = [&](vertex_t* q, unsigned& fill, vertex_t* out, int slot) {
    if (fill == 0)
      return;
    unsigned long long base = 0;
    if (lane == 0)
      base = atomicAdd(&counters[slot], (unsigned long long)fill);
    base = __shfl(base, 0, wave_size);
    for (unsigned j = lane; j < fill; j += wave_size)
      out[base + j] = q[j];
    fill = 0;
  };

  const std::size_t stride = (std::size_t)gridDim.x * ADV_BLOCK;
  const std::size_t rounds = (n_in + stride - 1) / stride;
  for (std::size_t r = 0; r < rounds; ++r) {
    const std::size_t idx = r * stride + blockIdx.x * (std::size_t)ADV_BLOCK + tid;
    vertex_t v = gunrock::numeric_limits<vertex_t>::invalid();
    unsigned deg = 0;
    edge_t first = 0;
    if (idx < n_in) {
      v = (IN == advance_io_type_t::graph) ? (vertex_t)idx : input[idx];
      if (util::limits::is_valid(v)) {
        first = G.get_starting_edge(v);
        deg = (unsigned)(G.get_starting_edge(v + 1) - first);
      }
    }
    // three disjoint classes even when the hub threshold is set below BUCKET_SMALL
    const unsigned small_limit = hub_threshold < BUCKET_SMALL ? hub_threshold : BUCKET_SMALL;
    const bool is_small = deg > 0 && deg < small_limit;
    bool is_medium = deg >= small_limit && deg < hub_threshold;
    // large lists: wave-aggregated chunk reservation
    const unsigned my_chunks = (deg >= hub_threshold) ? (deg + chunk_edges - 1) / chunk_edges : 0u;
    const unsigned incl = wave_inclusive_sum(my_chunks);
    const unsigned wave_chunks = __shfl(incl, wave_size - 1, wave_size);
    if (wave_chunks) {  // wave-uniform
      unsigned long long base = 0;
      if (lane == 0)
        base = atomicAdd(&counters[C_CHUNKS], (unsigned long long)wave_chunks);
      base = __shfl(base, 0, wave_size);
      const bool fits = base + wave_chunks <= chunk_capacity;
      const unsigned long long at = base + (incl - my_chunks);
      for (unsigned c = 0; c < my_chunks && at + c < chunk_capacity; ++c) {
        const unsigned off = c * chunk_edges;
        chunk_t<vertex_t, edge_t> d;
        d.source = v;
        d.first = first + (edge_t)off;
        d.count = fits ? (int)((deg - off < chunk_edges) ? deg - off : chunk_edges) : 0;
        chunks[at + c] = d;
      }
      if (!fits && my_chunks)
        is_medium = true;  // queue full: a wavefront walks it
    }
    unsigned long long ms = __ballot(is_small);
    if (ms) {
      if (n_small + wave_size > (unsigned)ADV_WQCAP)
        flush(qs, n_small, small_q, C_BUCKET0);
      if (is_small)
        qs[n_small + rank_in_mask(ms)] = v;
      n_small += (unsigned)__popcll(ms);
    }
    unsigned long long mm = __ballot(is_medium);
    if (mm) {
      if (n_medium + wave_size > (unsigned)ADV_WQCAP)
        flush(qm, n_medium, medium_q, C_BUCKET0 + 1);
      if (is_medium)
        qm[n_medium + rank_in_mask(mm)] = v;
      n_medium += (unsigned)__popcll(mm);
    }
  }
  flush(qs, n_small, small_q, C_BUCKET0);
  flush(qm, n_medium, medium_q, C_BUCKET0 + 1);
}

}  // namespace kernels
}  // namespace hip
}  // namespace gunrock
