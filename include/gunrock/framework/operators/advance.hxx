/**
 * @file advance.hxx
 * @brief operators::advance::execute -- neighbour expansion of a frontier.
 *
 * API of reference framework/operators/advance/advance.hxx:91-129 (frontier
 * overload) and :192-221 (enactor overload): template arguments <load balance,
 * direction, input type, output type>, functor bool(src, dst, edge, weight),
 * buffers swapped afterwards unless output_type == none or swap_buffers == false.
 * Unsupported combinations throw error::exception_t, like the reference.
 *
 * Host-side structure (per call, all on the context's stream):
 *   1. clear the device counters (one 64-byte memset);
 *   2. size the output: skipped when n_in * max_degree(G) already fits the output
 *      frontier; otherwise one degree-sum kernel + a pinned read-back (the
 *      reference does this reduction on EVERY call: advance/helpers.hxx:112-146);
 *   3. the expansion kernel(s) of the chosen schedule;
 *   4. ONE 8-byte read-back of the packed length + stream synchronise (operators
 *      are synchronous, like block_mapped.hxx:204).
 * No allocation happens in the steady state (the reference allocates a device
 * cursor per call, block_mapped.hxx:200).
 */
#pragma once

#include <gunrock/framework/operators/by_destination.hxx>
#include <gunrock/framework/operators/configs.hxx>
#include <gunrock/hip/context.hxx>
#include <gunrock/hip/kernels/advance_kernels.hxx>
#include <gunrock/hip/primitives.hxx>

#include <atomic>

/// Compile-time override of the schedule hard-coded by a client header, e.g.
/// -DGRX_ADVANCE_LB_OVERRIDE=bucketing to run the unchanged sssp.hxx (which
/// spells block_mapped, algorithms/sssp.hxx:139) with degree bucketing.
#ifdef GRX_ADVANCE_LB_OVERRIDE
#define GRX_LB_EFFECTIVE(lb) (::gunrock::operators::load_balance_t::GRX_ADVANCE_LB_OVERRIDE)
#else
#define GRX_LB_EFFECTIVE(lb) (lb)
#endif

namespace gunrock {
namespace operators {
namespace advance {

namespace detail {

namespace k = ::gunrock::hip::kernels;

inline unsigned grid_for(std::size_t items, std::size_t per_block, unsigned cap = 0x7fffffffu) {
  std::size_t g = (items + per_block - 1) / per_block;
  if (g < 1)
    g = 1;
  return (unsigned)(g > cap ? cap : g);
}

/**
 * @brief Hand the device counters to the host WITHOUT a memcpy command, a memset command or a
 * stream-synchronise call: a one-lane kernel copies the first 16 counters into the pinned
 * mirror, zeroes them for the next operator and then stores a sequence number; the host spins
 * on that word.  (The reference pays a thrust reduce + D2H + cudaStreamSynchronize + a
 * cudaMalloc'ed cursor per advance: block_mapped.hxx:160-204.)  Invariant: counters 0..15 are
 * zero whenever no operator is in flight.
 */
template <int header_only = 0>  // a template so that every translation unit may define it
__global__ void publish_counters_kernel(unsigned long long* counters, unsigned long long* mirror,
                                        unsigned long long sequence, long long* copy_to,
                                        int copy_slot, unsigned long long* zero_this) {
  const int i = threadIdx.x;
#ifdef GRX_TILE_TIMING
  if (i < 31) {  // diagnostic build: slots 24..30 carry the tile kernel's phase clocks
#else
  if (i < 24) {  // 0..15 operator counters, 16..23 tile-pool cursors
#endif
    // the counters were updated by device-scope atomics (memory side); read and clear them
    // with cache-bypassing accesses instead of trusting what this XCD's L2 may still hold
    const unsigned long long value =
        __hip_atomic_exchange(&counters[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mirror[i] = value;
    // fused pipelines: leave one counter where the next device-side consumer reads it (a send
    // slot's header) and clear one device word (a frontier length the next admit accumulates)
    if (copy_to && i == copy_slot)
      *copy_to = (long long)value;
  }
  if (zero_this && i == 32)
    __hip_atomic_store(zero_this, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __threadfence_system();
  __syncthreads();
  if (i == 0) {
    __hip_atomic_store(&mirror[gcuda::workspace_t::sequence_slot], sequence, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

/// Enqueue the hand-off (copy to the mirror, clear, stamp); returns the sequence number.
inline unsigned long long publish_counters(gcuda::standard_context_t& ctx, long long* copy_to = nullptr,
                                           int copy_slot = 0,
                                           unsigned long long* zero_this = nullptr) {
  auto& ws = ctx.workspace();
  const unsigned long long seq = ws.next_sequence();
  publish_counters_kernel<0><<<1, 64, 0, ctx.stream()>>>(ws.counters(), ws.mirror(), seq, copy_to,
                                                         copy_slot, zero_this);
  GRX_HIP_CHECK(hipGetLastError());
  return seq;
}

/// Wait until hand-off `seq` has landed; returns the pinned mirror.
inline unsigned long long* await_counters(gcuda::standard_context_t& ctx, unsigned long long seq) {
  auto& ws = ctx.workspace();
  volatile unsigned long long* flag = ws.mirror() + gcuda::workspace_t::sequence_slot;
  unsigned spins = 0;
  while (*flag < seq) {
    __builtin_ia32_pause();
    if ((++spins & 0xFFFFu) == 0) {
      // every ~100 us: make sure the stream is still healthy (a faulted kernel never publishes)
      hipError_t st = hipStreamQuery(ctx.stream());
      if (st != hipSuccess && st != hipErrorNotReady)
        error::throw_if_exception(st, "operator kernels failed");
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return ws.mirror();
}

/// Publish the counters and wait for them; returns the pinned mirror.
inline unsigned long long* fetch_counters(gcuda::standard_context_t& ctx) {
  return await_counters(ctx, publish_counters(ctx));
}

struct clocked_t {
  gcuda::standard_context_t& ctx;
  explicit clocked_t(gcuda::standard_context_t& c) : ctx(c) {
    if (ctx.options().time_kernels)
      ctx.kernel_clock().start(ctx.stream());
  }
  void stop() {
    if (ctx.options().time_kernels)
      ctx.kernel_clock().stop(ctx.stream());
  }
};

template <typename graph_t>
unsigned long long max_degree(graph_t& G, gcuda::standard_context_t& ctx) {
  if (G.properties.max_degree)  // the view's builder knows it
    return G.properties.max_degree;
  auto& ws = ctx.workspace();
  const void* key = (const void*)G.get_row_offsets();
  const std::size_t n = (std::size_t)G.get_number_of_vertices();
  const std::size_t m = (std::size_t)G.get_number_of_edges();
  if (auto* f = ws.find_graph(key, n, m))
    return f->max_degree;
  unsigned long long* counters = ws.counters();
  if (n) {
    k::max_degree_kernel<<<grid_for(n, k::ADV_BLOCK, 1024), k::ADV_BLOCK, 0, ctx.stream()>>>(
        G, counters);
    GRX_HIP_CHECK(hipGetLastError());
  }
  unsigned long long md = fetch_counters(ctx)[k::C_MAXDEG];
  if (std::getenv("GRX_DEBUG"))
    std::fprintf(stderr, "[grx] max_degree(%zu vertices) = %llu (seq %llu)\n", n, md,
                 ws.mirror()[gcuda::workspace_t::sequence_slot]);
  gcuda::workspace_t::graph_facts_t facts{key, n, md, m};
  return ws.remember_graph(facts)->max_degree;
}

/// Sum of degrees of the valid input slots (64-bit).
template <advance_io_type_t input_type, typename graph_t, typename vertex_t>
unsigned long long degree_sum(graph_t& G, const vertex_t* input, std::size_t n_in,
                              gcuda::standard_context_t& ctx) {
  if (input_type == advance_io_type_t::graph)
    return (unsigned long long)G.get_number_of_edges();
  k::degree_sum_kernel<input_type><<<grid_for(n_in, k::ADV_BLOCK, 1024), k::ADV_BLOCK, 0,
                                     ctx.stream()>>>(G, input, n_in, ctx.workspace().counters());
  GRX_HIP_CHECK(hipGetLastError());
  return fetch_counters(ctx)[k::C_WORK];
}

/// How an ascending frontier is dealt across the tiles of the wide-level kernel (advance_kernels.hxx):
/// 2 = across the tiles, workgroup-major; 1 = across the tiles, wave-major (GRX_DEALT_MODE, experiments).
inline int dealt_mode() {
  static const int mode = std::getenv("GRX_DEALT_MODE") ? std::atoi(std::getenv("GRX_DEALT_MODE")) : 2;
  return mode;
}

inline unsigned long long saturating_mul(unsigned long long a, unsigned long long b) {
  if (a == 0 || b == 0)
    return 0;
  if (a > ~0ull / b)
    return ~0ull;
  return a * b;
}

/**
 * @brief Make sure `output` can take the result.  Returns false when the
 * operator has nothing to do (no work), with the output already set empty.
 * `total` receives the exact work size when it had to be computed (else ~0).
 */
template <advance_io_type_t input_type, typename graph_t, typename frontier_t>
bool size_output(graph_t& G, frontier_t& input, frontier_t& output, std::size_t n_in, bool exact,
                 unsigned long long& total, gcuda::standard_context_t& ctx) {
  total = ~0ull;
  unsigned long long bound = saturating_mul(n_in, max_degree(G, ctx));
  if (input_type != advance_io_type_t::graph && input.work_hint() < bound)
    bound = input.work_hint();  // left by the operator that produced this frontier
  if (input_type == advance_io_type_t::graph)
    bound = (unsigned long long)G.get_number_of_edges();
  if (!exact && bound <= output.get_capacity())
    return true;
  total = degree_sum<input_type>(G, input.data(), n_in, ctx);
  if (total == 0) {
    output.set_number_of_elements(0);
    return false;
  }
  if (output.get_capacity() < total)
    output.reserve(total);
  return true;
}

template <typename frontier_t>
void finish_output(frontier_t& output, bool holes, unsigned long long total,
                   gcuda::standard_context_t& ctx) {
  unsigned long long* m = fetch_counters(ctx);
  ctx.kernel_clock().collect();
  if (std::getenv("GRX_DEBUG"))
    std::fprintf(stderr, "[grx] advance done: out %llu chunks %llu next_work %llu\n", m[k::C_OUT],
                 m[k::C_CHUNKS], m[k::C_NEXT_WORK]);
#ifdef GRX_TILE_TIMING
  if (std::getenv("GRX_DEBUG") && m[30])
    std::fprintf(stderr,
                 "[grx] tile timing (us, mean per workgroup of %llu): stage %.1f edges %.1f drain %.1f "
                 "total %.1f max-total %.1f | tiles/wg %.2f iters/wg %.2f\n",
                 m[30], m[24] / 100.0 / m[30], m[25] / 100.0 / m[30], m[29] / 100.0 / m[30],
                 m[28] / 100.0 / m[30], m[7] / 100.0, (double)m[27] / m[30], (double)m[26] / m[30]);
#endif
  error::throw_if_exception(m[k::C_OVERFLOW] != 0,
                            "advance: output frontier capacity exceeded");
  output.set_number_of_elements(holes ? (std::size_t)total : (std::size_t)m[k::C_OUT]);
  if (!holes)
    output.set_work_hint(m[k::C_NEXT_WORK]);
}

/// Input slots per tile of the block_mapped kernel: 256.  Narrower tiles were measured slower on
/// wide levels (more partially filled 1024-edge steps) AND on narrow ones (a 163 K-vertex level at
/// 64 slots: BFS 1.47 vs 1.39 ms) -- DESIGN.md section 5; options().tile_width is an experiment knob.
inline unsigned tile_width_for(std::size_t n_in, unsigned persistent_workgroups,
                               gcuda::standard_context_t& ctx) {
  (void)n_in;
  (void)persistent_workgroups;
  unsigned w = ctx.options().tile_width;
  if (w == 0)
    return (unsigned)k::ADV_BLOCK;
  unsigned p = 16;  // round down to a power of two in [16, 256]
  while (p * 2 <= w && p * 2 <= (unsigned)k::ADV_BLOCK)
    p *= 2;
  return p;
}

/// Device chunk queue sized for every hub of this call: a list of d edges makes at most
/// d / chunk_edges + 1 chunks, and only lists of >= hub_threshold edges make any.
template <typename vertex_t, typename edge_t, typename graph_t>
k::chunk_t<vertex_t, edge_t>* chunk_queue(graph_t& G, std::size_t n_in, unsigned long long work_bound,
                                          unsigned long long& capacity,
                                          gcuda::standard_context_t& ctx) {
  const unsigned chunk_edges = ctx.options().chunk_edges ? ctx.options().chunk_edges : 1024u;
  const unsigned hub = ctx.options().hub_threshold ? ctx.options().hub_threshold : 1u;
  unsigned long long hubs = n_in;
  if (work_bound != ~0ull) {
    if (work_bound / hub < hubs)
      hubs = work_bound / hub;
  } else {
    work_bound = (unsigned long long)G.get_number_of_edges() + (unsigned long long)n_in * hub;
  }
  capacity = work_bound / chunk_edges + hubs + 1024;
  if (ctx.options().chunk_queue_limit && capacity > ctx.options().chunk_queue_limit)
    capacity = ctx.options().chunk_queue_limit;
  return reinterpret_cast<k::chunk_t<vertex_t, edge_t>*>(
      ctx.workspace().queue(capacity * sizeof(k::chunk_t<vertex_t, edge_t>)));
}

}  // namespace detail

// ===========================================================================
// block_mapped (and work_stealing = the same kernel with dynamic tile claims)
// ===========================================================================
namespace block_mapped {

template <advance_direction_t direction,
          advance_io_type_t input_type,
          advance_io_type_t output_type,
          bool dynamic_tiles = false,
          typename graph_t,
          typename operator_t,
          typename frontier_t>
void execute(graph_t& G,
             operator_t op,
             frontier_t& input,
             frontier_t& output,
             gcuda::standard_context_t& context) {
  namespace k = detail::k;
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  constexpr bool has_out = (output_type != advance_io_type_t::none);

  const std::size_t n_in = (input_type == advance_io_type_t::graph)
                               ? (std::size_t)G.get_number_of_vertices()
                               : input.get_number_of_elements();
  if (n_in == 0) {
    if (has_out)
      output.set_number_of_elements(0);
    return;
  }
  const bool holes = has_out && context.options().holes_layout;
  const unsigned long long max_deg = detail::max_degree(G, context);
  unsigned long long total = ~0ull;
  if (has_out) {
    if (!detail::size_output<input_type>(G, input, output, n_in, holes, total, context))
      return;
  }

  unsigned long long chunk_capacity = 0;
  unsigned long long work_bound = total;
  if (work_bound == ~0ull)
    work_bound = (input_type == advance_io_type_t::graph) ? (unsigned long long)G.get_number_of_edges()
                                                          : input.work_hint();
  auto* chunks = detail::chunk_queue<vertex_t, edge_t>(G, n_in, work_bound, chunk_capacity, context);
  const unsigned hub_threshold = context.options().hub_threshold;
  const unsigned chunk_edges = context.options().chunk_edges ? context.options().chunk_edges : 1024u;
  unsigned long long* counters = context.workspace().counters();
  const unsigned persistent = (unsigned)context.compute_units() * context.options().tile_blocks_per_cu;
  const unsigned tile_width = detail::tile_width_for(n_in, persistent, context);
  const std::size_t n_tiles = (n_in + tile_width - 1) / tile_width;
  const unsigned grid = (unsigned)(n_tiles < persistent ? n_tiles : persistent);
  vertex_t* out_ptr = has_out ? output.data() : nullptr;
  const std::size_t capacity = has_out ? output.get_capacity() : 0;

  detail::clocked_t clock(context);
  // wide frontiers: hub pre-pass + ONE kernel that expands tiles and then claims hub chunks
  // dynamically (advance_kernels.hxx: classify_hubs_kernel / expand_fused_kernel)
  const std::size_t fused_from = context.options().fused_min_slots;
  bool use_settled = false;
  if constexpr (settled_traits<operator_t>::value)
    use_settled = ((op.settled.bits && op.settled.limit > 0) || operator_t::has_predicate) &&
                  context.options().settled_filter && work_bound != ~0ull &&
                  work_bound >= context.options().settled_min_work;
  if (!holes && !dynamic_tiles && ((fused_from && n_in >= fused_from) || use_settled)) {
    // scratch: [8 claim cursors, one 128-B line each | hub mask, one bit per input slot]
    auto* cursors = reinterpret_cast<unsigned long long*>(context.workspace().scratch(
        (8 * k::CLAIM_LINE + (n_in + 63) / 64) * sizeof(unsigned long long)));
    auto* mask = cursors + 8 * k::CLAIM_LINE;
    k::classify_hubs_kernel<input_type>
        <<<detail::grid_for(n_in, k::CLASSIFY_TILE, (unsigned)context.compute_units() * 8u), k::ADV_BLOCK,
           0, context.stream()>>>(G, input.data(), n_in, nullptr, chunks, chunk_capacity, hub_threshold,
                                  chunk_edges, mask, cursors, counters);
    bool expanded = false;
    if constexpr (settled_traits<operator_t>::value) {
      // the client named settled destinations: one 1024-thread workgroup per CU with the bitmap
      // in its LDS (advance_kernels.hxx: expand_settled_kernel)
      if (use_settled) {
        auto kernel = k::expand_settled_kernel<input_type, output_type, graph_t, operator_t, vertex_t, edge_t>;
        if (!op.settled.bits)  // a predicate only: no image to keep in LDS
          op.settled.limit = 0;
        const std::size_t lds = op.settled.limit > 0 ? op.lds_bytes() : 16;
        // dynamic LDS beyond 64 KB needs an opt-in, per device and instantiation; the kernel's static
        // LDS + the image must fit the CU (a wide edge_t makes the static part larger): if not, the
        // fused form below expands this level with the functor called for every edge
        struct fit_t {
          std::size_t static_bytes = 0, allowed = 0;
          bool known = false;
        };
        static fit_t fits[64];  // by device ordinal
        fit_t& fit = fits[context.ordinal() & 63];
        if (!fit.known) {
          hipFuncAttributes fa;
          GRX_HIP_CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)));
          fit.static_bytes = fa.sharedSizeBytes;
          fit.known = true;
        }
        const std::size_t cu_lds = 160u << 10;
        if (fit.static_bytes + lds <= cu_lds) {
          if (fit.allowed < lds) {
            GRX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)(cu_lds - fit.static_bytes)));
            fit.allowed = cu_lds - fit.static_bytes;
          }
          kernel<<<(unsigned)context.compute_units(), k::SET_BLOCK, lds, context.stream()>>>(
              G, op, input.data(), n_in, out_ptr, capacity, counters, chunks, chunk_capacity, mask, cursors,
              (const unsigned long long*)nullptr,
              input_type == advance_io_type_t::vertices && input.ascending() ? detail::dealt_mode() : 0);
          GRX_HIP_CHECK(hipGetLastError());
          expanded = true;
        }
      }
    }
    if (!expanded) {
      const unsigned fgrid = (unsigned)context.compute_units() * context.options().fused_blocks_per_cu;
      k::expand_fused_kernel<input_type, output_type><<<fgrid, k::ADV_BLOCK, 0, context.stream()>>>(
          G, op, input.data(), n_in, nullptr, out_ptr, capacity, counters, chunks, chunk_capacity, mask,
          cursors, input_type == advance_io_type_t::vertices && input.ascending() && detail::dealt_mode() ? 1 : 0);
    }
  } else if (holes) {
    k::block_mapped_kernel<true, dynamic_tiles, input_type, output_type>
        <<<grid, k::ADV_BLOCK, 0, context.stream()>>>(G, op, input.data(), n_in, out_ptr, capacity,
                                                      counters, chunks, chunk_capacity,
                                                      hub_threshold, chunk_edges, nullptr, tile_width);
  } else {
    k::block_mapped_kernel<false, dynamic_tiles, input_type, output_type>
        <<<grid, k::ADV_BLOCK, 0, context.stream()>>>(G, op, input.data(), n_in, out_ptr, capacity,
                                                      counters, chunks, chunk_capacity,
                                                      hub_threshold, chunk_edges, nullptr, tile_width);
    (void)max_deg;  // sizing only: whether hub chunks were queued is the DEVICE's knowledge, and a
                    // remembered max degree (keyed by address) must never decide if they are expanded
    {
      const unsigned chunk_grid =
          (unsigned)context.compute_units() * context.options().chunk_blocks_per_cu;
      if (context.options().wave_chunks)
        k::wave_chunk_kernel<output_type><<<chunk_grid, k::ADV_BLOCK, 0, context.stream()>>>(
            G, op, chunks, chunk_capacity, out_ptr, capacity, counters);
      else
        k::chunk_kernel<output_type><<<chunk_grid, k::ADV_BLOCK, 0, context.stream()>>>(
            G, op, chunks, chunk_capacity, out_ptr, capacity, counters);
    }
  }
  GRX_HIP_CHECK(hipGetLastError());
  if (!has_out && context.options().defer_sync_of_none_output)
    return;  // enqueue only: the operator that follows fetches the counters and stops the clock
  clock.stop();
  if (has_out)
    detail::finish_output(output, holes, total, context);
  else {
    detail::fetch_counters(context);  // waits for the kernels and leaves the counters clean
    context.kernel_clock().collect();
  }
#ifdef GRX_SETTLED_STATS
  if (use_settled) {
    const unsigned long long* m = context.workspace().mirror();
    std::fprintf(stderr, "[settled] slots %zu work %llu: rounds %llu, predicate tests %llu, functor calls %llu "
                 "carrying %llu edges\n", n_in, work_bound, m[14], m[13], m[10], m[11]);
  }
#endif
}

/**
 * @brief Enqueue-only form for fused pipelines (vertex-partitioned supersteps): packed output,
 * frontier length read from DEVICE memory (`n_in_device`, at most `n_in_bound`), nothing fetched,
 * nothing awaited.  The caller reads counters[C_OUT] / the counters' hand-off later.
 */
template <typename graph_t, typename operator_t, typename vertex_t>
void enqueue_packed(graph_t& G,
                    operator_t op,
                    const vertex_t* input,
                    std::size_t n_in_bound,
                    const unsigned long long* n_in_device,
                    unsigned long long work_bound,
                    vertex_t* output,
                    std::size_t capacity,
                    gcuda::standard_context_t& context) {
  namespace k = detail::k;
  using edge_t = typename graph_t::edge_type;
  constexpr advance_io_type_t vin = advance_io_type_t::vertices;
  if (n_in_bound == 0)
    return;
  const unsigned long long max_deg = detail::max_degree(G, context);
  unsigned long long chunk_capacity = 0;
  auto* chunks = detail::chunk_queue<vertex_t, edge_t>(G, n_in_bound, work_bound, chunk_capacity, context);
  const unsigned hub_threshold = context.options().hub_threshold;
  const unsigned chunk_edges = context.options().chunk_edges ? context.options().chunk_edges : 1024u;
  unsigned long long* counters = context.workspace().counters();
  const std::size_t n_tiles = (n_in_bound + k::ADV_BLOCK - 1) / k::ADV_BLOCK;
  const unsigned persistent = (unsigned)context.compute_units() * context.options().tile_blocks_per_cu;
  const unsigned grid = (unsigned)(n_tiles < persistent ? n_tiles : persistent);
  k::block_mapped_kernel<false, false, vin, vin><<<grid, k::ADV_BLOCK, 0, context.stream()>>>(
      G, op, input, n_in_bound, output, capacity, counters, chunks, chunk_capacity, hub_threshold,
      chunk_edges, n_in_device);
  (void)max_deg;
  k::chunk_kernel<vin><<<(unsigned)context.compute_units() * context.options().chunk_blocks_per_cu,
                           k::ADV_BLOCK, 0, context.stream()>>>(G, op, chunks, chunk_capacity, output,
                                                                capacity, counters);
  GRX_HIP_CHECK(hipGetLastError());
}

/**
 * @brief enqueue_packed for a WIDE frontier of a client that named its settled destinations
 * (operators/settled.hxx): hub pre-pass + expand_settled_kernel, frontier length read on the device,
 * nothing fetched, nothing awaited.  Returns false (nothing enqueued) when the kernel's LDS image
 * does not fit this device: the caller falls back to enqueue_packed.
 */
template <typename graph_t, typename operator_t, typename vertex_t>
bool enqueue_packed_settled(graph_t& G,
                            operator_t op,
                            const vertex_t* input,
                            std::size_t n_in_bound,
                            const unsigned long long* n_in_device,
                            unsigned long long work_bound,
                            vertex_t* output,
                            std::size_t capacity,
                            gcuda::standard_context_t& context) {
  namespace k = detail::k;
  using edge_t = typename graph_t::edge_type;
  constexpr advance_io_type_t vin = advance_io_type_t::vertices;
  static_assert(settled_traits<operator_t>::value, "enqueue_packed_settled takes a hinted functor");
  if (n_in_bound == 0)
    return true;
  auto kernel = k::expand_settled_kernel<vin, vin, graph_t, operator_t, vertex_t, edge_t>;
  if (!op.settled.bits)
    op.settled.limit = 0;
  const std::size_t lds = op.settled.limit > 0 ? op.lds_bytes() : 16;
  hipFuncAttributes fa;
  GRX_HIP_CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)));
  const std::size_t cu_lds = 160u << 10;
  if (fa.sharedSizeBytes + lds > cu_lds)
    return false;
  GRX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(cu_lds - fa.sharedSizeBytes)));
  (void)detail::max_degree(G, context);
  unsigned long long chunk_capacity = 0;
  auto* chunks = detail::chunk_queue<vertex_t, edge_t>(G, n_in_bound, work_bound, chunk_capacity, context);
  const unsigned hub_threshold = context.options().hub_threshold;
  const unsigned chunk_edges = context.options().chunk_edges ? context.options().chunk_edges : 1024u;
  unsigned long long* counters = context.workspace().counters();
  auto* cursors = reinterpret_cast<unsigned long long*>(context.workspace().scratch(
      (8 * k::CLAIM_LINE + (n_in_bound + 63) / 64) * sizeof(unsigned long long)));
  auto* mask = cursors + 8 * k::CLAIM_LINE;
  k::classify_hubs_kernel<vin>
      <<<detail::grid_for(n_in_bound, k::CLASSIFY_TILE, (unsigned)context.compute_units() * 8u), k::ADV_BLOCK, 0,
         context.stream()>>>(G, input, n_in_bound, n_in_device, chunks, chunk_capacity, hub_threshold,
                             chunk_edges, mask, cursors, counters);
  kernel<<<(unsigned)context.compute_units(), k::SET_BLOCK, lds, context.stream()>>>(
      G, op, input, n_in_bound, output, capacity, counters, chunks, chunk_capacity, mask, cursors, n_in_device,
      0);
  GRX_HIP_CHECK(hipGetLastError());
  return true;
}

}  // namespace block_mapped

// ===========================================================================
// merge_path: device-wide degree scan + equal shares of edges
// ===========================================================================
namespace merge_path {

/// segments[0..n_in] = exclusive scan of the input slots' degrees; returns the total.
/// (reference advance/helpers.hxx:38-96, compute_output_offsets)
template <advance_io_type_t input_type, typename graph_t, typename vertex_t, typename work_tiles_t>
unsigned long long scan_degrees(graph_t& G, const vertex_t* input, std::size_t n_in,
                                work_tiles_t& segments, gcuda::standard_context_t& context) {
  namespace k = detail::k;
  using edge_t = typename graph_t::edge_type;
  if (segments.size() < n_in + 1)
    segments.resize(n_in + 1);
  edge_t* seg = segments.data();
  k::slot_degree_kernel<input_type><<<detail::grid_for(n_in + 1, k::ADV_BLOCK, 4096),
                                      k::ADV_BLOCK, 0, context.stream()>>>(G, input, n_in, seg);
  GRX_HIP_CHECK(hipGetLastError());
  std::size_t bytes = hip::exclusive_sum_temp_bytes(seg, seg, edge_t(0), n_in + 1);
  void* temp = context.workspace().scratch(bytes);
  hip::exclusive_sum(temp, bytes, seg, seg, edge_t(0), n_in + 1, context.stream());
  auto& ws = context.workspace();
  edge_t* landing = reinterpret_cast<edge_t*>(ws.mirror() + 24);
  GRX_HIP_CHECK(hipMemcpyAsync(landing, seg + n_in, sizeof(edge_t), hipMemcpyDeviceToHost,
                               context.stream()));
  context.synchronize();
  return (unsigned long long)*landing;
}

template <advance_direction_t direction,
          advance_io_type_t input_type,
          advance_io_type_t output_type,
          typename graph_t,
          typename operator_t,
          typename frontier_t,
          typename work_tiles_t>
void execute(graph_t& G,
             operator_t op,
             frontier_t& input,
             frontier_t& output,
             work_tiles_t& segments,
             gcuda::standard_context_t& context) {
  namespace k = detail::k;
  using vertex_t = typename graph_t::vertex_type;
  constexpr bool has_out = (output_type != advance_io_type_t::none);
  const std::size_t n_in = (input_type == advance_io_type_t::graph)
                               ? (std::size_t)G.get_number_of_vertices()
                               : input.get_number_of_elements();
  if (n_in == 0) {
    if (has_out)
      output.set_number_of_elements(0);
    return;
  }
  const bool holes = has_out && context.options().holes_layout;
  const unsigned long long total = scan_degrees<input_type>(G, input.data(), n_in, segments, context);
  if (total == 0) {
    if (has_out)
      output.set_number_of_elements(0);
    return;
  }
  if (has_out && output.get_capacity() < total)
    output.reserve(total);
  vertex_t* out_ptr = has_out ? output.data() : nullptr;
  const std::size_t capacity = has_out ? output.get_capacity() : 0;
  const unsigned grid = detail::grid_for(total, k::MP_TILE, (unsigned)context.compute_units() * 8u);
  unsigned long long* counters = context.workspace().counters();
  detail::clocked_t clock(context);
  if (holes)
    k::merge_path_kernel<true, input_type, output_type><<<grid, k::ADV_BLOCK, 0, context.stream()>>>(
        G, op, input.data(), n_in, segments.data(), total, out_ptr, capacity, counters);
  else
    k::merge_path_kernel<false, input_type, output_type><<<grid, k::ADV_BLOCK, 0, context.stream()>>>(
        G, op, input.data(), n_in, segments.data(), total, out_ptr, capacity, counters);
  GRX_HIP_CHECK(hipGetLastError());
  clock.stop();
  if (has_out)
    detail::finish_output(output, holes, total, context);
  else {
    detail::fetch_counters(context);  // waits for the kernels and leaves the counters clean
    context.kernel_clock().collect();
  }
}

}  // namespace merge_path

// ===========================================================================
// thread_mapped / warp_mapped
// ===========================================================================
namespace thread_mapped {

template <advance_direction_t direction,
          advance_io_type_t input_type,
          advance_io_type_t output_type,
          typename graph_t,
          typename operator_t,
          typename frontier_t,
          typename work_tiles_t>
void execute(graph_t& G,
             operator_t op,
             frontier_t& input,
             frontier_t& output,
             work_tiles_t& segments,
             gcuda::standard_context_t& context) {
  namespace k = detail::k;
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  constexpr bool has_out = (output_type != advance_io_type_t::none);
  const std::size_t n_in = (input_type == advance_io_type_t::graph)
                               ? (std::size_t)G.get_number_of_vertices()
                               : input.get_number_of_elements();
  if (n_in == 0) {
    if (has_out)
      output.set_number_of_elements(0);
    return;
  }
  const bool holes = has_out && context.options().holes_layout;
  unsigned long long total = ~0ull;
  const edge_t* seg = nullptr;
  if (holes) {
    total = merge_path::scan_degrees<input_type>(G, input.data(), n_in, segments, context);
    if (total == 0) {
      output.set_number_of_elements(0);
      return;
    }
    if (output.get_capacity() < total)
      output.reserve(total);
    seg = segments.data();
  } else if (has_out) {
    if (!detail::size_output<input_type>(G, input, output, n_in, false, total, context))
      return;
  }
  vertex_t* out_ptr = has_out ? output.data() : nullptr;
  const std::size_t capacity = has_out ? output.get_capacity() : 0;
  const unsigned grid = detail::grid_for(n_in, k::ADV_BLOCK, (unsigned)context.compute_units() * 8u);
  unsigned long long* counters = context.workspace().counters();
  detail::clocked_t clock(context);
  if (holes)
    k::thread_mapped_kernel<true, input_type, output_type>
        <<<grid, k::ADV_BLOCK, 0, context.stream()>>>(G, op, input.data(), n_in, seg, out_ptr,
                                                      capacity, counters);
  else
    k::thread_mapped_kernel<false, input_type, output_type>
        <<<grid, k::ADV_BLOCK, 0, context.stream()>>>(G, op, input.data(), n_in, seg, out_ptr,
                                                      capacity, counters);
  GRX_HIP_CHECK(hipGetLastError());
  clock.stop();
  if (has_out)
    detail::finish_output(output, holes, total, context);
  else {
    detail::fetch_counters(context);  // waits for the kernels and leaves the counters clean
    context.kernel_clock().collect();
  }
}

}  // namespace thread_mapped

namespace warp_mapped {

template <advance_direction_t direction,
          advance_io_type_t input_type,
          advance_io_type_t output_type,
          typename graph_t,
          typename operator_t,
          typename frontier_t>
void execute(graph_t& G,
             operator_t op,
             frontier_t& input,
             frontier_t& output,
             gcuda::standard_context_t& context) {
  namespace k = detail::k;
  using vertex_t = typename graph_t::vertex_type;
  constexpr bool has_out = (output_type != advance_io_type_t::none);
  const std::size_t n_in = (input_type == advance_io_type_t::graph)
                               ? (std::size_t)G.get_number_of_vertices()
                               : input.get_number_of_elements();
  if (n_in == 0) {
    if (has_out)
      output.set_number_of_elements(0);
    return;
  }
  unsigned long long total = ~0ull;
  if (has_out) {
    if (!detail::size_output<input_type>(G, input, output, n_in, false, total, context))
      return;
  }
  vertex_t* out_ptr = has_out ? output.data() : nullptr;
  const std::size_t capacity = has_out ? output.get_capacity() : 0;
  const unsigned grid = detail::grid_for(n_in, k::ADV_WAVES, (unsigned)context.compute_units() * 8u);
  detail::clocked_t clock(context);
  k::wave_mapped_kernel<input_type, output_type><<<grid, k::ADV_BLOCK, 0, context.stream()>>>(
      G, op, input.data(), n_in, out_ptr, capacity, context.workspace().counters());
  GRX_HIP_CHECK(hipGetLastError());
  clock.stop();
  if (has_out)
    detail::finish_output(output, false, total, context);
  else {
    detail::fetch_counters(context);  // waits for the kernels and leaves the counters clean
    context.kernel_clock().collect();
  }
}

}  // namespace warp_mapped

// ===========================================================================
// bucketing: thread / wavefront / chunk schedules by degree class
// ===========================================================================
namespace bucketing {

template <advance_direction_t direction,
          advance_io_type_t input_type,
          advance_io_type_t output_type,
          typename graph_t,
          typename operator_t,
          typename frontier_t>
void execute(graph_t& G,
             operator_t op,
             frontier_t& input,
             frontier_t& output,
             gcuda::standard_context_t& context) {
  namespace k = detail::k;
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  constexpr bool has_out = (output_type != advance_io_type_t::none);
  constexpr advance_io_type_t vin = advance_io_type_t::vertices;
  const std::size_t n_in = (input_type == advance_io_type_t::graph)
                               ? (std::size_t)G.get_number_of_vertices()
                               : input.get_number_of_elements();
  if (n_in == 0) {
    if (has_out)
      output.set_number_of_elements(0);
    return;
  }
  unsigned long long total = ~0ull;
  if (has_out) {
    if (!detail::size_output<input_type>(G, input, output, n_in, false, total, context))
      return;
  }
  auto& ws = context.workspace();
  unsigned long long chunk_capacity = 0;
  unsigned long long work_bound = total;
  if (work_bound == ~0ull)
    work_bound = (input_type == advance_io_type_t::graph) ? (unsigned long long)G.get_number_of_edges()
                                                          : input.work_hint();
  auto* chunks = detail::chunk_queue<vertex_t, edge_t>(G, n_in, work_bound, chunk_capacity, context);
  vertex_t* bins = reinterpret_cast<vertex_t*>(ws.scratch(2 * n_in * sizeof(vertex_t)));
  vertex_t* small_q = bins;
  vertex_t* medium_q = bins + n_in;
  unsigned long long* counters = ws.counters();
  const unsigned hub_threshold = context.options().hub_threshold;
  const unsigned chunk_edges = context.options().chunk_edges ? context.options().chunk_edges : 1024u;
  const unsigned persistent = (unsigned)context.compute_units() * 8u;

  detail::clocked_t clock(context);
  k::bucket_kernel<input_type><<<detail::grid_for(n_in, k::ADV_BLOCK, (unsigned)context.compute_units() * 8u),
                                 k::ADV_BLOCK, 0, context.stream()>>>(
      G, input.data(), n_in, small_q, medium_q, chunks, chunk_capacity, hub_threshold, chunk_edges,
      counters);
  GRX_HIP_CHECK(hipGetLastError());
  unsigned long long* m = detail::fetch_counters(context);
  const std::size_t n_small = (std::size_t)m[k::C_BUCKET0];
  const std::size_t n_medium = (std::size_t)m[k::C_BUCKET0 + 1];
  const unsigned long long n_chunks = m[k::C_CHUNKS];

  vertex_t* out_ptr = has_out ? output.data() : nullptr;
  const std::size_t capacity = has_out ? output.get_capacity() : 0;
  if (n_small)
    k::thread_mapped_kernel<false, vin, output_type>
        <<<detail::grid_for(n_small, k::ADV_BLOCK, persistent), k::ADV_BLOCK, 0, context.stream()>>>(
            G, op, small_q, n_small, (const edge_t*)nullptr, out_ptr, capacity, counters);
  if (n_medium)
    k::wave_mapped_kernel<vin, output_type>
        <<<detail::grid_for(n_medium, k::ADV_WAVES, persistent), k::ADV_BLOCK, 0, context.stream()>>>(
            G, op, medium_q, n_medium, out_ptr, capacity, counters);
  if (n_chunks)
    k::chunk_kernel<output_type>
        <<<(unsigned)context.compute_units() * context.options().chunk_blocks_per_cu, k::ADV_BLOCK, 0,
           context.stream()>>>(G, op, chunks, chunk_capacity, out_ptr, capacity, counters,
                               (long long)n_chunks);
  GRX_HIP_CHECK(hipGetLastError());
  clock.stop();
  if (has_out)
    detail::finish_output(output, false, total, context);
  else {
    detail::fetch_counters(context);  // waits for the kernels and leaves the counters clean
    context.kernel_clock().collect();
  }
}

}  // namespace bucketing

// ===========================================================================
// pull (advance_direction_t::backward): candidates in, newly hit candidates out
// ===========================================================================
namespace pull {

/**
 * @brief Pull advance.  `input` holds candidate DESTINATION vertices; for each one its
 * in-edges are walked (CSR view of an undirected graph = its transpose) and
 * op(in_neighbour, candidate, edge, weight) is called until it returns true; such
 * candidates are written, packed, to `output`.  The op is called at most once per in-edge
 * and never again for a candidate after its first true.  With an attached transpose the edge
 * id handed to the op is the position in the TRANSPOSED arrays (transposed_t::edge_ids maps it
 * back to the CSR edge).  The reference declares this
 * direction but throws for it (advance_direction_t::backward / optimized,
 * framework/operators/configs.hxx:58-62, advance/merge_path.hxx:41-56).
 *
 * `rejected` (optional): receives, packed, the candidates with at least one in-edge for which
 * the op never returned true -- the candidate list of the next pull level, so that a caller
 * alternating pull levels needs no separate compaction pass.
 */
template <advance_io_type_t input_type,
          advance_io_type_t output_type,
          typename graph_t,
          typename operator_t,
          typename frontier_t>
void execute(graph_t& G,
             operator_t op,
             frontier_t& input,
             frontier_t& output,
             gcuda::standard_context_t& context,
             frontier_t* rejected = nullptr) {
  namespace k = detail::k;
  using vertex_t = typename graph_t::vertex_type;
  constexpr bool has_out = (output_type != advance_io_type_t::none);
  error::throw_if_exception(!G.can_pull(),
                            "pull advance needs in-edges: the graph is marked directed and has no "
                            "attached transpose (graph::build::transpose(G, ctx).attach_to(G), or "
                            "G.properties.directed = false for a symmetric CSR)");
  auto Gin = G.in_edges();  // the graph whose out-edges are G's in-edges
  error::throw_if_exception(input_type != advance_io_type_t::vertices,
                            "pull advance takes a vertex frontier of candidates");
  const std::size_t n_in = input.get_number_of_elements();
  if (n_in == 0) {
    if (has_out)
      output.set_number_of_elements(0);
    return;
  }
  if (has_out && output.get_capacity() < n_in)
    output.reserve(n_in);  // at most every candidate is emitted once
  if (rejected) {
    error::throw_if_exception(rejected->data() == input.data() && rejected->data() != nullptr,
                              "pull advance: `rejected` must not alias the candidates");
    if (rejected->get_capacity() < n_in)
      rejected->reserve(n_in);
  }
  auto& ws = context.workspace();
  unsigned long long* counters = ws.counters();
  auto* long_queue = reinterpret_cast<k::resume_t<vertex_t>*>(
      ws.queue(n_in * sizeof(k::resume_t<vertex_t>)));
  vertex_t* out_ptr = has_out ? output.data() : nullptr;
  const std::size_t capacity = has_out ? output.get_capacity() : 0;
  const unsigned persistent = (unsigned)context.compute_units() * 8u;
  detail::clocked_t clock(context);
  const unsigned probe_grid = detail::grid_for(n_in, k::ADV_BLOCK, persistent);
  const unsigned long_grid = (unsigned)context.compute_units() * 4u;
  // a separate in-edge view: the emitted vertices' OUT-degrees (the next push's work) come from
  // the forward offsets; an undirected CSR is its own transpose and needs no second lookup
  const auto* fwd = G.has_in_edges() ? G.get_row_offsets() : nullptr;
  if (rejected) {
    k::pull_probe_kernel<output_type, true><<<probe_grid, k::ADV_BLOCK, 0, context.stream()>>>(
        Gin, op, input.data(), n_in, out_ptr, capacity, rejected->data(), long_queue,
        (unsigned long long)n_in, counters, fwd);
    k::pull_long_kernel<output_type, true><<<long_grid, k::ADV_BLOCK, 0, context.stream()>>>(
        Gin, op, long_queue, (unsigned long long)n_in, out_ptr, capacity, rejected->data(),
        (unsigned long long)n_in, counters, fwd);
  } else {
    k::pull_probe_kernel<output_type, false><<<probe_grid, k::ADV_BLOCK, 0, context.stream()>>>(
        Gin, op, input.data(), n_in, out_ptr, capacity, (vertex_t*)nullptr, long_queue,
        (unsigned long long)n_in, counters, fwd);
    k::pull_long_kernel<output_type, false><<<long_grid, k::ADV_BLOCK, 0, context.stream()>>>(
        Gin, op, long_queue, (unsigned long long)n_in, out_ptr, capacity, (vertex_t*)nullptr, 0ull,
        counters, fwd);
  }
  GRX_HIP_CHECK(hipGetLastError());
  clock.stop();
  if (has_out)
    detail::finish_output(output, false, ~0ull, context);
  else {
    detail::fetch_counters(context);  // waits for the kernels and leaves the counters clean
    context.kernel_clock().collect();
  }
  if (rejected) {
    rejected->set_number_of_elements((std::size_t)context.workspace().mirror()[k::C_BUCKET0]);
    error::throw_if_exception(context.workspace().mirror()[k::C_OVERFLOW] != 0,
                              "pull advance: rejected list overflow");
  }
}

}  // namespace pull

// ===========================================================================
// dispatch
// ===========================================================================

/**
 * @brief Frontier-level entry (reference advance.hxx:91-129).
 */
template <load_balance_t lb,
          advance_direction_t direction,
          advance_io_type_t input_type,
          advance_io_type_t output_type,
          typename graph_t,
          typename operator_t,
          typename frontier_t,
          typename work_tiles_t>
void execute(graph_t& G,
             operator_t op,
             frontier_t* input,
             frontier_t* output,
             work_tiles_t& segments,
             gcuda::multi_context_t& context) {
  error::throw_if_exception(context.size() != 1, "`context.size() != 1` not supported");
  error::throw_if_exception(direction == advance_direction_t::optimized,
                            "advance: the push/pull choice is the client's (see the direction-"
                            "optimising BFS in essentials_amd/csrc/clients.hxx); ask for forward or backward");
  if constexpr (direction == advance_direction_t::backward) {
    pull::execute<input_type, output_type>(G, op, *input, *output, *context.get_context(0));
    return;
  }
  error::throw_if_exception(input_type == advance_io_type_t::edges ||
                                output_type == advance_io_type_t::edges ||
                                output_type == advance_io_type_t::graph,
                            "Advance type not supported.");
  auto& ctx = *context.get_context(0);
  constexpr load_balance_t schedule = GRX_LB_EFFECTIVE(lb);
  // every edge of the graph, nothing written: the order of the calls is the engine's choice, and
  // grouped by destination the functor's atomics combine (operators/by_destination.hxx)
  if constexpr (input_type == advance_io_type_t::graph && output_type == advance_io_type_t::none) {
    if (const void* items =
            by_destination::prepared(G, ctx.workspace().by_destination().current_run, ctx)) {
      detail::clocked_t clock(ctx);
      by_destination::enqueue(G, items, op, ctx);
      if (ctx.options().defer_sync_of_none_output)
        return;
      clock.stop();
      detail::fetch_counters(ctx);  // waits for the kernel
      ctx.kernel_clock().collect();
      return;
    }
  }
  // deterministic output positions exist only for merge_path / thread_mapped / block_mapped
  const bool holes = ctx.options().holes_layout && output_type != advance_io_type_t::none;

  if constexpr (schedule == load_balance_t::block_mapped) {
    block_mapped::execute<direction, input_type, output_type, false>(G, op, *input, *output, ctx);
  } else if constexpr (schedule == load_balance_t::work_stealing) {
    block_mapped::execute<direction, input_type, output_type, true>(G, op, *input, *output, ctx);
  } else if constexpr (schedule == load_balance_t::merge_path ||
                       schedule == load_balance_t::merge_path_v2) {
    merge_path::execute<direction, input_type, output_type>(G, op, *input, *output, segments, ctx);
  } else if constexpr (schedule == load_balance_t::thread_mapped) {
    thread_mapped::execute<direction, input_type, output_type>(G, op, *input, *output, segments, ctx);
  } else if constexpr (schedule == load_balance_t::warp_mapped) {
    if (holes)
      merge_path::execute<direction, input_type, output_type>(G, op, *input, *output, segments, ctx);
    else
      warp_mapped::execute<direction, input_type, output_type>(G, op, *input, *output, ctx);
  } else if constexpr (schedule == load_balance_t::bucketing) {
    if (holes)
      merge_path::execute<direction, input_type, output_type>(G, op, *input, *output, segments, ctx);
    else
      bucketing::execute<direction, input_type, output_type>(G, op, *input, *output, ctx);
  } else {
    error::throw_if_exception(true, "Advance type not supported.");
  }
}

/**
 * @brief Enactor-level entry (reference advance.hxx:192-221): uses the enactor's
 * input/output frontiers and scan workspace, then swaps the buffers.
 */
template <load_balance_t lb = load_balance_t::merge_path,
          advance_direction_t direction = advance_direction_t::forward,
          advance_io_type_t input_type = advance_io_type_t::vertices,
          advance_io_type_t output_type = advance_io_type_t::vertices,
          typename graph_t,
          typename enactor_type,
          typename operator_type>
void execute(graph_t& G,
             enactor_type* E,
             operator_type op,
             gcuda::multi_context_t& context,
             bool swap_buffers = true) {
  if constexpr (input_type == advance_io_type_t::graph && output_type == advance_io_type_t::none)
    context.get_context(0)->workspace().by_destination().current_run = E->unique_id;
  execute<lb, direction, input_type, output_type>(G, op, E->get_input_frontier(),
                                                  E->get_output_frontier(),
                                                  E->scanned_work_domain, context);
  if constexpr (input_type == advance_io_type_t::graph && output_type == advance_io_type_t::none)
    context.get_context(0)->workspace().by_destination().current_run = 0;
  if (swap_buffers && (output_type != advance_io_type_t::none))
    E->swap_frontier_buffers();
}

}  // namespace advance
}  // namespace operators
}  // namespace gunrock
