/**
 * @file clients.hxx
 * @brief The build's own BFS / SSSP / PageRank written against the engine's
 * public API -- the conformance clients that travel to the GPU box (the
 * reference's algorithm headers cannot).
 *
 * They exercise exactly the call forms the reference's clients use
 * (SURVEY.md 8b): problem_t / enactor_t subclasses, prepare_frontier ->
 * push_back, operators::advance::execute<lb>(G, E, op, context) with a
 * bool(vertex const&, vertex const&, edge const&, weight const&) functor,
 * operators::filter::execute<bypass>(G, E, pred, context), the 4-argument
 * <lb, forward, graph, none> advance, math::atomic::{min,add}, thread::load,
 * enactor_properties_t::self_manage_frontiers, enact() -> ms.
 * Algorithms: reference algorithms/bfs.hxx:80-132, sssp.hxx:98-151,
 * pr.hxx:64-178.  Independently written; the schedule is a template parameter.
 */
#pragma once
#include <numeric>

#include <gunrock/hip/kernels/reduce_kernels.hxx>

#include <gunrock/framework/framework.hxx>
#include <gunrock/hip/algorithms.hxx>

namespace essentials_amd {
namespace clients {

using namespace gunrock;
using operators::load_balance_t;

/// Per-run record filled by the enactors (host side, no device cost).
struct level_log_t {
  static constexpr int max_levels = 64;
  int levels = 0;
  long long input_slots[max_levels] = {0};
  long long edges_expanded = 0;   // sum of the input frontiers' work hints (exact when an advance
                                  // produced the frontier; the source's own degree is added by the caller)
  long long slots_total = 0;      // sum of the input frontiers' lengths, all levels
  int unknown_work_levels = 0;    // levels whose frontier carried no work hint (level 0 always)
  void note(std::size_t slots, unsigned long long work = ~0ull) {
    if (levels < max_levels)
      input_slots[levels] = (long long)slots;
    slots_total += (long long)slots;
    if (work != ~0ull)
      edges_expanded += (long long)work;
    else
      ++unknown_work_levels;
    ++levels;
  }
};

// ---------------------------------------------------------------------------
// breadth-first search: depth[v] = hops from the source, INT_MAX if unreachable
// ---------------------------------------------------------------------------
template <typename graph_t>
struct bfs_problem_t : gunrock::problem_t<graph_t> {
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;

  vertex_t source;
  vertex_t* depth;  // device, |V|, caller-owned
  level_log_t log;
  // Push search on graphs whose 4-byte depths outgrow the caches: ONE BYTE per vertex while the
  // search runs (0xFF = unvisited), four to a word; a vertex is claimed by clearing bits of its
  // byte (atomic AND -- one RMW per discovery, like the atomic min on a word), the caller's array is
  // written when the run ends or when level 254 is reached (the search then goes on in it).
  bool byte_labels = false;
  hip::device_array_t<unsigned> bytes;
  // The graph is a renumbered copy of the caller's (graph::build::hot_first): `source` is in the
  // graph's numbering, the search runs in an array of its own and deliver() hands every label to
  // the caller's array in the caller's numbering, depth[scatter_to[v]] = label of v.
  const vertex_t* scatter_to = nullptr;
  // the inverse permutation, when the owner has it on the device: labels are then delivered as
  // depth[v] = label[gather_from[v]] -- coalesced stores, and the lookups of 64 consecutive v fall
  // into few lines (vertices of one degree keep their order in both numberings, so each degree
  // class is an ascending stream); the scatter writes 64 different lines per wavefront (55 -> 20 us
  // for the 4.2 M labels of R-MAT-22)
  const vertex_t* gather_from = nullptr;
  hip::device_array_t<vertex_t> own_labels;

  bfs_problem_t(graph_t& G, vertex_t _source, vertex_t* _depth,
                std::shared_ptr<gcuda::multi_context_t> ctx)
      : gunrock::problem_t<graph_t>(G, ctx), source(_source), depth(_depth) {}

  /// The 4-byte array the search runs in (the graph's numbering).
  vertex_t* labels() { return scatter_to ? own_labels.data() : depth; }

  void init() override {
    const std::size_t n = (std::size_t)this->get_graph().get_number_of_vertices();
    if (byte_labels)
      bytes.resize((n + 3) / 4);
    else if (scatter_to)
      own_labels.resize(n);
  }
  void reset() override {
    auto ctx = this->get_single_context();
    const std::size_t n = (std::size_t)this->get_graph().get_number_of_vertices();
    // one pass, enqueued on the context's stream ahead of the traversal: nothing to wait for
    const vertex_t s = source;
    // the source's degree, for the run's statistics: left in pinned memory by the reset pass itself
    auto G = this->get_graph();
    unsigned long long* facts = ctx->workspace().run_facts();
    if (byte_labels) {
      unsigned* w = bytes.data();
      hip::for_each_index(
          (n + 3) / 4, [w, s, G, facts] __device__(std::size_t i) {
            w[i] = (std::size_t)(s >> 2) == i ? ~(0xFFu << ((s & 3) * 8)) : 0xFFFFFFFFu;
            if ((std::size_t)(s >> 2) == i)
              facts[0] = (unsigned long long)G.get_number_of_neighbors(s);
          },
          ctx->stream());
    } else {
      vertex_t* d = labels();
      hip::for_each_index(
          n, [d, s, G, facts] __device__(std::size_t i) {
            d[i] = (vertex_t)i == s ? vertex_t(0) : std::numeric_limits<vertex_t>::max();
            if ((vertex_t)i == s)
              facts[0] = (unsigned long long)G.get_number_of_neighbors(s);
          },
          ctx->stream());
    }
    log = level_log_t();
  }
  /// Byte form: write the 4-byte array (one pass) and continue / end in it.
  void unpack() {
    if (!byte_labels)
      return;
    auto ctx = this->get_single_context();
    const std::size_t n = (std::size_t)this->get_graph().get_number_of_vertices();
    if (scatter_to)
      own_labels.resize(n);
    const unsigned* w = bytes.data();
    vertex_t* d = labels();
    hip::for_each_index(
        n, [w, d] __device__(std::size_t i) {
          const unsigned b = (w[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
          d[i] = b == 0xFFu ? std::numeric_limits<vertex_t>::max() : (vertex_t)b;
        },
        ctx->stream());
    byte_labels = false;
  }
  /// End of a run (inside the timed enact()): every label in the caller's array and numbering.
  void deliver() {
    if (!scatter_to) {
      unpack();
      return;
    }
    auto ctx = this->get_single_context();
    const std::size_t n = (std::size_t)this->get_graph().get_number_of_vertices();
    const vertex_t* to = scatter_to;
    const vertex_t* from = gather_from;
    vertex_t* d = depth;
    if (byte_labels) {
      const unsigned* w = bytes.data();
      if (from)
        hip::for_each_index(
            n, [w, d, from] __device__(std::size_t v) {
              const std::size_t i = (std::size_t)from[v];
              const unsigned b = (w[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
              d[v] = b == 0xFFu ? std::numeric_limits<vertex_t>::max() : (vertex_t)b;
            },
            ctx->stream());
      else
        hip::for_each_index(
            n, [w, d, to] __device__(std::size_t i) {
              const unsigned b = (w[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
              d[to[i]] = b == 0xFFu ? std::numeric_limits<vertex_t>::max() : (vertex_t)b;
            },
            ctx->stream());
      byte_labels = false;
    } else {
      const vertex_t* l = own_labels.data();
      if (from)
        hip::for_each_index(n, [l, d, from] __device__(std::size_t v) { d[v] = l[from[v]]; }, ctx->stream());
      else
        hip::for_each_index(n, [l, d, to] __device__(std::size_t i) { d[to[i]] = l[i]; }, ctx->stream());
    }
  }
};

template <typename problem_type, load_balance_t lb>
struct bfs_enactor_t : gunrock::enactor_t<problem_type> {
  using base_t = gunrock::enactor_t<problem_type>;
  using vertex_t = typename problem_type::vertex_t;
  using edge_t = typename problem_type::edge_t;
  using weight_t = typename problem_type::weight_t;
  using frontier_t = typename base_t::frontier_t;
  int max_iterations = 0;

  bfs_enactor_t(problem_type* p, std::shared_ptr<gcuda::multi_context_t> ctx,
                enactor_properties_t props = enactor_properties_t())
      : base_t(p, ctx, props) {}

  void prepare_frontier(frontier_t* f, gcuda::multi_context_t& context) override {
    f->push_back(this->get_problem()->source, context.get_context(0)->stream());
  }

  bool is_converged(gcuda::multi_context_t& context) override {
    if (max_iterations && this->iteration >= max_iterations)
      return true;
    return base_t::is_converged(context);
  }

  void finalize(gcuda::multi_context_t&) override { this->get_problem()->deliver(); }

  /// One level.  `visit` discovers, `has_depth` is its pure "already discovered" test,
  /// `found_now(v)` says that THIS level discovered v (asked after the level), `mark` is visit for a
  /// level whose output nobody reads: it labels an unlabelled destination and returns nothing of use.
  template <typename visit_t, typename has_depth_t, typename found_t, typename mark_t>
  void expand(visit_t visit, has_depth_t has_depth, found_t found_now, mark_t mark,
              gcuda::multi_context_t& context) {
    auto E = this->get_enactor();
    auto G = this->get_problem()->get_graph();
    // wide levels (block_mapped's fused form): every vertex that has a depth is settled -- visit()
    // returns false for it and changes nothing -- so the engine may answer those lookups from LDS
    // (operators/settled.hxx).  The bitmap is rebuilt from the depths, one small kernel per level.
    auto ctx = context.get_context(0);
    const unsigned long long work = E->get_input_frontier()->work_hint();
    if (lb == load_balance_t::block_mapped && ctx->options().settled_filter &&
        !ctx->options().holes_layout && work != frontier_t::unknown_work &&
        work >= ctx->options().settled_min_work) {
      {  // part of this level's advance: timed with it when kernels are timed
        operators::advance::detail::clocked_t clock(*ctx);
        settled.refresh((std::size_t)G.get_number_of_vertices(), has_depth, *ctx);
        clock.stop();
      }
      if (ctx->options().label_scan_min_work && work >= ctx->options().label_scan_min_work) {
        // the widest levels: no output frontier at all -- the labels say what the level found, and
        // one pass over them builds the next frontier in ascending runs with its degree sum
        // (operators::filter::select_range; the graph's vertices without edges are skipped when a
        // hot-first numbering put them last)
        // Nobody reads this level's output, so nobody needs to know WHO reached a vertex first: every
        // arrival that finds it unlabelled writes the same depth.  `mark` is a plain conditional
        // store where visit() is a fresh look + a read-modify-write at the memory side (27 G/s on this
        // part whatever the scope, tools/scatter_probe.hip) whose round trips the surviving edges'
        // wavefronts wait for: level 1 of RMAT-22 spent 122 of its 396 us in them, 70 of 346 us with
        // the store (calling a store-only functor for the edges the predicate has just passed, without
        // the conditional's load, was measured too: 343 us -- not kept).
        const bool idempotent = ctx->options().settled_filter && mark_without_claim;
        ctx->options().defer_sync_of_none_output = true;  // select_range below is its hand-off
        if (idempotent)
          operators::advance::execute<lb, operators::advance_direction_t::forward,
                                      operators::advance_io_type_t::vertices,
                                      operators::advance_io_type_t::none>(
              G, E, operators::advance::with_settled(mark, settled.view(), has_depth), context);
        else
          operators::advance::execute<lb, operators::advance_direction_t::forward,
                                      operators::advance_io_type_t::vertices,
                                      operators::advance_io_type_t::none>(
              G, E, operators::advance::with_settled(visit, settled.view(), has_depth), context);
        ctx->options().defer_sync_of_none_output = false;
        const std::size_t n_scan = G.properties.leading_connected
                                       ? (std::size_t)G.properties.leading_connected
                                       : (std::size_t)G.get_number_of_vertices();
        // the same pass leaves the NEXT level's settled bitmap (what rebuild() would compute then:
        // has_depth as it stands after this level) when the graph is larger than the image
        std::size_t bit_limit = 0;
        unsigned long long* bit_words = nullptr;
        if ((std::size_t)G.get_number_of_vertices() >= operators::advance::settled_max_ids)
          bit_words = settled.prepare((std::size_t)G.get_number_of_vertices(), bit_limit);
        operators::filter::select_range</*narrow_claims=*/true>(
            G, n_scan, found_now, *E->get_output_frontier(), *ctx, hip::kernels::select_no_each_t(), has_depth,
            bit_words, bit_limit);
        E->swap_frontier_buffers();
        return;
      }
      operators::advance::execute<lb>(
          G, E, operators::advance::with_settled(visit, settled.view(), has_depth), context);
    } else {
      operators::advance::execute<lb>(G, E, visit, context);
    }
  }

  void loop(gcuda::multi_context_t& context) override {
    auto E = this->get_enactor();
    auto P = this->get_problem();
    auto G = P->get_graph();
    P->log.note(E->get_input_frontier()->get_number_of_elements(),
                E->get_input_frontier()->work_hint());

    const vertex_t next_level = this->iteration + 1;
    if (P->byte_labels && next_level >= 255)
      P->unpack();  // a byte cannot hold this level: go on in the 4-byte array
    vertex_t* depth = P->labels();
    if (P->byte_labels) {
      unsigned* words = P->bytes.data();
      const unsigned level = (unsigned)next_level;
      auto visit_byte = [words, level] __device__(vertex_t const& src, vertex_t const& dst,
                                                  edge_t const& edge, weight_t const& weight) -> bool {
        unsigned* w = words + ((unsigned)dst >> 2);
        const unsigned shift = ((unsigned)dst & 3u) * 8u;
        // a fresh (agent-scope) look first: the RMW executes at the memory side
        if (((__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> shift) & 0xFFu) != 0xFFu)
          return false;
        // 0xFF & level = level; a concurrent claim of the same level leaves the same byte
        const unsigned old = ::atomicAnd(w, ~(0xFFu << shift) | (level << shift));
        return ((old >> shift) & 0xFFu) == 0xFFu;
      };
      auto has_byte = [words] __device__(vertex_t const& v) -> bool {
        return ((words[(unsigned)v >> 2] >> (((unsigned)v & 3u) * 8u)) & 0xFFu) != 0xFFu;
      };
      auto found_byte = [words, level] __device__(vertex_t const& v) -> bool {
        return ((words[(unsigned)v >> 2] >> (((unsigned)v & 3u) * 8u)) & 0xFFu) == level;
      };
      unsigned char* bytes = reinterpret_cast<unsigned char*>(words);  // byte v of the array (little endian)
      auto mark_byte = [bytes, level] __device__(vertex_t const& src, vertex_t const& dst, edge_t const& edge,
                                                 weight_t const& weight) -> bool {
        if (bytes[(unsigned)dst] == 0xFFu)
          bytes[(unsigned)dst] = (unsigned char)level;  // a 1-byte store: no neighbour's byte is touched
        return false;
      };
      expand(visit_byte, has_byte, found_byte, mark_byte, context);
      return;
    }
    auto visit = [depth, next_level] __host__ __device__(vertex_t const& src, vertex_t const& dst,
                                                         edge_t const& edge,
                                                         weight_t const& weight) -> bool {
      // first arrival wins; later arrivals of the same level see an equal depth.  (Measured
      // neutral in front of this test: a dense "seen" bitmap, and a batched read-only pre-test of
      // a lane's four in-flight edges -- DESIGN.md section 5.)
      return next_level < math::atomic::min(&depth[dst], next_level);
    };
    auto has_depth = [depth] __device__(vertex_t const& v) -> bool {
      return depth[v] != std::numeric_limits<vertex_t>::max();
    };
    auto found_now = [depth, next_level] __device__(vertex_t const& v) -> bool {
      return depth[v] == next_level;
    };
    auto mark = [depth, next_level] __device__(vertex_t const& src, vertex_t const& dst, edge_t const& edge,
                                               weight_t const& weight) -> bool {
      if (depth[dst] == std::numeric_limits<vertex_t>::max())
        depth[dst] = next_level;
      return false;
    };
    expand(visit, has_depth, found_now, mark, context);
  }
  operators::advance::settled_filter_t<vertex_t> settled;
  bool mark_without_claim = true;  // label-scan levels label with a plain store (GRX_BFS_MARK=0: the claim)
};

// ---------------------------------------------------------------------------
// direction-optimising BFS (Beamer et al.): push while the frontier is small, PULL
// (advance_direction_t::backward) while it is wide, push again for the tail.  Same
// depths as the push-only search; new relative to the reference, whose advance
// throws for the backward / optimized directions (configs.hxx:58-62).
// Needs in-edges: an undirected (symmetric) CSR, or a directed one with an attached transpose
// (graph::build::transpose).
// ---------------------------------------------------------------------------
template <typename problem_type, load_balance_t lb>
struct bfs_do_enactor_t : gunrock::enactor_t<problem_type> {
  using base_t = gunrock::enactor_t<problem_type>;
  using vertex_t = typename problem_type::vertex_t;
  using edge_t = typename problem_type::edge_t;
  using weight_t = typename problem_type::weight_t;
  using frontier_t = typename base_t::frontier_t;
  int max_iterations = 0;
  float alpha = 4.0f;   // pull when frontier edges > unexplored edges / alpha (swept on RMAT-22)
  float beta = 24.0f;   // push again when frontier vertices < |V| / beta
  int pull_iterations = 0;

  frontier_t candidates[2];  // still-unvisited vertices with at least one edge
  int cand = 0;
  bool have_candidates = false;
  bool candidates_current = false;
  bool pulling = false;
  frontier::bitmap_frontier_t<vertex_t> in_frontier;  // dense view of the current frontier
  unsigned long long unexplored = 0;                  // edges out of unvisited vertices

  bfs_do_enactor_t(problem_type* p, std::shared_ptr<gcuda::multi_context_t> ctx,
                   enactor_properties_t props = enactor_properties_t())
      : base_t(p, ctx, props) {
    auto g = p->get_graph();
    in_frontier.resize((std::size_t)g.get_number_of_vertices());
    unexplored = (unsigned long long)g.get_number_of_edges();
  }

  void prepare_frontier(frontier_t* f, gcuda::multi_context_t& context) override {
    f->push_back(this->get_problem()->source, context.get_context(0)->stream());
  }

  bool is_converged(gcuda::multi_context_t& context) override {
    if (max_iterations && this->iteration >= max_iterations)
      return true;
    return base_t::is_converged(context);
  }

  void finalize(gcuda::multi_context_t&) override { this->get_problem()->deliver(); }

  void loop(gcuda::multi_context_t& context) override {
    auto E = this->get_enactor();
    auto P = this->get_problem();
    auto G = P->get_graph();
    auto ctx = context.get_context(0);
    frontier_t* in = E->get_input_frontier();
    P->log.note(in->get_number_of_elements());

    vertex_t* depth = P->labels();
    const vertex_t next_level = this->iteration + 1;
    const std::size_t n_vertices = (std::size_t)G.get_number_of_vertices();
    const std::size_t n_f = in->get_number_of_elements();
    const unsigned long long m_f = in->work_hint();

    if (!pulling) {
      if (m_f != frontier_t::unknown_work && (double)m_f > (double)unexplored / alpha)
        pulling = true;
    } else if ((double)n_f < (double)n_vertices / beta) {
      pulling = false;
    }

    if (!pulling) {
      auto visit = [depth, next_level] __host__ __device__(vertex_t const& src, vertex_t const& dst,
                                                           edge_t const& edge,
                                                           weight_t const& weight) -> bool {
        return next_level < math::atomic::min(&depth[dst], next_level);
      };
      operators::advance::execute<lb>(G, E, visit, context);
      candidates_current = false;  // a push level visits vertices behind the candidates' back
    } else {
      ++pull_iterations;
      // 1. candidates = unvisited vertices that have edges.  Built by one compaction when pulling
      //    starts (or resumes after push levels); afterwards each pull level hands its rejects
      //    over as the next level's candidates.
      if (!candidates_current) {
        auto Gin = G.in_edges();
        auto unvisited = [depth, Gin] __host__ __device__(vertex_t const& u) -> bool {
          return depth[u] == std::numeric_limits<vertex_t>::max() && Gin.get_number_of_neighbors(u) > 0;
        };
        if (!have_candidates) {
          candidates[cand].sequence(vertex_t(0), n_vertices, ctx->stream());
          have_candidates = true;
        }
        operators::filter::execute<operators::filter_algorithm_t::predicated>(
            G, unvisited, &candidates[cand], &candidates[cand ^ 1], context);
        cand ^= 1;
        candidates_current = true;
      }
      // 2. membership bitmap of the current frontier (512 KB at 2^22 vertices: it lives in
      //    every XCD's L2, unlike the 16 MB label array), built by one pass over the labels
      const vertex_t this_level = this->iteration;
      in_frontier.assign_if(
          [depth, this_level] __device__(std::size_t v) { return depth[v] == this_level; }, *ctx);
      const frontier::bitmap_view_t bits = in_frontier.view();
      // 3. every candidate looks for a parent among its in-neighbours
      auto adopt = [bits, depth, next_level] __host__ __device__(vertex_t const& parent,
                                                                 vertex_t const& child,
                                                                 edge_t const& edge,
                                                                 weight_t const& weight) -> bool {
        if (bits.test((std::size_t)parent)) {
          depth[child] = next_level;
          return true;
        }
        return false;
      };
      operators::advance::pull::execute<operators::advance_io_type_t::vertices,
                                        operators::advance_io_type_t::vertices>(
          G, adopt, candidates[cand], *E->get_output_frontier(), *ctx, &candidates[cand ^ 1]);
      cand ^= 1;
      E->swap_frontier_buffers();
    }
    const unsigned long long found = E->get_input_frontier()->work_hint();
    if (found != frontier_t::unknown_work)
      unexplored = found < unexplored ? unexplored - found : 0;
  }
};

// ---------------------------------------------------------------------------
// single-source shortest paths (Bellman-Ford style frontier relaxation)
// ---------------------------------------------------------------------------
/// sssp_problem_t::unpack with the run's statistics folded in: distance[to[i]] <- the packed label's
/// distance, and (reached vertices, sum of their degrees) -> pinned run facts by the LAST workgroup to
/// finish (ticket on a device counter): no separate statistics pass, no hand-off to wait for.
template <typename vertex_t, typename weight_t, typename graph_t, typename decode_t>
__global__ void __launch_bounds__(1024)
    unpack_with_stats_kernel(graph_t G, std::size_t n, const unsigned long long* packed, unsigned far_bits,
                             weight_t* out, const vertex_t* to, const vertex_t* from, decode_t decode,
                             unsigned long long* counters, unsigned long long* facts) {
  __shared__ unsigned long long s_v[16], s_e[16];
  unsigned long long reached = 0, edges = 0;
  if (from)  // delivery by gather (see bfs_problem_t::gather_from); the pass below then only counts
    for (std::size_t v = blockIdx.x * (std::size_t)1024 + threadIdx.x; v < n; v += (std::size_t)gridDim.x * 1024)
      out[v] = decode((unsigned)(packed[from[v]] >> 32));
  for (std::size_t i = blockIdx.x * (std::size_t)1024 + threadIdx.x; i < n; i += (std::size_t)gridDim.x * 1024) {
    const unsigned bits = (unsigned)(packed[i] >> 32);
    if (!from)
      out[to ? (std::size_t)to[i] : i] = decode(bits);
    if (bits != far_bits) {
      ++reached;
      edges += (unsigned long long)G.get_number_of_neighbors((vertex_t)i);
    }
  }
  reached = hip::wave_sum(reached);
  edges = hip::wave_sum(edges);
  if ((threadIdx.x & 63) == 0) {
    s_v[threadIdx.x / 64] = reached;
    s_e[threadIdx.x / 64] = edges;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long v = 0, e = 0;
    for (int k = 0; k < 16; ++k) {
      v += s_v[k];
      e += s_e[k];
    }
    namespace k = hip::kernels;
    if (v)
      atomicAdd(&counters[k::C_OUT], v);
    if (e)
      atomicAdd(&counters[k::C_WORK], e);
    __threadfence();
    if (atomicAdd(&counters[k::C_SELECT], 1ull) + 1ull == (unsigned long long)gridDim.x) {
      // last one out: totals to pinned memory, the three counters back to zero (the invariant
      // between operators)
      facts[1] = atomicExch(&counters[k::C_OUT], 0ull);
      facts[2] = atomicExch(&counters[k::C_WORK], 0ull);
      atomicExch(&counters[k::C_SELECT], 0ull);
      __threadfence_system();
      facts[3] = 1ull;
    }
  }
}

template <typename graph_t>
struct sssp_problem_t : gunrock::problem_t<graph_t> {
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;

  vertex_t source;
  weight_t* distance;  // device, |V|, caller-owned
  hip::device_array_t<int> stamp;  // iteration in which a vertex last entered the frontier
  // One-pass form: (order-preserving bits of the distance) << 32 | round + 1 of the last improvement,
  // so that ONE 64-bit atomic min both lowers the distance and tells the first improver of a round
  // (the two-word form needs an atomic min and an atomic exchange per improvement, i.e. two scattered
  // memory-side RMWs -- the rate SSSP runs at).  Unpacked into `distance` when the run ends.
  bool packed_labels = true;
  hip::device_array_t<unsigned long long> packed;
  level_log_t log;
#ifdef GRX_SSSP_DIAG
  hip::device_array_t<unsigned long long> diag;  // diagnostic build: 64 x {issued, won, first of round}
#endif
  // Wide iterations: a 2-byte UPPER BOUND of every distance as it stood when the iteration began
  // (the top 16 of the order-preserving bits, rounded up), 8 MB at 2^22 vertices against 32 MB of
  // packed labels: the engine's batched per-edge predicate rejects a relaxation whose candidate
  // is not below that bound without touching the label (section 5 of DESIGN.md: the wide SSSP
  // iterations run at the fabric's rate of 128-byte label lines).
  hip::device_array_t<unsigned short> bound16;
  bool bounds_fresh = false;  // the previous iteration's label scan has just written them
  // renumbered graph (see bfs_problem_t): distances are delivered as distance[scatter_to[v]], or
  // gathered through the inverse permutation when the owner has it on the device
  const vertex_t* scatter_to = nullptr;
  const vertex_t* gather_from = nullptr;
  hip::device_array_t<weight_t> own_distance;
  /// unpack() also counts the reached vertices and their degrees into the context's pinned run
  /// facts (packed form): the caller reads them after enact() instead of running a statistics pass.
  bool collect_reach = false;
  /// The float array the two-word form runs in (the graph's numbering).
  weight_t* run_distance() { return scatter_to ? own_distance.data() : distance; }

  /// float <-> 32 bits whose UNSIGNED order is the float order (negative values included).
  __host__ __device__ static unsigned ordered_bits(weight_t x) {
    static_assert(sizeof(weight_t) == 4, "the packed SSSP labels hold a 32-bit distance");
    unsigned b;
    memcpy(&b, &x, 4);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
  }
  __host__ __device__ static weight_t from_ordered_bits(unsigned u) {
    const unsigned b = u ^ ((u >> 31) ? 0x80000000u : 0xffffffffu);
    weight_t x;
    memcpy(&x, &b, 4);
    return x;
  }

  sssp_problem_t(graph_t& G, vertex_t _source, weight_t* _distance,
                 std::shared_ptr<gcuda::multi_context_t> ctx)
      : gunrock::problem_t<graph_t>(G, ctx), source(_source), distance(_distance) {}

  void init() override {
    const std::size_t n = (std::size_t)this->get_graph().get_number_of_vertices();
    if (packed_labels)
      packed.resize(n);
    else
      stamp.resize(n);
    if (!packed_labels && scatter_to)
      own_distance.resize(n);
#ifdef GRX_SSSP_DIAG
    diag.resize(64 * 16);
    diag.zero();
#endif
    if (packed_labels)
      bound16.resize(n);
  }
  /// bound16[v] <- ceil16(ordered bits of the current distance): one pass, on the context's stream.
  void snapshot_bounds() {
    auto ctx = this->get_single_context();
    const std::size_t n = (std::size_t)this->get_graph().get_number_of_vertices();
    const unsigned long long* p = packed.data();
    unsigned short* b = bound16.data();
    hip::for_each_index(
        n, [p, b] __device__(std::size_t i) {
          const unsigned bits = (unsigned)(p[i] >> 32);
          const unsigned up = (bits >> 16) + ((bits & 0xffffu) ? 1u : 0u);
          b[i] = (unsigned short)(up > 0xffffu ? 0xffffu : up);
        },
        ctx->stream());
  }
  void reset() override {
    auto ctx = this->get_single_context();
    const std::size_t n = (std::size_t)this->get_graph().get_number_of_vertices();
    const vertex_t s = source;
    if (packed_labels) {
      unsigned long long* p = packed.data();
      const unsigned long long zero = (unsigned long long)ordered_bits(weight_t(0)) << 32;
      const unsigned long long far = (unsigned long long)ordered_bits(std::numeric_limits<weight_t>::max()) << 32;
      auto G = this->get_graph();
      unsigned long long* facts = ctx->workspace().run_facts();  // [0] <- the source's degree
      hip::for_each_index(
          n, [p, s, zero, far, G, facts] __device__(std::size_t i) {
            p[i] = (vertex_t)i == s ? zero : far;
            if ((vertex_t)i == s)
              facts[0] = (unsigned long long)G.get_number_of_neighbors(s);
          },
          ctx->stream());
    } else {
      weight_t* d = run_distance();
      int* st = stamp.data();
      hip::for_each_index(
          n, [d, st, s] __device__(std::size_t i) {
            d[i] = (vertex_t)i == s ? weight_t(0) : std::numeric_limits<weight_t>::max();
            st[i] = -1;
          },
          ctx->stream());
    }
    log = level_log_t();
  }
  /// End of a run: the caller's distance array in the caller's numbering (one pass, on the
  /// context's stream; nothing to do for the two-word form on the caller's own numbering).
  void unpack() {
    auto ctx = this->get_single_context();
    const std::size_t n = (std::size_t)this->get_graph().get_number_of_vertices();
    const unsigned long long* p = packed.data();
    weight_t* d = distance;
    const vertex_t* to = scatter_to;
    if (packed_labels && collect_reach) {
      auto G = this->get_graph();
      unsigned long long* facts = ctx->workspace().run_facts();
      facts[3] = 0ull;
      const unsigned far_bits = ordered_bits(std::numeric_limits<weight_t>::max());
      const unsigned grid = (unsigned)std::min<std::size_t>((n + 1023) / 1024, (std::size_t)ctx->compute_units() * 2);
      unpack_with_stats_kernel<vertex_t, weight_t><<<grid ? grid : 1, 1024, 0, ctx->stream()>>>(
          G, n, p, far_bits, d, to, to ? gather_from : nullptr,
          [] __device__(unsigned bits) { return from_ordered_bits(bits); },
          ctx->workspace().counters(), facts);
      GRX_HIP_CHECK(hipGetLastError());
      return;
    }
    if (packed_labels) {
      if (to)
        hip::for_each_index(
            n, [p, d, to] __device__(std::size_t i) { d[to[i]] = from_ordered_bits((unsigned)(p[i] >> 32)); },
            ctx->stream());
      else
        hip::for_each_index(
            n, [p, d] __device__(std::size_t i) { d[i] = from_ordered_bits((unsigned)(p[i] >> 32)); },
            ctx->stream());
    } else if (to) {
      const weight_t* own = own_distance.data();
      hip::for_each_index(n, [own, d, to] __device__(std::size_t i) { d[to[i]] = own[i]; }, ctx->stream());
    }
  }
};

template <typename problem_type, load_balance_t lb>
struct sssp_enactor_t : gunrock::enactor_t<problem_type> {
  using base_t = gunrock::enactor_t<problem_type>;
  using vertex_t = typename problem_type::vertex_t;
  using edge_t = typename problem_type::edge_t;
  using weight_t = typename problem_type::weight_t;
  using frontier_t = typename base_t::frontier_t;
  int max_iterations = 0;
  bool two_pass = false;  // true: advance + bypass filter, as reference algorithms/sssp.hxx does
  bool bound_filter = true;  // wide iterations: 2-byte bound mirror in front of the labels (packed form)
  int bound_from = -1;       // >= 0: first iteration that uses it (experiments); -1: by edges expanded so far
  bool early_live = true;    // the batched form before that point too (hot-first graphs), live labels beyond the image

  sssp_enactor_t(problem_type* p, std::shared_ptr<gcuda::multi_context_t> ctx,
                 enactor_properties_t props = enactor_properties_t())
      : base_t(p, ctx, props) {}

  void prepare_frontier(frontier_t* f, gcuda::multi_context_t& context) override {
    f->push_back(this->get_problem()->source, context.get_context(0)->stream());
  }

  bool is_converged(gcuda::multi_context_t& context) override {
    if (max_iterations && this->iteration >= max_iterations)
      return true;
    return base_t::is_converged(context);
  }

  void finalize(gcuda::multi_context_t&) override { this->get_problem()->unpack(); }

  /// The next frontier of a round that ran without an output frontier: the vertices whose packed
  /// label carries this round's tag (operators::filter::select_range), then swap.
  void scan_improved(const unsigned long long* packed, unsigned this_round, gcuda::multi_context_t& context) {
    auto E = this->get_enactor();
    auto P = this->get_problem();
    auto G = P->get_graph();
    auto ctx = context.get_context(0);
    const std::size_t n_scan = G.properties.leading_connected ? (std::size_t)G.properties.leading_connected
                                                              : (std::size_t)G.get_number_of_vertices();
    auto lowered_now = [packed, this_round] __device__(vertex_t const& v) -> bool {
      return (unsigned)packed[v] == this_round;
    };
    // the same pass leaves the next iteration's 2-byte bounds (snapshot_bounds() of the labels as
    // they stand now); vertices without edges are never a destination, their bounds are not read
    unsigned short* bound16 = P->bound16.data();
    auto bound_of = [packed, bound16] __device__(vertex_t const& v) {
      const unsigned bits = (unsigned)(packed[v] >> 32);
      const unsigned up = (bits >> 16) + ((bits & 0xffffu) ? 1u : 0u);
      bound16[v] = (unsigned short)(up > 0xffffu ? 0xffffu : up);
    };
    operators::filter::select_range(G, n_scan, lowered_now, *E->get_output_frontier(), *ctx, bound_of);
    P->bounds_fresh = true;
    E->swap_frontier_buffers();
  }

  void loop(gcuda::multi_context_t& context) override {
    auto E = this->get_enactor();
    auto P = this->get_problem();
    auto G = P->get_graph();
    P->log.note(E->get_input_frontier()->get_number_of_elements(),
                E->get_input_frontier()->work_hint());

    weight_t* distance = P->run_distance();
    int* stamp = P->stamp.data();
    const int round = this->iteration;

    // Relax and keep ONE copy of an improved vertex per round: the exchange on its stamp admits
    // exactly the first improver, so the output frontier is duplicate-free and the reference's
    // separate filter pass (sssp.hxx:126-139, a racy stamp test that lets concurrent duplicates
    // through) is not needed.  Later improvements of the same round still lower distance[dst];
    // the next round reads the latest value.
    if (two_pass) {
      // the formulation of the reference client (algorithms/sssp.hxx:110-144), for comparison:
      // every improvement is emitted, then a bypass filter drops what a (racy) stamp test sees
      // as already in this round's frontier; surviving duplicates are expanded again next round
      auto relax_all = [distance] __host__ __device__(vertex_t const& src, vertex_t const& dst,
                                                      edge_t const& edge, weight_t const& w) -> bool {
        weight_t through = thread::load(&distance[src]) + w;
        return through < math::atomic::min(&distance[dst], through);
      };
      auto once = [stamp, round] __host__ __device__(vertex_t const& v) -> bool {
        if (stamp[v] == round)
          return false;
        stamp[v] = round;
        return true;
      };
      operators::advance::execute<lb>(G, E, relax_all, context);
      operators::filter::execute<operators::filter_algorithm_t::bypass>(G, E, once, context);
      return;
    }
    if (P->packed_labels) {
      unsigned long long* packed = P->packed.data();
      const unsigned this_round = (unsigned)round + 1u;  // 0 = never improved
      auto relax_packed = [packed, this_round] __host__ __device__(
                              vertex_t const& src, vertex_t const& dst, edge_t const& edge,
                              weight_t const& w) -> bool {
        const weight_t through =
            problem_type::from_ordered_bits((unsigned)(thread::load(&packed[src]) >> 32)) + w;
        const unsigned long long key =
            ((unsigned long long)problem_type::ordered_bits(through) << 32) | this_round;
        const unsigned long long old = math::atomic::min(&packed[dst], key);
        if (!((unsigned)(key >> 32) < (unsigned)(old >> 32)))
          return false;                        // no shorter: (an equal distance never lowers the word,
                                               // rounds only grow)
        return (unsigned)old != this_round;    // the first improver of this round enqueues the vertex
      };
#ifdef GRX_SSSP_DIAG
      {  // diagnostic build: how many RMWs pass the pre-test, how many of them lower the label
        unsigned long long* diag = P->diag.data();
        auto relax_counted = [packed, this_round, diag] __device__(
                                 vertex_t const& src, vertex_t const& dst, edge_t const& edge,
                                 weight_t const& w) -> bool {
          const weight_t through =
              problem_type::from_ordered_bits((unsigned)(thread::load(&packed[src]) >> 32)) + w;
          const unsigned long long key =
              ((unsigned long long)problem_type::ordered_bits(through) << 32) | this_round;
          unsigned long long* slot = diag + (blockIdx.x & 63u) * 16;
          const unsigned long long seen =
              __hip_atomic_load(&packed[dst], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!(key < seen))
            return false;
          atomicAdd(slot + 0, 1ull);
          const unsigned long long old = ::atomicMin(&packed[dst], key);
          if (!((unsigned)(key >> 32) < (unsigned)(old >> 32)))
            return false;
          atomicAdd(slot + 1, 1ull);
          if ((unsigned)old == this_round)
            return false;
          atomicAdd(slot + 2, 1ull);
          return true;
        };
        operators::advance::execute<lb>(G, E, relax_counted, context);
        auto h = P->diag.to_host();
        unsigned long long t[3] = {0, 0, 0};
        for (int b = 0; b < 64; ++b)
          for (int k = 0; k < 3; ++k)
            t[k] += h[b * 16 + k];
        std::fprintf(stderr, "[sssp diag] iteration %d: RMWs issued %llu, lowered the label %llu, first of the round %llu\n",
                     round, t[0], t[1], t[2]);
        P->diag.zero();
        return;
      }
#endif
      {
        auto ctx = context.get_context(0);
        const unsigned long long work = E->get_input_frontier()->work_hint();
        // The bounds are those of the iteration's START: while most destinations are still
        // unreached (no bound) nearly every edge passes and the batched form only adds its packing
        // (RMAT-22 source 0, iteration 1: 1.38 against 1.10 ms); once a quarter of the graph's edges
        // have been expanded the hubs are reached and it pays (iterations 2 / 3: 0.98 -> 0.82, 0.30 ->
        // 0.28 ms; with hot-first numbering 0.93 -> 0.64, 0.29 -> 0.25 ms).
        const bool reached_enough =
            bound_from >= 0 ? round >= bound_from
                            : work != frontier_t::unknown_work &&
                                  4 * ((unsigned long long)P->log.edges_expanded - work) >=  // before this one
                                      (unsigned long long)G.get_number_of_edges();
        // before that point the batched form still pays on a hot-first numbered graph when the ids the
        // image does not cover are asked for their LIVE label instead of the (mostly absent) bound:
        // the frontier of such an iteration is the hubs, whose bounds the image holds
        const bool early = !reached_enough && early_live && G.properties.leading_connected != 0;
        if (lb == load_balance_t::block_mapped && bound_filter && (reached_enough || early) &&
            ctx->options().settled_filter &&
            !ctx->options().holes_layout && work != frontier_t::unknown_work &&
            work >= ctx->options().settled_min_work) {
          if (!P->bounds_fresh) {  // part of this iteration's advance: timed with it when kernels are timed
            operators::advance::detail::clocked_t clock(*ctx);
            P->snapshot_bounds();
            clock.stop();
          }
          P->bounds_fresh = false;
          const unsigned short* bound16 = P->bound16.data();
          // PURE: "relax_packed would return false for this edge and change nothing" -- the
          // candidate is not below an upper bound of the destination's distance (0xffff: no bound)
          auto cached = [packed] __device__(vertex_t const& src, vertex_t const& dst, edge_t const& edge,
                                            weight_t const& w, unsigned short const& bound) -> bool {
            const weight_t through =
                problem_type::from_ordered_bits((unsigned)(packed[src] >> 32)) + w;
            const unsigned b = bound;
            return b != 0xffffu && problem_type::ordered_bits(through) >= (b << 16);
          };
          const bool scan = ctx->options().label_scan_min_work && work >= ctx->options().label_scan_min_work;
          auto run = [&](auto hinted) {
            if (scan) {
              // no output frontier: the round tag in a label's low word says "lowered in this round"
              ctx->options().defer_sync_of_none_output = true;  // scan_improved below is its hand-off
              operators::advance::execute<lb, operators::advance_direction_t::forward,
                                          operators::advance_io_type_t::vertices,
                                          operators::advance_io_type_t::none>(G, E, hinted, context);
              ctx->options().defer_sync_of_none_output = false;
              scan_improved(packed, this_round, context);
            } else {
              operators::advance::execute<lb>(G, E, hinted, context);
            }
          };
          if (reached_enough) {
            auto not_shorter = [cached, bound16] __device__(vertex_t const& src, vertex_t const& dst,
                                                            edge_t const& edge, weight_t const& w) -> bool {
              return cached(src, dst, edge, w, bound16[dst]);
            };
            run(operators::advance::with_bounds<vertex_t>(relax_packed, not_shorter, cached, bound16,
                                                          (std::size_t)G.get_number_of_vertices()));
          } else {
            auto not_shorter_live = [packed] __device__(vertex_t const& src, vertex_t const& dst,
                                                        edge_t const& edge, weight_t const& w) -> bool {
              const weight_t through =
                  problem_type::from_ordered_bits((unsigned)(packed[src] >> 32)) + w;
              return problem_type::ordered_bits(through) >= (unsigned)(packed[dst] >> 32);
            };
            run(operators::advance::with_bounds<vertex_t>(relax_packed, not_shorter_live, cached, bound16,
                                                          (std::size_t)G.get_number_of_vertices()));
          }
          return;
        }
      }
      {
        auto ctx = context.get_context(0);
        const unsigned long long work = E->get_input_frontier()->work_hint();
        if (lb == load_balance_t::block_mapped && !ctx->options().holes_layout &&
            ctx->options().label_scan_min_work && work != frontier_t::unknown_work &&
            work >= ctx->options().label_scan_min_work) {
          ctx->options().defer_sync_of_none_output = true;  // scan_improved below is its hand-off
          operators::advance::execute<lb, operators::advance_direction_t::forward,
                                      operators::advance_io_type_t::vertices,
                                      operators::advance_io_type_t::none>(G, E, relax_packed, context);
          ctx->options().defer_sync_of_none_output = false;
          scan_improved(packed, this_round, context);
          return;
        }
      }
      operators::advance::execute<lb>(G, E, relax_packed, context);
      return;
    }
    auto relax = [distance, stamp, round] __host__ __device__(
                     vertex_t const& src, vertex_t const& dst, edge_t const& edge,
                     weight_t const& w) -> bool {
      weight_t through = thread::load(&distance[src]) + w;
      if (!(through < math::atomic::min(&distance[dst], through)))
        return false;
      return math::atomic::exch(&stamp[dst], round) != round;
    };
    // Measured and not used here: the per-edge form of the settled hint (operators/settled.hxx,
    // with_rejects: the engine evaluates `!(distance[src] + w < distance[dst])` for every edge with
    // the loads batched and calls relax() on packed groups of the edges that pass).  RMAT-22, source
    // 0: iterations 1 / 2 / 3 take 1387 / 928 / 290 us against 1310 / 968 / 288 us -- 10 % of the edges
    // survive, and their relax() calls cost what the batching saves.
    operators::advance::execute<lb>(G, E, relax, context);
  }
};

// ---------------------------------------------------------------------------
// PageRank, push formulation with dangling-mass redistribution
// ---------------------------------------------------------------------------
template <typename graph_t, typename weight_t>
__global__ void __launch_bounds__(256)
    out_scale_kernel(graph_t G, std::size_t n, unsigned long long stride, weight_t alpha, weight_t* scale) {
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  const unsigned lane = threadIdx.x & 63u;
  const std::size_t waves = (std::size_t)gridDim.x * 4;
  // wavefront w takes rows (i * stride) mod n for i = w, w + waves, ...: with i itself, the rows of
  // one wavefront are a power-of-two apart, which in an R-MAT numbering are all the hubs (20 ms of
  // one wavefront walking them one after the other, against 2 ms)
  for (std::size_t i = (std::size_t)blockIdx.x * 4 + threadIdx.x / 64; i < n; i += waves) {
    const std::size_t v = (std::size_t)(((unsigned long long)i * stride) % (unsigned long long)n);
    const edge_t begin = G.get_starting_edge((vertex_t)v);
    const edge_t count = G.get_number_of_neighbors((vertex_t)v);
    weight_t total = 0;
    edge_t e = (edge_t)lane;
    if (count >= 64 * 8) {  // a hub is one wavefront's alone: eight loads in flight per lane
      weight_t part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (; e + 64 * 7 < count; e += 64 * 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          part[k] += G.get_edge_weight(begin + e + 64 * k);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        total += part[k];
    }
    for (; e < count; e += 64)
      total += G.get_edge_weight(begin + e);
    total = hip::wave_sum(total);
    if (lane == 0)
      scale[v] = total != 0 ? alpha / total : weight_t(0);
  }
}

template <typename graph_t>
struct pr_problem_t : gunrock::problem_t<graph_t> {
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;

  weight_t alpha, tol;
  weight_t* rank;  // device, |V|: the caller's array, or `own_rank` on a renumbered copy of the graph
  // A run on the hot-first copy of the caller's graph (reorder.hxx; the sources every edge looks up
  // are then packed by falling out-degree: push 6.0 -> 3.9, pull 3.0 -> 2.4 ms per iteration on a
  // directed R-MAT-24, a scrambled input 8.7 / 4.3 ms, tools/pr_layout_probe.py): ranks are kept in
  // the copy's numbering and handed over as caller_rank[v] = rank[gather_from[v]] when the run ends.
  const vertex_t* gather_from = nullptr;
  weight_t* caller_rank = nullptr;
  hip::device_array_t<weight_t> own_rank;
  hip::device_array_t<weight_t> previous;
  hip::device_array_t<weight_t> out_scale;  // alpha / (sum of out-weights), 0 for dangling
  // pull formulation (pr_pull_enactor_t): what every vertex hands to each out-neighbour this
  // iteration, and the in-edge lists of >= RED_HUB edges cut into chunks (once per problem)
  bool pull = false;
  hip::device_array_t<weight_t> contribution;
  using hub_chunk_t = hip::kernels::row_chunk_t<vertex_t, edge_t>;
  hip::device_array_t<hub_chunk_t> hub_chunks;

  pr_problem_t(graph_t& G, weight_t _alpha, weight_t _tol, weight_t* _rank,
               std::shared_ptr<gcuda::multi_context_t> ctx)
      : gunrock::problem_t<graph_t>(G, ctx), alpha(_alpha), tol(_tol), rank(_rank) {}

  /// End of a run (inside the timed enact()): the caller's array in the caller's numbering.
  void deliver() {
    if (!gather_from)
      return;
    auto ctx = this->get_single_context();
    const std::size_t n = (std::size_t)this->get_graph().get_number_of_vertices();
    const vertex_t* from = gather_from;
    const weight_t* own = rank;
    weight_t* out = caller_rank;
    hip::for_each_index(n, [own, out, from] __device__(std::size_t v) { out[v] = own[from[v]]; }, ctx->stream());
  }

  void init() override {
    auto g = this->get_graph();
    const std::size_t n = (std::size_t)g.get_number_of_vertices();
    if (gather_from) {
      caller_rank = rank;
      own_rank.resize(n);
      rank = own_rank.data();
    }
    previous.resize(n);
    out_scale.resize(n);
    if (pull)
      contribution.resize(n);
    if (pull && g.can_pull()) {
      auto in = g.in_edges();
      std::vector<edge_t> offsets(n + 1);
      GRX_HIP_CHECK(hipMemcpy(offsets.data(), in.get_row_offsets(), (n + 1) * sizeof(edge_t),
                              hipMemcpyDeviceToHost));
      std::vector<hub_chunk_t> chunks;
      for (std::size_t v = 0; v < n; ++v) {
        const unsigned deg = (unsigned)(offsets[v + 1] - offsets[v]);
        if (deg < hip::kernels::RED_HUB)
          continue;
        for (unsigned off = 0; off < deg; off += hip::kernels::RED_CHUNK) {
          hub_chunk_t c;
          c.row = (vertex_t)v;
          c.first = offsets[v] + (edge_t)off;
          c.count = (int)((deg - off < hip::kernels::RED_CHUNK) ? deg - off : hip::kernels::RED_CHUNK);
          chunks.push_back(c);
        }
      }
      hub_chunks.assign(chunks);
    }
  }
  void reset() override {
    auto ctx = this->get_single_context();
    auto g = this->get_graph();
    const std::size_t n = (std::size_t)g.get_number_of_vertices();
    hip::fill(rank, n, (weight_t)(1.0 / (double)n), ctx->stream());
    hip::fill(previous.data(), n, weight_t(0), ctx->stream());
    // alpha / (sum of a vertex's out-weights): one WAVEFRONT per row (a thread per row walked the
    // 370 K edges of R-MAT-24's largest hub alone: 52 ms of every reset against 1.6 ms)
    unsigned long long stride = 1;  // coprime with n: i -> (i * stride) mod n is a permutation
    for (unsigned long long p : {2654435761ull, 40503ull, 7919ull, 104729ull, 1299709ull})
      if (n > 1 && std::gcd(p % (unsigned long long)n, (unsigned long long)n) == 1 && p % n != 0) {
        stride = p % (unsigned long long)n;
        break;
      }
    out_scale_kernel<<<(unsigned)ctx->compute_units() * 8u, 256, 0, ctx->stream()>>>(g, n, stride, alpha,
                                                                                       out_scale.data());
    GRX_HIP_CHECK(hipGetLastError());
    ctx->synchronize();
  }
};

template <typename problem_type, load_balance_t lb>
struct pr_enactor_t : gunrock::enactor_t<problem_type> {
  using base_t = gunrock::enactor_t<problem_type>;
  using vertex_t = typename problem_type::vertex_t;
  using edge_t = typename problem_type::edge_t;
  using weight_t = typename problem_type::weight_t;
  int max_iterations = 0;

  pr_enactor_t(problem_type* p, std::shared_ptr<gcuda::multi_context_t> ctx,
               enactor_properties_t props)
      : base_t(p, ctx, props) {}

  void loop(gcuda::multi_context_t& context) override {
    auto E = this->get_enactor();
    auto P = this->get_problem();
    auto G = P->get_graph();
    auto ctx = context.get_context(0);
    const std::size_t n = (std::size_t)G.get_number_of_vertices();
    weight_t* rank = P->rank;
    weight_t* previous = P->previous.data();
    weight_t* scale = P->out_scale.data();
    const weight_t alpha = P->alpha;

    GRX_HIP_CHECK(hipMemcpyAsync(previous, rank, n * sizeof(weight_t), hipMemcpyDeviceToDevice,
                                 ctx->stream()));
    const weight_t dangling = hip::transform_reduce(
        n,
        [rank, scale, alpha] __device__(std::size_t i) -> weight_t {
          return scale[i] == 0 ? alpha * rank[i] : weight_t(0);
        },
        weight_t(0), rocprim::plus<weight_t>(), *ctx);
    hip::fill(rank, n, (1 - alpha + dangling) / (weight_t)n, ctx->stream());

    // PageRank iterates: the destination-sorted list pays for itself within this run, so this
    // client asks for it before its first advance (the operator alone would walk the first call
    // row by row -- 41 ms on R-MAT-24 -- and sort on the second)
    if (this->iteration == 0)
      (void)operators::advance::by_destination::prepared(G, this->unique_id, *ctx);
    auto spread = [rank, previous, scale] __host__ __device__(vertex_t const& src,
                                                              vertex_t const& dst,
                                                              edge_t const& edge,
                                                              weight_t const& w) -> bool {
      math::atomic::add(rank + dst, previous[src] * scale[src] * w);
      return false;
    };
    operators::advance::execute<lb, operators::advance_direction_t::forward,
                                operators::advance_io_type_t::graph,
                                operators::advance_io_type_t::none>(G, E, spread, context);
  }

  void finalize(gcuda::multi_context_t&) override { this->get_problem()->deliver(); }

  bool is_converged(gcuda::multi_context_t& context) override {
    if (this->iteration == 0)
      return false;
    if (max_iterations && this->iteration >= max_iterations)
      return true;
    auto P = this->get_problem();
    auto ctx = context.get_context(0);
    const std::size_t n = (std::size_t)P->get_graph().get_number_of_vertices();
    weight_t* rank = P->rank;
    weight_t* previous = P->previous.data();
    const weight_t err = hip::transform_reduce(
        n,
        [rank, previous] __device__(std::size_t i) -> weight_t {
          weight_t d = rank[i] - previous[i];
          return d < 0 ? -d : d;
        },
        weight_t(0), rocprim::maximum<weight_t>(), *ctx);
    return err < P->tol;
  }
};

/// PageRank, PULL formulation: p[v] = (1 - alpha + dangling) / n + sum over IN-edges (u -> v) of
/// p_prev[u] * scale[u] * w -- the same quantities as the push client above (and pr.hxx), summed
/// per destination instead of scattered with one float atomic per edge.  Needs in-edges (an
/// undirected graph, or a directed one with an attached transpose).  New relative to the
/// reference; float sums are taken in a different order, hence a tolerance against the push form.
template <typename problem_type>
struct pr_pull_enactor_t : gunrock::enactor_t<problem_type> {
  using base_t = gunrock::enactor_t<problem_type>;
  using vertex_t = typename problem_type::vertex_t;
  using edge_t = typename problem_type::edge_t;
  using weight_t = typename problem_type::weight_t;
  int max_iterations = 0;
  bool walk_sorted_list = true;  ///< GRX_PR_PULL_WALK=0: always the per-destination lists

  pr_pull_enactor_t(problem_type* p, std::shared_ptr<gcuda::multi_context_t> ctx,
                    enactor_properties_t props)
      : base_t(p, ctx, props) {}

  void loop(gcuda::multi_context_t& context) override {
    namespace k = hip::kernels;
    auto P = this->get_problem();
    auto G = P->get_graph();
    auto ctx = context.get_context(0);

    const std::size_t n = (std::size_t)G.get_number_of_vertices();
    weight_t* rank = P->rank;
    weight_t* previous = P->previous.data();
    weight_t* scale = P->out_scale.data();
    weight_t* give = P->contribution.data();
    const weight_t alpha = P->alpha;

    GRX_HIP_CHECK(hipMemcpyAsync(previous, rank, n * sizeof(weight_t), hipMemcpyDeviceToDevice,
                                 ctx->stream()));
    hip::for_each_index(
        n, [previous, scale, give] __device__(std::size_t i) { give[i] = previous[i] * scale[i]; },
        ctx->stream());
    const weight_t dangling = hip::transform_reduce(
        n,
        [previous, scale, alpha] __device__(std::size_t i) -> weight_t {
          return scale[i] == 0 ? alpha * previous[i] : weight_t(0);
        },
        weight_t(0), rocprim::plus<weight_t>(), *ctx);
    const weight_t base = (1 - alpha + dangling) / (weight_t)n;
    // the sums per destination as a walk over the engine's destination-sorted edge list
    // (operators/by_destination.hxx) once it exists for this graph: one lane per edge, ONE lookup
    // give[src] per edge, the adds of neighbouring lanes into one rank word combined -- the lists
    // below keep 16 lanes on a destination and gather at a third of the rate
    if (walk_sorted_list) {
      namespace bd = operators::advance::by_destination;
      auto gather = [rank, give] __host__ __device__(vertex_t const& src, vertex_t const& dst,
                                                     edge_t const& edge, weight_t const& w) -> bool {
        math::atomic::add(rank + dst, give[src] * w);
        return false;
      };
      const void* items = bd::prepared(G, this->unique_id, *ctx);
      if (!items && !G.can_pull())  // a copy without a transpose: the list is all it can sum over
        items = bd::prepared(G, this->unique_id, *ctx);
      if (items) {
        hip::for_each_index(n, [rank, base] __device__(std::size_t i) { rank[i] = base; }, ctx->stream());
        bd::enqueue(G, items, gather, *ctx);
        return;
      }
      if (!G.can_pull()) {  // no room for the list: the same sums scattered row by row
        hip::for_each_index(n, [rank, base] __device__(std::size_t i) { rank[i] = base; }, ctx->stream());
        operators::advance::execute<load_balance_t::block_mapped, operators::advance_direction_t::forward,
                                    operators::advance_io_type_t::graph,
                                    operators::advance_io_type_t::none>(G, this->get_enactor(), gather, context);
        return;
      }
    }
    error::throw_if_exception(!G.can_pull(), "pull PageRank needs in-edges (attach a transpose)");
    auto in = G.in_edges();
    const unsigned grid = (unsigned)ctx->compute_units() * 8;
    k::row_group_sum_kernel<<<grid, k::RED_BLOCK, 0, ctx->stream()>>>(in, give, base, rank);
    GRX_HIP_CHECK(hipGetLastError());
    if (P->hub_chunks.size()) {
      const std::size_t chunks = P->hub_chunks.size();
      const unsigned hub_grid = (unsigned)(chunks < grid ? chunks : grid);
      k::hub_chunk_sum_kernel<<<hub_grid, k::RED_BLOCK, 0, ctx->stream()>>>(
          in, give, P->hub_chunks.data(), chunks, rank);
      GRX_HIP_CHECK(hipGetLastError());
    }
  }

  void finalize(gcuda::multi_context_t&) override { this->get_problem()->deliver(); }

  bool is_converged(gcuda::multi_context_t& context) override {
    if (this->iteration == 0)
      return false;
    if (max_iterations && this->iteration >= max_iterations)
      return true;
    auto P = this->get_problem();
    auto ctx = context.get_context(0);
    const std::size_t n = (std::size_t)P->get_graph().get_number_of_vertices();
    weight_t* rank = P->rank;
    weight_t* previous = P->previous.data();
    const weight_t err = hip::transform_reduce(
        n,
        [rank, previous] __device__(std::size_t i) -> weight_t {
          weight_t d = rank[i] - previous[i];
          return d < 0 ? -d : d;
        },
        weight_t(0), rocprim::maximum<weight_t>(), *ctx);
    return err < P->tol;
  }
};

}  // namespace clients
}  // namespace essentials_amd
