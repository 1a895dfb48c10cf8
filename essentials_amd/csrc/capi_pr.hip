/** @file capi_pr.hip  grx_pagerank == gunrock::pr::run (reference algorithms/pr.hxx:182-216). */
#include "capi_internal.hxx"
#include "clients.hxx"

using namespace essentials_amd;

extern "C" int grx_pagerank(grx_context_t ctx, grx_graph_t g, float alpha, float tol, float* d_p,
                            const grx_options* opt, grx_stats* stats) {
  if (!ctx || !g || !d_p)
    return invalid("grx_pagerank: NULL argument");
  grx_options o;
  grx_default_options(&o);
  if (opt)
    o = *opt;
  if (o.holes_layout)
    o.holes_layout = 0;  // no output frontier
  return guarded([&] {
    return with_load_balance(o.load_balance, [&](auto lb_tag) -> int {
      constexpr auto lb = decltype(lb_tag)::value;
      using problem_type = clients::pr_problem_t<graph_type>;
      using enactor_type = clients::pr_enactor_t<problem_type, lb>;
      if (o.direction_optimized)
        if (int rc = ensure_can_pull(ctx, g))
          return rc;
      scoped_options scope(ctx->single(), &o);
      // both forms run on the hot-first renumbered copy of the graph (reorder.hxx) when it is large
      // enough for the walk over the destination-sorted edge list (operators/by_destination.hxx:
      // that walk needs out-edges only) and hand the ranks over in the caller's numbering
      grx_graph_s* run_on = g;
      const unsigned long long walk_from = ctx->single().options().by_destination_min_edges;
      if (walk_from && (unsigned long long)g->nnz >= walk_from)
        if (grx_graph_s* h = hot_copy(ctx, g, /*csr_only=*/true))
          run_on = h;
      if (const char* e = std::getenv("GRX_PR_HOT_FIRST"))
        if (std::atoi(e) == 0)
          run_on = g;
      bool pull_walk = true;  // GRX_PR_PULL_WALK=0: round 2's per-destination lists (they need the transpose)
      if (const char* e = std::getenv("GRX_PR_PULL_WALK"))
        pull_walk = std::atoi(e) != 0;
      if (o.direction_optimized && !pull_walk)
        run_on = g;
      graph_type G = run_on->view();
      problem_type problem(G, alpha, tol, d_p, ctx->mc);
      if (run_on != g)
        problem.gather_from = g->hot_rank_of_device.data();
      problem.pull = o.direction_optimized != 0;
      problem.init();
      problem.reset();
      enactor_properties_t props;
      props.self_manage_frontiers = true;
      float ms = 0;
      int iterations = 0;
      if (problem.pull) {
        clients::pr_pull_enactor_t<problem_type> enactor(&problem, ctx->mc, props);
        enactor.max_iterations = o.max_iterations;
        enactor.walk_sorted_list = pull_walk;
        ms = enactor.enact();
        iterations = enactor.iteration;
      } else {
        enactor_type enactor(&problem, ctx->mc, props);
        enactor.max_iterations = o.max_iterations;
        ms = enactor.enact();
        iterations = enactor.iteration;
      }
      if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->elapsed_ms = ms;
        stats->iterations = iterations;
        stats->advance_kernel_ms = ctx->single().kernel_clock().total_ms;
        stats->advance_launches = ctx->single().kernel_clock().launches;
        stats->vertices_reached = g->n_rows;
        stats->edges_traversed = (int64_t)g->nnz * iterations;
        stats->pull_iterations = problem.pull ? iterations : 0;
      }
      return (int)GRX_OK;
    });
  });
}

/* One PageRank iteration's local half on a vertex-partitioned graph (replicas of p,
 * every rank scatters from the rows it owns into a private partial vector, the host all-reduces
 * the partials).  The arithmetic per edge is pr.hxx:140-146's. */
extern "C" int grx_pagerank_partitioned_scatter(grx_context_t ctx, grx_graph_t local, float alpha,
                                                const float* d_rank, float* d_scale,
                                                int32_t compute_scale, float* d_partial,
                                                int32_t row_begin, int32_t row_end,
                                                const grx_options* opt) {
  if (!ctx || !local || !d_rank || !d_scale || !d_partial || row_begin < 0 || row_end < row_begin ||
      row_end > local->n_rows)
    return invalid("grx_pagerank_partitioned_scatter: bad arguments");
  grx_options o;
  grx_default_options(&o);
  if (opt)
    o = *opt;
  o.holes_layout = 0;
  return guarded([&] {
    return with_load_balance(o.load_balance, [&](auto lb_tag) -> int {
      constexpr auto lb = decltype(lb_tag)::value;
      auto& sc = ctx->single();
      scoped_options scope(sc, &o);
      graph_type G = local->view();
      const std::size_t n = (std::size_t)local->n_rows;
      float* scale = d_scale;
      if (compute_scale) {
        // alpha / (sum of out-weights) for owned rows with edges; 0 elsewhere (pr.hxx:77-91)
        hip::for_each_index(
            n,
            [G, scale, alpha] __device__(std::size_t i) {
              float total = 0;
              const edge_t begin = G.get_starting_edge((vertex_t)i);
              const edge_t end = begin + G.get_number_of_neighbors((vertex_t)i);
              for (edge_t e = begin; e < end; ++e)
                total += G.get_edge_weight(e);
              scale[i] = total != 0 ? alpha / total : 0.0f;
            },
            sc.stream());
      }
      // partial[0..V) = 0, partial[V] = alpha * (rank mass of the OWNED dangling vertices)
      hip::fill(d_partial, n, 0.0f, sc.stream());
      const std::size_t owned = (std::size_t)(row_end - row_begin);
      const float* rank = d_rank;
      const std::size_t lo = (std::size_t)row_begin;
      const float dangling = hip::transform_reduce(
          owned,
          [rank, scale, alpha, lo] __device__(std::size_t i) -> float {
            return scale[lo + i] == 0 ? alpha * rank[lo + i] : 0.0f;
          },
          0.0f, rocprim::plus<float>(), sc);
      hip::fill(d_partial + n, 1, dangling, sc.stream());
      float* partial = d_partial;
      auto spread = [partial, rank, scale] __host__ __device__(vertex_t const& src, vertex_t const& dst,
                                                               edge_t const& edge,
                                                               weight_t const& w) -> bool {
        math::atomic::add(partial + dst, rank[src] * scale[src] * w);
        return false;
      };
      frontier_type none_in, none_out;
      hip::device_array_t<edge_t> segments;
      operators::advance::execute<lb, operators::advance_direction_t::forward,
                                  operators::advance_io_type_t::graph,
                                  operators::advance_io_type_t::none>(G, spread, &none_in, &none_out,
                                                                      segments, *ctx->mc);
      sc.synchronize();
      return (int)GRX_OK;
    });
  });
}
