/** @file capi_bfs.hip  grx_bfs == gunrock::bfs::run (reference algorithms/bfs.hxx:151-176). */
#include "capi_internal.hxx"
#include "clients.hxx"

using namespace essentials_amd;

extern "C" int grx_bfs(grx_context_t ctx, grx_graph_t g, int32_t source, int32_t* d_distances,
                       int32_t* /*d_predecessors*/, const grx_options* opt, grx_stats* stats) {
  if (!ctx || !g || !d_distances)
    return invalid("grx_bfs: NULL argument");
  if (source < 0 || source >= g->n_rows)
    return invalid("grx_bfs: source out of range");
  grx_options o;
  grx_default_options(&o);
  if (opt)
    o = *opt;
  return guarded([&] {
    return with_load_balance(o.load_balance, [&](auto lb_tag) -> int {
      constexpr auto lb = decltype(lb_tag)::value;
      using problem_type = clients::bfs_problem_t<graph_type>;
      if (o.direction_optimized)
        if (int rc = ensure_can_pull(ctx, g))
          return rc;
      scoped_options scope(ctx->single(), &o);
      // the search runs on the hot-first renumbered copy of the graph (reorder.hxx) and delivers
      // its depths in the caller's numbering; the form that stands for the unchanged reference
      // client (call_every_edge) and the holes layout keep the caller's graph
      grx_graph_s* run_on = g;
      if (!o.call_every_edge && !o.holes_layout)
        if (grx_graph_s* h = hot_copy(ctx, g))
          run_on = h;
      graph_type G = run_on->view();
      problem_type problem(G, run_on == g ? source : g->hot_rank_of[(std::size_t)source], d_distances,
                           ctx->mc);
      if (run_on != g)
        problem.scatter_to = g->hot_vertex_of.data();
        problem.gather_from = g->hot_rank_of_device.data();
      // push search: one byte per vertex while it runs once 4-byte depths outgrow the eight L2s
      // (GRX_BFS_BYTE_LABELS=0/1 overrides; measurements in DESIGN.md, "Larger graphs")
      if (!o.direction_optimized) {
        problem.byte_labels = g->n_rows > (1ll << 22);
        if (const char* e = std::getenv("GRX_BFS_BYTE_LABELS"))
          problem.byte_labels = std::atoi(e) != 0;
      }
      problem.init();
      problem.reset();
      enactor_properties_t props;
      if (o.frontier_sizing_factor > 0)
        props.frontier_sizing_factor = o.frontier_sizing_factor;
      float ms = 0;
      int iterations = 0, pulls = 0;
      if (o.direction_optimized) {
        clients::bfs_do_enactor_t<problem_type, lb> enactor(&problem, ctx->mc, props);
        enactor.max_iterations = o.max_iterations;
        if (o.do_alpha > 0) enactor.alpha = o.do_alpha;
        if (o.do_beta > 0) enactor.beta = o.do_beta;
        ms = enactor.enact();
        iterations = enactor.iteration;
        pulls = enactor.pull_iterations;
      } else {
        clients::bfs_enactor_t<problem_type, lb> enactor(&problem, ctx->mc, props);
        enactor.max_iterations = o.max_iterations;
        if (const char* e = std::getenv("GRX_BFS_MARK"))
          enactor.mark_without_claim = std::atoi(e) != 0;
        ms = enactor.enact();
        iterations = enactor.iteration;
      }
      if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->pull_iterations = pulls;
        stats->elapsed_ms = ms;
        stats->iterations = iterations;
        stats->advance_kernel_ms = ctx->single().kernel_clock().total_ms;
        stats->advance_launches = ctx->single().kernel_clock().launches;
        stats->levels_recorded = problem.log.levels < 64 ? problem.log.levels : 64;
        for (int i = 0; i < stats->levels_recorded; ++i)
          stats->frontier_slots[i] = problem.log.input_slots[i];
        // the first frontier ({source}) carries no work hint: add the source's own degree.  A push
        // search with packed frontiers discovers every vertex exactly once and every level's frontier
        // came with the degree sum of its vertices: the counts are sums over the level log, the
        // source's degree was left in pinned memory by the reset pass -- no statistics pass
        const unsigned long long* facts = ctx->single().workspace().run_facts();
        if (!o.direction_optimized && !o.holes_layout && o.max_iterations == 0 &&
            problem.log.unknown_work_levels == 1) {
          stats->vertices_reached = problem.log.slots_total;
          stats->edges_traversed = problem.log.edges_expanded + (long long)facts[0];
          stats->edges_expanded = stats->edges_traversed;
        } else {
          stats->edges_expanded = problem.log.edges_expanded +
                                  reach_stats(g, d_distances, (int32_t)INT32_MAX, source, ctx->single(), stats);
        }
      }
      return (int)GRX_OK;
    });
  });
}
