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
      scoped_options scope(ctx->single(), &o);
      graph_type G = g->view();
      problem_type problem(G, alpha, tol, d_p, ctx->mc);
      problem.init();
      problem.reset();
      enactor_properties_t props;
      props.self_manage_frontiers = true;
      enactor_type enactor(&problem, ctx->mc, props);
      enactor.max_iterations = o.max_iterations;
      const float ms = enactor.enact();
      if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->elapsed_ms = ms;
        stats->iterations = enactor.iteration;
        stats->advance_kernel_ms = ctx->single().kernel_clock().total_ms;
        stats->advance_launches = ctx->single().kernel_clock().launches;
        stats->vertices_reached = g->n_rows;
        stats->edges_traversed = (int64_t)g->nnz * enactor.iteration;
      }
      return (int)GRX_OK;
    });
  });
}
