/** @file capi_sssp.hip  grx_sssp == gunrock::sssp::run (reference algorithms/sssp.hxx:155-185). */
#include "capi_internal.hxx"
#include "clients.hxx"

using namespace essentials_amd;

extern "C" int grx_sssp(grx_context_t ctx, grx_graph_t g, int32_t source, float* d_distances,
                        int32_t* /*d_predecessors*/, const grx_options* opt, grx_stats* stats) {
  if (!ctx || !g || !d_distances)
    return invalid("grx_sssp: NULL argument");
  if (source < 0 || source >= g->n_rows)
    return invalid("grx_sssp: source out of range");
  grx_options o;
  grx_default_options(&o);
  if (opt)
    o = *opt;
  return guarded([&] {
    return with_load_balance(o.load_balance, [&](auto lb_tag) -> int {
      constexpr auto lb = decltype(lb_tag)::value;
      using problem_type = clients::sssp_problem_t<graph_type>;
      using enactor_type = clients::sssp_enactor_t<problem_type, lb>;
      scoped_options scope(ctx->single(), &o);
      graph_type G = g->view();
      problem_type problem(G, source, d_distances, ctx->mc);
      problem.packed_labels = o.sssp_two_pass == 0;  // the reference's formulation keeps its two arrays
      problem.init();
      problem.reset();
      enactor_properties_t props;
      if (o.frontier_sizing_factor > 0)
        props.frontier_sizing_factor = o.frontier_sizing_factor;
      enactor_type enactor(&problem, ctx->mc, props);
      enactor.max_iterations = o.max_iterations;
      enactor.two_pass = o.sssp_two_pass != 0;
      const float ms = enactor.enact();
      if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->elapsed_ms = ms;
        stats->iterations = enactor.iteration;
        stats->advance_kernel_ms = ctx->single().kernel_clock().total_ms;
        stats->advance_launches = ctx->single().kernel_clock().launches;
        stats->levels_recorded = problem.log.levels < 64 ? problem.log.levels : 64;
        for (int i = 0; i < stats->levels_recorded; ++i)
          stats->frontier_slots[i] = problem.log.input_slots[i];
        // the first frontier ({source}) carries no work hint: add the source's own degree
        stats->edges_expanded =
            problem.log.edges_expanded + reach_stats(g, d_distances, FLT_MAX, source, ctx->single(), stats);
      }
      return (int)GRX_OK;
    });
  });
}
