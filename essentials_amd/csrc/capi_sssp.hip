/** @file capi_sssp.hip  grx_sssp == gunrock::sssp::run (reference algorithms/sssp.hxx:155-185). */
#include "capi_internal.hxx"
#include "clients.hxx"

using namespace essentials_amd;

namespace essentials_amd {
/// Largest |V| for which grx_sssp keeps packed 64-bit labels (GRX_SSSP_PACKED_MAX_VERTICES overrides).
inline long long packed_sssp_max_vertices() {
  if (const char* e = std::getenv("GRX_SSSP_PACKED_MAX_VERTICES"))
    return std::atoll(e);
  return 1ll << 22;  // 32 MB of labels: what the eight 4 MB L2s hold between them
}
}  // namespace essentials_amd

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
      // hot-first renumbered copy, distances delivered in the caller's numbering (see grx_bfs); the
      // reference's two-pass formulation and the every-edge form keep the caller's graph
      grx_graph_s* run_on = g;
      if (!o.sssp_two_pass && !o.call_every_edge && !o.holes_layout)
        if (grx_graph_s* h = hot_copy(ctx, g))
          run_on = h;
      graph_type G = run_on->view();
      problem_type problem(G, run_on == g ? source : g->hot_rank_of[(std::size_t)source], d_distances,
                           ctx->mc);
      if (run_on != g)
        problem.scatter_to = g->hot_vertex_of.data();
        problem.gather_from = g->hot_rank_of_device.data();
      // one 64-bit label per vertex (one RMW per improvement) while 8 bytes per vertex stay
      // cache-sized; beyond that the doubled label footprint costs more lookups that miss than the
      // saved RMWs are worth.  Measured crossover on R-MAT (tools/sssp_packed_vs_words.py, mean
      // enact of 3 sources, packed / two words): scale 20 1.25 / 1.42 ms, 22 2.92 / 3.11, 23 5.56 /
      // 5.46, 24 12.8 / 11.0, 26 58.2 / 53.5.  GRX_SSSP_PACKED=0/1 overrides; the reference's
      // two-pass formulation keeps its two arrays.
      bool packed = g->n_rows <= essentials_amd::packed_sssp_max_vertices();
      if (const char* e = std::getenv("GRX_SSSP_PACKED"))
        packed = std::atoi(e) != 0;
      problem.packed_labels = packed && o.sssp_two_pass == 0;
      // the unpacking pass at the end of a packed run also counts what the run reached
      problem.collect_reach = stats != nullptr && problem.packed_labels;
      problem.init();
      problem.reset();
      enactor_properties_t props;
      if (o.frontier_sizing_factor > 0)
        props.frontier_sizing_factor = o.frontier_sizing_factor;
      enactor_type enactor(&problem, ctx->mc, props);
      enactor.max_iterations = o.max_iterations;
      enactor.two_pass = o.sssp_two_pass != 0;
      enactor.bound_filter = o.call_every_edge == 0;
      if (const char* e = std::getenv("GRX_SSSP_BOUND_FILTER"))
        enactor.bound_filter = std::atoi(e) != 0;
      if (const char* e = std::getenv("GRX_SSSP_EARLY_LIVE"))
        enactor.early_live = std::atoi(e) != 0;
      if (const char* e = std::getenv("GRX_SSSP_BOUND_FROM"))
        enactor.bound_from = std::atoi(e);
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
        const unsigned long long* facts = ctx->single().workspace().run_facts();
        if (problem.collect_reach && facts[3] == 1ull) {  // left in pinned memory by unpack()
          stats->vertices_reached = (int64_t)facts[1];
          stats->edges_traversed = (int64_t)facts[2];
          stats->edges_expanded = problem.log.edges_expanded + (long long)facts[0];
        } else {
          stats->edges_expanded =
              problem.log.edges_expanded + reach_stats(g, d_distances, FLT_MAX, source, ctx->single(), stats);
        }
      }
      return (int)GRX_OK;
    });
  });
}
