/**
 * @file problem.hxx
 * @brief Base class of an algorithm's data slice.
 *
 * Same surface as reference framework/problem.hxx:29-59: holds the graph view
 * BY VALUE plus the shared multi-context, exposes get_graph() (a copy),
 * get_multi_context(), get_single_context(), and the pure virtuals init()/reset();
 * not copyable.
 */
#pragma once

#include <memory>

#include <gunrock/graph/graph.hxx>
#include <gunrock/hip/context.hxx>

namespace gunrock {

template <typename graph_t>
struct problem_t {
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;

  graph_t graph_slice;
  std::shared_ptr<gcuda::multi_context_t> context;

  problem_t() = default;
  problem_t(graph_t& G, std::shared_ptr<gcuda::multi_context_t> _context)
      : graph_slice(G), context(std::move(_context)) {}
  virtual ~problem_t() = default;

  problem_t(const problem_t&) = delete;
  problem_t& operator=(const problem_t&) = delete;

  auto get_graph() { return graph_slice; }
  auto get_multi_context() { return context; }
  auto get_single_context(gcuda::device_id_t device = 0) { return context->get_context(device); }

  virtual void init() = 0;
  virtual void reset() = 0;
};

}  // namespace gunrock
