/**
 * @file bsp.hxx
 * @brief The two base classes of an algorithm in the bulk-synchronous model: problem_t (the data
 * slice: graph view + context + init()/reset(), reference framework/problem.hxx:29-59) and
 * enactor_t, the bulk-synchronous driver: two frontiers, a scan workspace and the
 * `while (!is_converged) { loop(); ++iteration; }` host loop.
 *
 * Same surface as reference framework/enactor.hxx:31-54 (enactor_properties_t)
 * and :78-310 (enactor_t): get_problem / get_enactor / get_input_frontier /
 * get_output_frontier / swap_frontier_buffers / enact() -> milliseconds, virtuals
 * prepare_frontier, loop, is_converged, finalize; public members iteration,
 * context, scanned_work_domain, active_frontier, inactive_frontier.
 * enact() times with events on the context's stream, around the loop only
 * (enactor.hxx:245-253).
 */
#pragma once

#include <atomic>
#include <memory>
#include <vector>

#include <gunrock/framework/frontier.hxx>
#include <gunrock/framework/partitioned.hxx>
#include <gunrock/graph/graph.hxx>
#include <gunrock/hip/context.hxx>

namespace gunrock {

/**
 * @brief Data slice of an algorithm.  Owns nothing but a COPY of the (non-owning) graph view and
 * a share of the context; concrete problems add their arrays and implement init() / reset().
 * Not copyable: clients hand out `this` to their enactor.
 */
template <typename graph_t>
struct problem_t {
  typedef typename graph_t::weight_type weight_t;
  typedef typename graph_t::edge_type edge_t;
  typedef typename graph_t::vertex_type vertex_t;

  problem_t() = default;
  problem_t(graph_t& G, std::shared_ptr<gcuda::multi_context_t> shared_context)
      : graph_slice(G), context(std::move(shared_context)) {}
  problem_t(problem_t const&) = delete;
  problem_t& operator=(problem_t const&) = delete;
  virtual ~problem_t() = default;

  /// Once per problem (allocate) / before every run (re-initialise).
  virtual void init() = 0;
  virtual void reset() = 0;

  graph_t get_graph() { return graph_slice; }  // by value, like the reference (:45)
  std::shared_ptr<gcuda::multi_context_t> get_multi_context() { return context; }
  gcuda::standard_context_t* get_single_context(gcuda::device_id_t device = 0) {
    return context->get_context(device);
  }

  graph_t graph_slice;
  std::shared_ptr<gcuda::multi_context_t> context;
};

struct enactor_properties_t {
  /// Frontier buffers are reserved to factor * max(|E|, |V|) elements up front.
  float frontier_sizing_factor{1.5f};
  std::size_t number_of_frontier_buffers{2};
  /// true: the enactor allocates no frontier storage (PageRank, pr.hxx:210-211).
  bool self_manage_frontiers{false};
};

namespace detail {
/// One counter for every enactor type of the process.
inline unsigned long long next_enactor_id() {
  static std::atomic<unsigned long long> last{0};
  return ++last;
}
}  // namespace detail

template <typename algorithm_problem_t,
          frontier::frontier_kind_t frontier_kind = frontier::frontier_kind_t::vertex_frontier,
          frontier::frontier_view_t frontier_view = frontier::frontier_view_t::vector>
struct enactor_t {
  using vertex_t = typename algorithm_problem_t::vertex_t;
  using edge_t = typename algorithm_problem_t::edge_t;
  using frontier_t = frontier::frontier_t<vertex_t, edge_t, frontier_kind, frontier_view>;

  enactor_properties_t properties;
  std::shared_ptr<gcuda::multi_context_t> context;
  algorithm_problem_t* problem;
  std::vector<frontier_t> frontiers;
  hip::device_array_t<edge_t> scanned_work_domain;
  frontier_t* active_frontier;
  frontier_t* inactive_frontier;
  int buffer_selector;
  int iteration;
  /// Never reused within the process (operators key per-run state by it, not by the enactor's address).
  const unsigned long long unique_id = detail::next_enactor_id();

  enactor_t(const enactor_t&) = delete;
  enactor_t& operator=(const enactor_t&) = delete;

  enactor_t(algorithm_problem_t* _problem,
            std::shared_ptr<gcuda::multi_context_t> _context,
            enactor_properties_t _properties = enactor_properties_t())
      : properties(_properties),
        context(std::move(_context)),
        problem(_problem),
        frontiers(properties.number_of_frontier_buffers < 2 ? 2
                                                            : properties.number_of_frontier_buffers),
        active_frontier(&frontiers[0]),
        inactive_frontier(&frontiers[1]),
        buffer_selector(0),
        iteration(0) {
    if (!properties.self_manage_frontiers) {
      auto g = problem->get_graph();
      const std::size_t e = (std::size_t)g.get_number_of_edges();
      const std::size_t v = (std::size_t)g.get_number_of_vertices();
      const std::size_t initial = e > v ? e : v;
      for (auto& f : frontiers) {
        f.set_resizing_factor(properties.frontier_sizing_factor);
        f.reserve(initial);
      }
    }
  }
  virtual ~enactor_t() = default;

  algorithm_problem_t* get_problem() { return problem; }
  enactor_t* get_enactor() { return this; }
  frontier_t* get_input_frontier() { return active_frontier; }
  frontier_t* get_output_frontier() { return inactive_frontier; }

  void swap_frontier_buffers() {
    buffer_selector ^= 1;
    active_frontier = &frontiers[buffer_selector];
    inactive_frontier = &frontiers[buffer_selector ^ 1];
  }

  /// Run to convergence; returns the milliseconds spent in the loop.  With a job attached to the
  /// context (one process per GPU) the ranks exchange their frontiers after every loop():
  /// framework/partitioned.hxx.
  float enact() {
    if (context->communicator().attached())
      return enact_partitioned();
    auto single_context = context->get_context(0);
    prepare_frontier(get_input_frontier(), *context);
    auto& timer = single_context->timer();
    timer.begin();
    while (!is_converged(*context)) {
      loop(*context);
      ++iteration;
    }
    finalize(*context);
    return timer.end();
  }

  /// enact() of one rank of a vertex-partitioned job.  The client's loop() runs unchanged on the
  /// rank's slice; what it discovered is exchanged and the owned part becomes the next input
  /// frontier.  Convergence = no rank discovered anything (is_converged() is a LOCAL test and is
  /// not consulted: a rank whose own frontier is empty must keep taking part in the collectives).
  float enact_partitioned() {
    using traits = partitioned::exchange_traits<algorithm_problem_t>;
    if constexpr (!traits::enabled || frontier_kind != frontier::frontier_kind_t::vertex_frontier) {
      error::throw_if_exception(true,
                                "this problem declares no replica combiner (partitioned::exchange_traits): "
                                "it runs as independent replicas only, not vertex-partitioned");
      return 0.0f;
    } else {
      auto single_context = context->get_context(0);
      error::throw_if_exception(properties.self_manage_frontiers,
                                "a partitioned run needs the enactor's frontiers");
      const std::size_t n = (std::size_t)problem->get_graph().get_number_of_vertices();
      prepare_frontier(get_input_frontier(), *context);
      partitioned::keep_owned(*get_input_frontier(), *context);
      auto& timer = single_context->timer();
      timer.begin();
      for (;;) {
        loop(*context);  // advance (+ filter): the input frontier now holds this rank's finds
        ++iteration;
        const unsigned long long found = partitioned::exchange(
            *get_input_frontier(), *get_output_frontier(), traits::labels(*problem), n,
            exchange_state, *context);
        swap_frontier_buffers();
        if (found == 0)
          break;
      }
      finalize(*context);
      return timer.end();
    }
  }
  partitioned::exchange_state_t exchange_state;

  virtual void loop(gcuda::multi_context_t& context) = 0;
  virtual void prepare_frontier(frontier_t*, gcuda::multi_context_t&) {}
  virtual bool is_converged(gcuda::multi_context_t&) { return active_frontier->is_empty(); }
  virtual void finalize(gcuda::multi_context_t&) {}
};

}  // namespace gunrock
