/**
 * @file capi_operators.hip
 * @brief C ABI: frontier-level operator calls with built-in functors
 * (advance.hxx:91-129, filter.hxx:59-86, uniquify.hxx:15-42 of the reference).
 * Frontiers wrap caller-owned device arrays for the duration of the call.
 */
#include "capi_internal.hxx"

using namespace essentials_amd;

namespace {

/// One functor for every grx_edge_op (run-time switch: this path is for tests).
struct edge_functor_t {
  int kind;
  void* state;
  int iparam;
  __host__ __device__ bool operator()(vertex_t const& src, vertex_t const& dst, edge_t const& edge,
                                      weight_t const& w) const {
    switch (kind) {
      case GRX_OP_ALL:
        return true;
      case GRX_OP_BFS: {
        int* depth = static_cast<int*>(state);
        return iparam + 1 < math::atomic::min(&depth[dst], iparam + 1);
      }
      case GRX_OP_SSSP: {
        float* dist = static_cast<float*>(state);
        float d = thread::load(&dist[src]) + w;
        return d < math::atomic::min(&dist[dst], d);
      }
      case GRX_OP_COUNT_EDGE: {
        int* calls = static_cast<int*>(state);
        math::atomic::add(&calls[edge], 1);
        return (src + dst) % 3 == 0;
      }
      case GRX_OP_SUM_WEIGHT: {
        float* acc = static_cast<float*>(state);
        math::atomic::add(&acc[dst], w);
        return false;
      }
    }
    return false;
  }
};

struct vertex_functor_t {
  int kind;
  void* state;
  int iparam;
  __host__ __device__ bool operator()(vertex_t const& v) const {
    switch (kind) {
      case GRX_PRED_ALL:
        return true;
      case GRX_PRED_ODD:
        return (v & 1) != 0;
      case GRX_PRED_ONCE: {
        int* stamp = static_cast<int*>(state);
        if (stamp[v] == iparam)
          return false;
        stamp[v] = iparam;
        return true;
      }
      case GRX_PRED_COUNT: {
        int* calls = static_cast<int*>(state);
        math::atomic::add(&calls[v], 1);
        return v % 3 != 0;
      }
    }
    return false;
  }
};

/// A frontier over caller-owned device memory: stage in/out through an engine frontier.
struct staged_frontier_t {
  frontier_type f;
  void load(const int32_t* d_src, std::size_t n, hipStream_t s) {
    f.reserve(n ? n : 1);
    if (n)
      GRX_HIP_CHECK(hipMemcpyAsync(f.data(), d_src, n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    f.set_number_of_elements(n);
  }
  void store(int32_t* d_dst, std::size_t capacity, int64_t* n_out, hipStream_t s) {
    const std::size_t n = f.get_number_of_elements();
    error::throw_if_exception(n > capacity, "output_capacity too small");
    if (n)
      GRX_HIP_CHECK(hipMemcpyAsync(d_dst, f.data(), n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    GRX_HIP_CHECK(hipStreamSynchronize(s));
    if (n_out)
      *n_out = (int64_t)n;
  }
};

template <operators::load_balance_t lb, operators::advance_io_type_t in, operators::advance_io_type_t out>
void advance_as(graph_type& G, edge_functor_t op, frontier_type* fin, frontier_type* fout,
                hip::device_array_t<edge_t>& segments, gcuda::multi_context_t& mc) {
  operators::advance::execute<lb, operators::advance_direction_t::forward, in, out>(
      G, op, fin, fout, segments, mc);
}

}  // namespace

extern "C" {

int grx_advance(grx_context_t ctx, grx_graph_t g, const grx_options* opt, int32_t edge_op,
                void* d_state, int32_t iparam, const int32_t* d_input, int64_t n_input,
                int32_t* d_output, int64_t output_capacity, int64_t* n_output) {
  using operators::advance_io_type_t;
  if (!ctx || !g)
    return invalid("grx_advance: NULL argument");
  if (edge_op < GRX_OP_ALL || edge_op > GRX_OP_SUM_WEIGHT)
    return invalid("grx_advance: unknown edge_op");
  if (edge_op != GRX_OP_ALL && !d_state)
    return invalid("grx_advance: this edge_op needs d_state");
  if (d_input && n_input < 0)
    return invalid("grx_advance: negative n_input");
  grx_options o;
  grx_default_options(&o);
  if (opt)
    o = *opt;
  return guarded([&] {
    return with_load_balance(o.load_balance, [&](auto lb_tag) -> int {
      constexpr auto lb = decltype(lb_tag)::value;
      auto& sc = ctx->single();
      scoped_options scope(sc, &o);
      graph_type G = g->view();
      edge_functor_t op{edge_op, d_state, iparam};
      staged_frontier_t fin, fout;
      hip::device_array_t<edge_t> segments;
      if (d_input)
        fin.load(d_input, (std::size_t)n_input, sc.stream());
      if (d_output)
        fout.f.reserve(output_capacity > 0 ? (std::size_t)output_capacity : 1);
      if (d_input && d_output)
        advance_as<lb, advance_io_type_t::vertices, advance_io_type_t::vertices>(G, op, &fin.f, &fout.f, segments, *ctx->mc);
      else if (d_input)
        advance_as<lb, advance_io_type_t::vertices, advance_io_type_t::none>(G, op, &fin.f, &fout.f, segments, *ctx->mc);
      else if (d_output)
        advance_as<lb, advance_io_type_t::graph, advance_io_type_t::vertices>(G, op, &fin.f, &fout.f, segments, *ctx->mc);
      else
        advance_as<lb, advance_io_type_t::graph, advance_io_type_t::none>(G, op, &fin.f, &fout.f, segments, *ctx->mc);
      if (d_output)
        fout.store(d_output, (std::size_t)output_capacity, n_output, sc.stream());
      else if (n_output)
        *n_output = 0;
      return (int)GRX_OK;
    });
  });
}

int grx_filter(grx_context_t ctx, grx_graph_t g, int32_t algorithm, int32_t vertex_op,
               void* d_state, int32_t iparam, const int32_t* d_input, int64_t n_input,
               int32_t* d_output, int64_t output_capacity, int64_t* n_output) {
  using operators::filter_algorithm_t;
  if (!ctx || !g || (n_input && !d_input) || !d_output || n_input < 0)
    return invalid("grx_filter: bad argument");
  if (vertex_op < GRX_PRED_ALL || vertex_op > GRX_PRED_COUNT)
    return invalid("grx_filter: unknown vertex_op");
  if (vertex_op >= GRX_PRED_ONCE && !d_state)
    return invalid("grx_filter: this vertex_op needs d_state");
  return guarded([&] {
    auto& sc = ctx->single();
    graph_type G = g->view();
    vertex_functor_t op{vertex_op, d_state, iparam};
    staged_frontier_t fin, fout;
    fin.load(d_input, (std::size_t)n_input, sc.stream());
    fout.f.reserve(n_input ? (std::size_t)n_input : 1);
    switch (algorithm) {
      case GRX_FILTER_REMOVE:
        operators::filter::execute<filter_algorithm_t::remove>(G, op, &fin.f, &fout.f, *ctx->mc);
        break;
      case GRX_FILTER_PREDICATED:
        operators::filter::execute<filter_algorithm_t::predicated>(G, op, &fin.f, &fout.f, *ctx->mc);
        break;
      case GRX_FILTER_COMPACT:
        operators::filter::execute<filter_algorithm_t::compact>(G, op, &fin.f, &fout.f, *ctx->mc);
        break;
      case GRX_FILTER_BYPASS:
        operators::filter::execute<filter_algorithm_t::bypass>(G, op, &fin.f, &fout.f, *ctx->mc);
        break;
      default:
        return unsupported("Filter type not supported.");
    }
    fout.store(d_output, (std::size_t)output_capacity, n_output, sc.stream());
    return (int)GRX_OK;
  });
}

int grx_uniquify(grx_context_t ctx, int32_t algorithm, int32_t best_effort, int32_t* d_input,
                 int64_t n_input, int32_t* d_output, int64_t output_capacity, int64_t* n_output) {
  using operators::uniquify_algorithm_t;
  if (!ctx || (n_input && !d_input) || n_input < 0)
    return invalid("grx_uniquify: bad argument");
  if (algorithm == GRX_UNIQUE_COPY && !d_output)
    return invalid("grx_uniquify: unique_copy needs d_output");
  return guarded([&] {
    auto& sc = ctx->single();
    staged_frontier_t fin, fout;
    fin.load(d_input, (std::size_t)n_input, sc.stream());
    fout.f.reserve(n_input ? (std::size_t)n_input : 1);
    if (algorithm == GRX_UNIQUE) {
      operators::uniquify::execute<uniquify_algorithm_t::unique>(&fin.f, &fout.f, *ctx->mc,
                                                                 best_effort != 0);
      fin.store(d_input, (std::size_t)n_input, n_output, sc.stream());
    } else if (algorithm == GRX_UNIQUE_COPY) {
      operators::uniquify::execute<uniquify_algorithm_t::unique_copy>(&fin.f, &fout.f, *ctx->mc,
                                                                      best_effort != 0);
      fout.store(d_output, (std::size_t)output_capacity, n_output, sc.stream());
    } else {
      return unsupported("Unqiue type not supported.");
    }
    return (int)GRX_OK;
  });
}

}  // extern "C"
