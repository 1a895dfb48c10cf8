/**
 * @file graph.hxx
 * @brief Non-owning graph views handed BY VALUE to kernels and client lambdas.
 *
 * Surface of reference include/gunrock/graph/graph.hxx:52-317,
 * graph/csr.hxx:31-234, graph/properties.hxx:19-44 and graph/build.hxx:26-52:
 * graph::graph_t<space, vertex_t, edge_t, weight_t, views...> with
 * get_number_of_vertices/edges, get_starting_edge, get_number_of_neighbors,
 * get_destination_vertex, get_edge_weight, get_source_vertex; built with
 * graph::build::from_csr<space, view_t::csr>(rows, cols, nnz, Ap, J, X[, I, Aj]).
 * The graph never owns memory; the caller keeps the arrays alive.
 *
 * Own design: one trivially-copyable CSR view (three pointers + two sizes, 40
 * bytes of kernel arguments) instead of the reference's variadic inheritance over
 * csr/csc/coo views.  HBM layout is the plain CSR triple: row_offsets[V+1]
 * (edge_t), column_indices[E] (vertex_t), values[E] (weight_t), each contiguous so
 * that a neighbour list is one coalesced run of column_indices.
 */
#pragma once

#include <gunrock/hip/runtime.hxx>
#include <gunrock/util/math.hxx>
#include <gunrock/util/type_limits.hxx>

namespace gunrock {
namespace graph {

using memory::memory_space_t;

struct graph_properties_t {
  bool directed{false};
  bool weighted{true};
  /// true: the builder KNOWS the CSR is its own transpose (informational; `directed` is what
  /// can_pull() tests).
  bool symmetric{false};
  /// Largest out-degree when the builder of the view knows it (0 = unknown).  Operators that
  /// need it otherwise reduce it once per context and remember it by (offsets pointer, |V|, |E|)
  /// -- which cannot tell apart two graphs built one after the other in the same memory, so
  /// owners of long-lived graphs should set it (the C ABI's graph handles do).
  unsigned long long max_degree{0};
  /// Hot-first renumbered graphs (graph/reorder.hxx): every vertex with an edge comes before every
  /// vertex without one, and this many vertices have edges (0 = not known / not so ordered).  A pass
  /// over "the vertices an edge can lead to" stops there.
  unsigned long long leading_connected{0};
};

enum view_t : uint32_t { invalid = 1u << 0, csr = 1u << 1, csc = 1u << 2, coo = 1u << 3 };

constexpr inline view_t operator|(view_t a, view_t b) {
  return static_cast<view_t>(static_cast<uint32_t>(a) | static_cast<uint32_t>(b));
}
constexpr inline view_t set(view_t a, view_t b) { return a | b; }
constexpr inline view_t unset(view_t a, view_t b) {
  return static_cast<view_t>(static_cast<uint32_t>(a) & ~static_cast<uint32_t>(b));
}
constexpr inline bool has(view_t a, view_t b) {
  return (static_cast<uint32_t>(a) & static_cast<uint32_t>(b)) == static_cast<uint32_t>(b);
}
constexpr inline view_t toggle(view_t a, view_t b) {
  return static_cast<view_t>(static_cast<uint32_t>(a) ^ static_cast<uint32_t>(b));
}

template <typename vertex_t>
struct vertex_pair_t {
  vertex_t source;
  vertex_t destination;
};

/// Compressed-sparse-row view: out-edges of v are column_indices[offsets[v] .. offsets[v+1]).
template <typename vertex_t, typename edge_t, typename weight_t>
class graph_csr_t {
 public:
  using vertex_type = vertex_t;
  using edge_type = edge_t;
  using weight_type = weight_t;
  using vertex_pair_type = vertex_pair_t<vertex_t>;

  __host__ __device__ graph_csr_t() {}

  void set(vertex_t const& rows, edge_t const& nnz, edge_t* Ap, vertex_t* Aj, weight_t* Ax) {
    number_of_vertices = rows;
    number_of_edges = nnz;
    offsets = Ap;
    indices = Aj;
    values = Ax;
  }

  __host__ __device__ __forceinline__ vertex_t get_number_of_vertices() const {
    return number_of_vertices;
  }
  __host__ __device__ __forceinline__ edge_t get_number_of_edges() const { return number_of_edges; }

  __host__ __device__ __forceinline__ edge_t get_starting_edge(vertex_t const& v) const {
    return offsets[v];
  }
  __host__ __device__ __forceinline__ edge_t get_number_of_neighbors(vertex_t const& v) const {
    return offsets[v + 1] - offsets[v];
  }
  __host__ __device__ __forceinline__ vertex_t get_destination_vertex(edge_t const& e) const {
#if defined(GRX_STREAM_NT) && defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load(&indices[e]);
#else
    return indices[e];
#endif
  }
  __host__ __device__ __forceinline__ weight_t get_edge_weight(edge_t const& e) const {
#if defined(GRX_STREAM_NT) && defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load(&values[e]);
#else
    return values[e];
#endif
  }

  /// Row that owns edge e: largest v with offsets[v] <= e (binary search over offsets).
  __host__ __device__ __forceinline__ vertex_t get_source_vertex(edge_t const& e) const {
    vertex_t lo = 0, hi = number_of_vertices;  // invariant: offsets[lo] <= e < offsets[hi]
    while (hi - lo > 1) {
      vertex_t mid = lo + (hi - lo) / 2;
      if (offsets[mid] <= e)
        lo = mid;
      else
        hi = mid;
    }
    return lo;
  }

  __host__ __device__ __forceinline__ vertex_pair_type
  get_source_and_destination_vertices(edge_t const& e) const {
    return {get_source_vertex(e), get_destination_vertex(e)};
  }

  /// Edge id of (source -> destination) in a row with sorted columns, or -1.
  __host__ __device__ __forceinline__ edge_t get_edge(vertex_t const& source,
                                                      vertex_t const& destination) const {
    edge_t lo = offsets[source], hi = offsets[source + 1];
    while (lo < hi) {
      edge_t mid = lo + (hi - lo) / 2;
      vertex_t c = indices[mid];
      if (c == destination)
        return mid;
      if (c < destination)
        lo = mid + 1;
      else
        hi = mid;
    }
    return static_cast<edge_t>(-1);
  }

  /**
   * @brief Common neighbours of two vertices whose rows have ASCENDING column ids:
   * on_intersection(w) is called for every w in both lists (once per matching pair of entries),
   * the number of calls is returned (reference graph/csr.hxx:110-167; triangle counting's inner
   * loop, algorithms/tc.hxx:81-94).  The shorter list is walked; in the longer one the search
   * resumes where the previous element was found (galloping binary search), so the cost is
   * O(short * log(long / short)) instead of the two-pointer walk's O(short + long).
   */
  template <typename operator_type>
  __host__ __device__ __forceinline__ vertex_t get_intersection_count(vertex_t const& source,
                                                                      vertex_t const& destination,
                                                                      operator_type on_intersection) const {
    edge_t a = offsets[source], a_end = offsets[source + 1];
    edge_t b = offsets[destination], b_end = offsets[destination + 1];
    if (a_end - a > b_end - b) {  // a = the shorter list
      edge_t t = a; a = b; b = t;
      t = a_end; a_end = b_end; b_end = t;
    }
    vertex_t count = 0;
    for (; a < a_end && b < b_end; ++a) {
      const vertex_t x = indices[a];
      // first position in [b, b_end) whose column is >= x: gallop, then bisect
      edge_t step = 1, lo = b, hi = b;
      while (hi < b_end && indices[hi] < x) {
        lo = hi + 1;
        hi += step;
        step *= 2;
      }
      if (hi > b_end)
        hi = b_end;
      while (lo < hi) {
        const edge_t mid = lo + (hi - lo) / 2;
        if (indices[mid] < x)
          lo = mid + 1;
        else
          hi = mid;
      }
      b = lo;
      if (b < b_end && indices[b] == x) {
        ++count;
        ++b;
        on_intersection(x);
      }
    }
    return count;
  }

  __host__ __device__ __forceinline__ edge_t* get_row_offsets() const { return offsets; }
  __host__ __device__ __forceinline__ vertex_t* get_column_indices() const { return indices; }
  __host__ __device__ __forceinline__ weight_t* get_nonzero_values() const { return values; }

 protected:
  vertex_t number_of_vertices = 0;
  edge_t number_of_edges = 0;
  edge_t* offsets = nullptr;
  vertex_t* indices = nullptr;
  weight_t* values = nullptr;
};

/**
 * @brief The graph type clients see.  `views` records which representations were
 * requested at build time; the CSR view is the one the advance path consumes.
 */
template <memory_space_t space, view_t views, typename vertex_t, typename edge_t, typename weight_t>
class graph_t : public graph_csr_t<vertex_t, edge_t, weight_t> {
 public:
  using vertex_type = vertex_t;
  using edge_type = edge_t;
  using weight_type = weight_t;
  using vertex_pair_type = vertex_pair_t<vertex_t>;
  using graph_csr_view_t = graph_csr_t<vertex_t, edge_t, weight_t>;

  __host__ __device__ graph_t() {}

  static constexpr memory_space_t memory_space() { return space; }
  static constexpr view_t built_views() { return views; }
  template <typename view_type = graph_csr_view_t>
  static constexpr bool contains_representation() {
    return std::is_same<view_type, graph_csr_view_t>::value && has(views, view_t::csr);
  }

  bool is_directed() const { return properties.directed; }

  // --- in-edges (the csc view of reference graph/csc.hxx, as a second CSR) -------------------
  /// Attach caller-owned arrays of the TRANSPOSE (in_offsets[V+1], in_indices[E] = sources,
  /// in_values[E]); see graph::build::transpose.  Pull advances use them.
  void attach_in_edges(edge_t* in_offsets, vertex_t* in_indices, weight_t* in_values) {
    in_view_.set(this->get_number_of_vertices(), this->get_number_of_edges(), in_offsets, in_indices,
                 in_values);
    has_in_view_ = true;
  }
  bool has_in_edges() const { return has_in_view_; }
  /// The graph whose out-edges are this graph's in-edges.  Without attached in-edges an
  /// undirected (symmetric) graph is its own transpose; a directed one has none.
  graph_csr_view_t in_edges() const {
    return has_in_view_ ? in_view_ : static_cast<graph_csr_view_t const&>(*this);
  }
  bool can_pull() const { return has_in_view_ || !properties.directed; }

  graph_properties_t properties;

 private:
  graph_csr_view_t in_view_;
  bool has_in_view_ = false;
};

namespace build {

/**
 * @brief Wrap caller-owned CSR arrays (device or host pointers, per `space`).
 * Only the CSR view is materialised; asking for csc/coo throws until the pull
 * direction lands.
 */
template <memory_space_t space, view_t build_views, typename edge_t, typename vertex_t,
          typename weight_t>
auto from_csr(vertex_t const& r, vertex_t const& c, edge_t const& nnz, edge_t* Ap, vertex_t* J,
              weight_t* X, vertex_t* I = nullptr, edge_t* Aj = nullptr) {
  (void)c; (void)I; (void)Aj;
  error::throw_if_exception(!has(build_views, view_t::csr),
                            "graph::build::from_csr: a csr view is required");
  error::throw_if_exception(has(build_views, view_t::csc) || has(build_views, view_t::coo),
                            "graph::build::from_csr: csc/coo views are not built by this engine");
  graph_t<space, build_views, vertex_t, edge_t, weight_t> G;
  G.set(r, nnz, Ap, J, X);
  return G;
}

}  // namespace build
}  // namespace graph
}  // namespace gunrock
