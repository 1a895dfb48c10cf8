/**
 * @file settled.hxx
 * @brief "Settled destinations": an optional, one-sided hint a client may attach to its advance
 * functor (engine extension; the reference has no counterpart -- its advance calls the functor for
 * every edge, framework/operators/advance/block_mapped.hxx:30-147).
 *
 * Contract.  A destination v is SETTLED when the client's functor is known to return false for it
 * and to change nothing (BFS: v already has a depth).  The client names settled destinations in
 * two forms, either of which may be stale or partial -- "not settled" promises nothing:
 *   - a bitmap, one bit per vertex id below `limit` (settled_filter_t), and / or
 *   - a PURE device predicate (plain loads, no side effects), either of the destination alone,
 *     `settled(v)`, or of the whole edge, `rejects(src, dst, edge, weight)` -- "the functor would
 *     return false for THIS edge and change nothing" (SSSP: dist[dst] <= dist[src] + w),
 * and the engine is then free -- not obliged -- to skip the functor call for such an edge.
 *
 * Why.  A wide level is bound by what happens per edge inside the opaque functor: a label lookup
 * (one L2 request per edge) and, where it passes, a memory-side atomic whose round trip every lane
 * of the wavefront waits for; the four calls a lane has in flight are serialised by the compiler
 * (DESIGN.md section 5).  With the hint, expand_settled_kernel
 *   1. answers the lookups of the low -- on power-law graphs: the hot -- ids from a copy of the
 *      bitmap in LDS (tools/lds_filter_probe.hip: 175 -> 340 G lookups/s on R-MAT-22 ids),
 *   2. evaluates the predicate for the rest with four independent loads in flight per lane,
 *   3. packs the surviving edges of a wavefront and calls the functor on full groups of 64 only.
 *
 *   operators::advance::settled_filter_t<vertex_t> settled;          // enactor member
 *   auto has_depth = [depth] __device__(vertex_t v) { return depth[v] != unvisited; };
 *   settled.rebuild(n_vertices, has_depth, ctx);
 *   operators::advance::execute<lb>(
 *       G, E, operators::advance::with_settled(visit, settled.view(), has_depth), context);
 *   // SSSP, per edge:  with_rejects<vertex_t>(relax, [dist] __device__(src, dst, e, w) { return !(dist[src] + w < dist[dst]); })
 *
 * Schedules other than block_mapped's wide-level form call the functor for every edge as before.
 */
#pragma once

#include <cstddef>
#include <type_traits>

#include <gunrock/hip/context.hxx>

namespace gunrock {
namespace operators {
namespace advance {

/// Ids covered by the LDS filter: 96 KB of bits, what fits beside the kernel's own LDS in a CU's
/// 160 KB.
constexpr std::size_t settled_max_ids = std::size_t(96) << 13;

template <typename vertex_t>
struct settled_view_t {
  const unsigned* bits = nullptr;  ///< bit v of word v / 32, 16-byte aligned
  vertex_t limit = 0;              ///< ids >= limit are not covered; a multiple of 128
};

/// The predicate of a client that has a bitmap only.
struct settled_never_t {
  template <typename vertex_t>
  __host__ __device__ __forceinline__ bool operator()(vertex_t const&) const {
    return false;
  }
};

/// A functor with a settled view and predicate attached.  Callable like the functor itself.
/// per_edge: the predicate takes (src, dst, edge, weight) instead of (dst).
template <typename op_t, typename pred_t, typename vertex_t, bool per_edge = false>
struct settled_op_t {
  static constexpr bool has_predicate = !std::is_same<pred_t, settled_never_t>::value;
  static constexpr bool per_edge_predicate = per_edge;
  static constexpr int lds_image = 1;  ///< what the workgroup keeps in LDS: 1 = the settled bitmap
  __host__ __device__ std::size_t lds_bytes() const { return (std::size_t)settled.limit / 8; }
  op_t op;
  settled_view_t<vertex_t> settled;
  pred_t is_settled;
  /// The predicate in whichever form the client wrote it.
  template <typename edge_t, typename weight_t>
  __device__ __forceinline__ bool rejects(vertex_t const& src, vertex_t const& dst, edge_t const& edge,
                                          weight_t const& weight) const {
    if constexpr (per_edge)
      return is_settled(src, dst, edge, weight);
    else
      return is_settled(dst);
  }
  template <typename... args_t>
  __host__ __device__ __forceinline__ bool operator()(args_t const&... args) const {
    return op(args...);
  }
};

/// Settled DESTINATIONS: a bitmap and / or a pure predicate of the destination vertex.
template <typename op_t, typename vertex_t, typename pred_t = settled_never_t>
settled_op_t<op_t, pred_t, vertex_t, false> with_settled(op_t op,
                                                         settled_view_t<vertex_t> view,
                                                         pred_t pred = pred_t()) {
  return settled_op_t<op_t, pred_t, vertex_t, false>{op, view, pred};
}

/// Rejected EDGES: a pure predicate of (src, dst, edge, weight).
template <typename vertex_t, typename op_t, typename pred_t>
settled_op_t<op_t, pred_t, vertex_t, true> with_rejects(op_t op, pred_t pred) {
  return settled_op_t<op_t, pred_t, vertex_t, true>{op, settled_view_t<vertex_t>{}, pred};
}

/// Cached BOUNDS (engine extension, per-edge form).  `values[v]`, 2 bytes per id below `limit`, is
/// some monotone summary of the client's label of v as it stood before the advance (SSSP: an upper
/// bound of the distance); the workgroup keeps a copy in LDS.  `cached(src, dst, edge, weight,
/// values[dst])` is PURE and says "the functor would return false for this edge and change nothing"
/// from that summary alone -- for ids below `limit` it is the whole test (no memory request for the
/// destination); ids >= limit are asked through the global per-edge predicate `is_settled`.
template <typename op_t, typename pred_t, typename cached_t, typename vertex_t>
struct bounded_op_t {
  static constexpr bool has_predicate = true;
  static constexpr bool per_edge_predicate = true;
  static constexpr int lds_image = 2;  ///< 2-byte values
  __host__ __device__ std::size_t lds_bytes() const { return (std::size_t)settled.limit * 2; }
  op_t op;
  settled_view_t<vertex_t> settled;  ///< bits = the 2-byte values, limit = ids covered (multiple of 8)
  pred_t is_settled;
  cached_t cached;
  template <typename edge_t, typename weight_t>
  __device__ __forceinline__ bool rejects(vertex_t const& src, vertex_t const& dst, edge_t const& edge,
                                          weight_t const& weight) const {
    return is_settled(src, dst, edge, weight);
  }
  template <typename... args_t>
  __host__ __device__ __forceinline__ bool operator()(args_t const&... args) const {
    return op(args...);
  }
};

/// Ids a workgroup's LDS can hold 2-byte values for (the same 96 KB as the bitmap).
constexpr std::size_t bounded_max_ids = (std::size_t(96) << 10) / 2;

template <typename vertex_t, typename op_t, typename pred_t, typename cached_t>
bounded_op_t<op_t, pred_t, cached_t, vertex_t> with_bounds(op_t op, pred_t pred, cached_t cached,
                                                           const unsigned short* values,
                                                           std::size_t n_values) {
  std::size_t ids = n_values < bounded_max_ids ? n_values : bounded_max_ids;
  ids = ids / 8 * 8;  // whole 16-byte groups
  return bounded_op_t<op_t, pred_t, cached_t, vertex_t>{
      op, settled_view_t<vertex_t>{reinterpret_cast<const unsigned*>(values), (vertex_t)ids}, pred, cached};
}

template <typename T>
struct settled_traits : std::false_type {};
template <typename op_t, typename pred_t, typename cached_t, typename vertex_t>
struct settled_traits<bounded_op_t<op_t, pred_t, cached_t, vertex_t>> : std::true_type {};
template <typename op_t, typename pred_t, typename vertex_t, bool per_edge>
struct settled_traits<settled_op_t<op_t, pred_t, vertex_t, per_edge>> : std::true_type {};

namespace detail_settled {
/// One wavefront per 64 ids: the ballot of the predicate IS the two bitmap words.
template <typename vertex_t, typename pred_t>
__global__ void __launch_bounds__(256)
    rebuild_kernel(unsigned long long* words, vertex_t limit, vertex_t n_vertices, pred_t pred) {
  const std::size_t stride = (std::size_t)gridDim.x * 256;
  for (std::size_t i = (std::size_t)blockIdx.x * 256 + threadIdx.x; i < (std::size_t)limit; i += stride) {
    const bool set = i < (std::size_t)n_vertices && pred((vertex_t)i);
    const unsigned long long word = __ballot(set);
    if ((threadIdx.x & 63) == 0)
      words[i / 64] = word;
  }
}
}  // namespace detail_settled

/// Owner of the bitmap.  rebuild() is one small kernel (reads `limit` labels, writes limit / 8
/// bytes) on the context's stream; call it before an advance that is worth it (a wide frontier).
template <typename vertex_t>
class settled_filter_t {
 public:
  template <typename pred_t>
  void rebuild(std::size_t n_vertices, pred_t pred, gcuda::standard_context_t& context) {
    std::size_t ids = n_vertices < settled_max_ids ? n_vertices : settled_max_ids;
    ids = (ids + 127) / 128 * 128;  // whole 16-byte groups; bits of ids >= n_vertices stay clear
    if (words_.size() < ids / 64)
      words_.resize(ids / 64);
    limit_ = (vertex_t)ids;
    const unsigned grid = (unsigned)((ids + 255) / 256 < 1024 ? (ids + 255) / 256 : 1024);
    detail_settled::rebuild_kernel<<<grid, 256, 0, context.stream()>>>(words_.data(), limit_,
                                                                      (vertex_t)n_vertices, pred);
    GRX_HIP_CHECK(hipGetLastError());
  }
  settled_view_t<vertex_t> view() const {
    return settled_view_t<vertex_t>{reinterpret_cast<const unsigned*>(words_.data()), limit_};
  }
  void clear() { limit_ = 0; }
  /// Another pass is about to write the bitmap (operators::filter::select_range with a bit
  /// predicate): storage for `n_vertices`, returns the word array; `limit` ids are covered.  The
  /// next rebuild() is then skipped once (the bits are those rebuild() would compute now).
  unsigned long long* prepare(std::size_t n_vertices, std::size_t& limit) {
    std::size_t ids = n_vertices < settled_max_ids ? n_vertices : settled_max_ids;
    ids = (ids + 127) / 128 * 128;
    if (words_.size() < ids / 64)
      words_.resize(ids / 64);
    limit_ = (vertex_t)ids;
    limit = ids;
    fresh_ = true;
    return words_.data();
  }
  /// rebuild() unless a prepare()d pass has just written the bits.
  template <typename pred_t>
  void refresh(std::size_t n_vertices, pred_t pred, gcuda::standard_context_t& context) {
    if (fresh_) {
      fresh_ = false;
      return;
    }
    rebuild(n_vertices, pred, context);
  }

 private:
  hip::device_array_t<unsigned long long> words_;
  vertex_t limit_ = 0;
  bool fresh_ = false;
};

}  // namespace advance
}  // namespace operators
}  // namespace gunrock
