/**
 * @file configs.hxx
 * @brief Compile-time selectors of the operators (the API vocabulary clients
 * spell, e.g. operators::load_balance_t::block_mapped in algorithms/bfs.hxx:125).
 * Same enumerators as reference framework/operators/configs.hxx:31-92; scoped
 * here so they cannot collide with the `graph`/`remove`/... namespaces.
 *
 * What each load-balancing schedule means in THIS engine (gfx950 kernels in
 * gunrock/hip/kernels/advance_kernels.hxx):
 *   thread_mapped  one lane walks one input slot's neighbour list
 *   warp_mapped    one 64-lane wavefront strides one neighbour list (coalesced)
 *   block_mapped   a 256-thread workgroup owns a tile of 256 input slots, stages
 *                  (vertex, first edge, scanned degree) in LDS and strides the
 *                  tile's concatenated neighbour lists; lists >= a hub threshold
 *                  are cut into equal chunks and spread over the whole GPU
 *   bucketing      input slots are binned by degree into thread / wavefront /
 *                  workgroup-chunk queues, one schedule per bin
 *   merge_path     device-wide scan of degrees, then every workgroup takes an
 *                  equal share of EDGES (slot found by binary search)
 *   merge_path_v2  alias of merge_path (the reference's second variant is
 *                  unfinished: advance/merge_path_v2.hxx:166-175,221)
 *   work_stealing  block_mapped tiles claimed dynamically from a device counter
 */
#pragma once

namespace gunrock {
namespace operators {

enum class load_balance_t {
  thread_mapped,
  warp_mapped,
  block_mapped,
  bucketing,
  merge_path,
  merge_path_v2,
  work_stealing
};

enum class advance_io_type_t {
  graph,     ///< every vertex of the graph is the input frontier
  vertices,  ///< vertex frontier
  edges,     ///< edge frontier
  none       ///< no output frontier is produced
};

enum class advance_direction_t {
  forward,   ///< push along out-edges
  backward,  ///< pull along in-edges
  optimized  ///< switch per iteration
};

enum class filter_algorithm_t {
  remove,      ///< stable copy of the elements the predicate keeps
  predicated,  ///< stable copy of the elements the predicate keeps
  compact,     ///< two-pass ballot compaction (count, then place)
  bypass       ///< same length, rejected elements become invalid
};

enum class uniquify_algorithm_t {
  unique,      ///< in place
  unique_copy  ///< into the output frontier
};

enum class parallel_for_each_t { vertex, edge, weight, element };

}  // namespace operators
}  // namespace gunrock
