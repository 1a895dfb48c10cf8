/**
 * @file configs.hxx
 * @brief Compile-time selectors of the operators: the vocabulary client code spells, e.g.
 * `operators::load_balance_t::block_mapped` (algorithms/bfs.hxx:125).  The enumerator NAMES are
 * the reference's (framework/operators/configs.hxx:31-92); they are scoped enums here so that
 * `graph`, `remove`, `vertex`, ... cannot collide with namespaces of the same name.
 *
 * What each advance schedule means in THIS engine (kernels: gunrock/hip/kernels/advance_kernels.hxx)
 *   block_mapped   persistent 256-thread workgroups own tiles of 256 input slots, stage (vertex,
 *                  first edge, scanned degree) in LDS and stride the tile's concatenated lists;
 *                  lists >= hub_threshold are cut into chunks spread over the whole GPU
 *   work_stealing  the same, tiles claimed dynamically from a device counter
 *   merge_path     device-wide scan of degrees, then equal shares of EDGES per workgroup step
 *   merge_path_v2  alias of merge_path (the reference's variant is unfinished:
 *                  advance/merge_path_v2.hxx:166-175,221)
 *   bucketing      slots binned by degree into thread / wavefront / chunk queues (the reference
 *                  names it but leaves it empty: advance/bucketing.hxx:24-36)
 *   thread_mapped  one lane walks one input slot's list
 *   warp_mapped    one 64-lane wavefront strides one list
 * Directions: forward = push along out-edges; backward = pull (candidates in, see advance.hxx);
 * optimized is the client's choice per iteration (direction-optimising BFS in clients.hxx).
 */
#pragma once

namespace gunrock {
namespace operators {

// -- advance ---------------------------------------------------------------------------------
enum class advance_direction_t : int { forward = 0, backward = 1, optimized = 2 };
enum class advance_io_type_t : int { graph = 0, vertices = 1, edges = 2, none = 3 };
enum class load_balance_t : int {
  thread_mapped = 0, warp_mapped = 1, block_mapped = 2, bucketing = 3,
  merge_path = 4, merge_path_v2 = 5, work_stealing = 6
};

// -- filter / uniquify -----------------------------------------------------------------------
// remove / predicated / compact: stable copy of what the predicate keeps (compact is the
// two-pass ballot form); bypass: same length, rejected elements become invalid.
enum class filter_algorithm_t : int { remove = 0, predicated = 1, compact = 2, bypass = 3 };
// unique: in place; unique_copy: into the output frontier.
enum class uniquify_algorithm_t : int { unique = 0, unique_copy = 1 };

// -- parallel_for ----------------------------------------------------------------------------
enum class parallel_for_each_t : int { vertex = 0, edge = 1, weight = 2, element = 3 };

}  // namespace operators
}  // namespace gunrock
