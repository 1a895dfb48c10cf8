/**
 * @file reduce_kernels.hxx
 * @brief Pull-side neighbour reduction over a CSR view of IN-edges:
 *        y[v] = base + sum over in-edges (u -> v) of x[u] * w(u -> v).
 *
 * This is what a pull PageRank iteration needs (the reference scatters with one float atomic per
 * edge, algorithms/pr.hxx:140-146; its segmented-reduce operator, operators/neighborreduce/
 * neighborreduce.hxx:55-101, sits on ModernGPU's transform_segreduce and is not used by pr.hxx).
 * No atomics on the common path:
 *   - rows shorter than RED_HUB (256): RED_GROUP (16) lanes of a wavefront share one row -- an
 *     R-MAT row has ~16 in-edges, so a group reads one 64-B column segment per step, at most 16
 *     steps -- and finish with a 4-step shuffle reduction; four rows per 64-lane wavefront;
 *   - longer rows (an RMAT-24 hub has > 10^5 in-edges) are cut on the host, once per graph, into
 *     chunks of RED_CHUNK edges; a whole workgroup sums a chunk and adds it with ONE atomic.
 */
#pragma once

#include <gunrock/hip/primitives.hxx>

namespace gunrock {
namespace hip {
namespace kernels {

constexpr int RED_BLOCK = 256;
constexpr int RED_GROUP = 16;        // lanes per row
constexpr unsigned RED_HUB = 256;    // rows at least this long go to hub_chunk_sum_kernel ...
constexpr unsigned RED_CHUNK = 2048; // ... in chunks of this many edges

template <typename vertex_t, typename edge_t>
struct row_chunk_t {
  vertex_t row;
  int count;
  edge_t first;
};

template <typename view_t, typename weight_t>
__global__ void __launch_bounds__(RED_BLOCK)
    row_group_sum_kernel(view_t in, const weight_t* __restrict__ x, weight_t base,
                         weight_t* __restrict__ y) {
  using vertex_t = typename view_t::vertex_type;
  using edge_t = typename view_t::edge_type;
  const std::size_t n = (std::size_t)in.get_number_of_vertices();
  const int lane = threadIdx.x & (RED_GROUP - 1);
  const std::size_t groups = (std::size_t)gridDim.x * (RED_BLOCK / RED_GROUP);
  // every wavefront makes the same number of trips: the shuffles below need all 64 lanes
  const std::size_t trips = (n + groups - 1) / groups;
  std::size_t row = ((std::size_t)blockIdx.x * RED_BLOCK + threadIdx.x) / RED_GROUP;
  for (std::size_t t = 0; t < trips; ++t, row += groups) {
    weight_t sum = 0;
    bool small = false;
    if (row < n) {
      const edge_t first = in.get_starting_edge((vertex_t)row);
      const unsigned deg = (unsigned)(in.get_starting_edge((vertex_t)(row + 1)) - first);
      small = deg < RED_HUB;
      if (small)
        for (unsigned j = lane; j < deg; j += RED_GROUP) {
          const edge_t e = first + (edge_t)j;
          sum += x[in.get_destination_vertex(e)] * in.get_edge_weight(e);
        }
    }
#pragma unroll
    for (int d = RED_GROUP / 2; d > 0; d >>= 1)
      sum += __shfl_xor(sum, d, RED_GROUP);
    if (row < n && lane == 0)
      y[row] = small ? base + sum : base;  // hub rows receive their chunks' sums atomically
  }
}

template <typename view_t, typename weight_t, typename vertex_t, typename edge_t>
__global__ void __launch_bounds__(RED_BLOCK)
    hub_chunk_sum_kernel(view_t in, const weight_t* __restrict__ x,
                         const row_chunk_t<vertex_t, edge_t>* __restrict__ chunks,
                         std::size_t n_chunks, weight_t* y) {
  __shared__ weight_t s_part[RED_BLOCK / wave_size];
  for (std::size_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const row_chunk_t<vertex_t, edge_t> d = chunks[c];
    weight_t sum = 0;
    for (int j = threadIdx.x; j < d.count; j += RED_BLOCK) {
      const edge_t e = d.first + (edge_t)j;
      sum += x[in.get_destination_vertex(e)] * in.get_edge_weight(e);
    }
    sum = wave_sum(sum);
    if (lane_id() == 0)
      s_part[threadIdx.x / wave_size] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
      weight_t total = 0;
#pragma unroll
      for (int w = 0; w < RED_BLOCK / wave_size; ++w)
        total += s_part[w];
      atomicAdd(&y[d.row], total);
    }
    __syncthreads();  // s_part is rewritten by the next chunk
  }
}

}  // namespace kernels
}  // namespace hip
}  // namespace gunrock
