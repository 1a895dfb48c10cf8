/**
 * @file compact_kernels.hxx
 * @brief Hand-written gfx950 stream-compaction kernels behind filter / uniquify.
 *
 * Replaces thrust::transform / copy_if / remove_copy_if / unique[_copy] and
 * mgpu::transform_compact of the reference (filter/bypass.hxx:39-45,
 * predicated.hxx:29-35, remove.hxx:28-34, compact.hxx:20-36, uniquify/unique*.hxx).
 *
 * Stable two-pass compaction, 64-lane ballots as the unit:
 *   pass 1 (flag_kernel)  : every wavefront evaluates the predicate ONCE per element
 *                           for 64 consecutive elements, stores the 64-bit ballot
 *                           word and adds its popcount to the tile's count;
 *   scan                  : exclusive scan of the per-tile counts (rocPRIM);
 *   pass 2 (place_kernel) : re-reads the ballot words (the predicate is NOT called
 *                           again: client predicates have side effects,
 *                           algorithms/sssp.hxx:126-136), ranks each kept lane with
 *                           mbcnt and writes it at tile base + word prefix + rank.
 * A tile is CMP_TILE = 256 threads x 4 items = 1024 consecutive elements = 16 words.
 */
#pragma once

#include <type_traits>

#include <gunrock/hip/primitives.hxx>
#include <gunrock/util/type_limits.hxx>

namespace gunrock {
namespace hip {
namespace kernels {

constexpr int CMP_BLOCK = 256;
constexpr int CMP_ITEMS = 4;
constexpr int CMP_TILE = CMP_BLOCK * CMP_ITEMS;          // 1024 elements
constexpr int CMP_WORDS = CMP_TILE / wave_size;          // 16 ballot words per tile

/// out[i] = valid(in[i]) && op(in[i]) ? in[i] : invalid   (filter::bypass)
template <typename type_t, typename op_t>
__global__ void __launch_bounds__(CMP_BLOCK)
    bypass_kernel(const type_t* in, std::size_t n, type_t* out, op_t op) {
  for (std::size_t i = blockIdx.x * (std::size_t)CMP_BLOCK + threadIdx.x; i < n;
       i += (std::size_t)gridDim.x * CMP_BLOCK) {
    type_t v = in[i];
    bool keep = false;
    if (util::limits::is_valid(v))
      keep = op(v);
    out[i] = keep ? v : gunrock::numeric_limits<type_t>::invalid();
  }
}

/**
 * @brief Pass 1.  flag(i, value) decides; it is called exactly once for every
 * i < n, in no particular order.
 */
template <typename type_t, typename flag_t>
__global__ void __launch_bounds__(CMP_BLOCK)
    flag_kernel(const type_t* in, std::size_t n, unsigned long long* words,
                unsigned* tile_counts, flag_t flag) {
  __shared__ unsigned s_count;
  if (threadIdx.x == 0)
    s_count = 0;
  __syncthreads();
  const std::size_t tile0 = (std::size_t)blockIdx.x * CMP_TILE;
  const int lane = lane_id();
  unsigned mine = 0;
#pragma unroll
  for (int k = 0; k < CMP_ITEMS; ++k) {
    const std::size_t i = tile0 + (std::size_t)k * CMP_BLOCK + threadIdx.x;
    bool keep = false;
    if (i < n)
      keep = flag(i, in[i]);
    const unsigned long long m = __ballot(keep);
    if (lane == 0) {
      words[i / wave_size] = m;  // i is a multiple of 64 on lane 0
      mine += (unsigned)__popcll(m);
    }
  }
  if (lane == 0 && mine)
    atomicAdd(&s_count, mine);
  __syncthreads();
  if (threadIdx.x == 0)
    tile_counts[blockIdx.x] = s_count;
}

/// Pass 2.  tile_offsets = exclusive scan of tile_counts.
template <typename type_t>
__global__ void __launch_bounds__(CMP_BLOCK)
    place_kernel(const type_t* in, std::size_t n, const unsigned long long* words,
                 const unsigned* tile_offsets, type_t* out) {
  __shared__ unsigned s_prefix[CMP_WORDS];
  const std::size_t tile0 = (std::size_t)blockIdx.x * CMP_TILE;
  const std::size_t word0 = tile0 / wave_size;
  const std::size_t n_words = (n + wave_size - 1) / wave_size;
  if (threadIdx.x < CMP_WORDS) {
    // word-level exclusive prefix inside the tile, by the first 16 lanes of wave 0
    unsigned long long m = (word0 + threadIdx.x < n_words) ? words[word0 + threadIdx.x] : 0ull;
    unsigned c = (unsigned)__popcll(m);
    unsigned incl = c;
#pragma unroll
    for (int d = 1; d < CMP_WORDS; d <<= 1) {
      unsigned y = __shfl_up(incl, d, wave_size);
      if ((int)threadIdx.x >= d)
        incl += y;
    }
    s_prefix[threadIdx.x] = incl - c;
  }
  __syncthreads();
  const unsigned base = tile_offsets[blockIdx.x];
  const int wave = threadIdx.x / wave_size;
  const int lane = lane_id();
#pragma unroll
  for (int k = 0; k < CMP_ITEMS; ++k) {
    const std::size_t i = tile0 + (std::size_t)k * CMP_BLOCK + threadIdx.x;
    const int w = k * (CMP_BLOCK / wave_size) + wave;
    if (word0 + w >= n_words)
      continue;
    const unsigned long long m = words[word0 + w];
    if ((m >> lane) & 1ull)
      out[(std::size_t)base + s_prefix[w] + rank_in_mask(m)] = in[i];
  }
}

// ---------------------------------------------------------------------------
// select_range: the ids i in [0, n) with pred(i), in ONE pass, as sorted runs.
//
// Why: a wide BFS / SSSP level spends 70-130 us of its 0.5-1.0 ms on its OUTPUT path (per-wavefront
// LDS queues, degree look-ups of the emitted neighbours, ~10 K claims on the output cursor, scattered
// 4-byte writes).  A client whose labels say what the level found (BFS: depth == level) can run the
// level with output_type = none and call this instead: one coalesced pass over the labels (16 MB at
// 2^22 vertices), one cursor claim per 8192 ids, the degree sum of the selection (the next
// advance's work hint) from the row offsets it streams past anyway, and a frontier whose ids ascend
// inside every run of one chunk.  Same result SET as sequence(0, n) + filter::predicated; the order
// between chunks is the order in which workgroups claimed space.
// ---------------------------------------------------------------------------
constexpr int SEL_BLOCK = 1024;
// ids per thread and claim, chosen by the caller: 8 (8192 ids per claim) keeps more loads in flight
// for predicates with side products (SSSP's scan: 20 / 19 / 16 us against 26 / 25 / 21 with 4); 4
// gives a plain label scan twice the workgroups (R-MAT-22's 2.4 M ids are 300 claims of 8192, 1.2
// per CU: 29 / 19 us against 26 / 14)
constexpr int SEL_ITEMS_WIDE = 8;
constexpr int SEL_ITEMS_NARROW = 4;
constexpr int SEL_WAVES = SEL_BLOCK / wave_size;  // 16

/// Side products of the same pass (both optional): `each(i)` runs once for every id (lane-local side
/// effects: SSSP writes the 2-byte bound of i for the next iteration), and the ballot of `bit(i)` is
/// stored as bit i of `words` for i < bit_limit (BFS: the settled bitmap of the next level) -- the
/// label array is read once for all three.
struct select_no_each_t {
  template <typename vertex_t>
  __device__ __forceinline__ void operator()(vertex_t const&) const {}
};
struct select_no_bit_t {
  template <typename vertex_t>
  __device__ __forceinline__ bool operator()(vertex_t const&) const { return false; }
};

template <typename vertex_t, int SEL_ITEMS, typename graph_t, typename pred_t, typename each_t, typename bit_t>
__global__ void __launch_bounds__(SEL_BLOCK)
    select_range_kernel(graph_t G, std::size_t n, pred_t pred, vertex_t* __restrict__ out,
                        std::size_t capacity, unsigned long long* counters, int cursor_slot,
                        int work_slot, int overflow_slot, std::size_t n_select, each_t each, bit_t bit,
                        unsigned long long* __restrict__ words, std::size_t bit_limit) {
  __shared__ unsigned s_count[SEL_ITEMS * SEL_WAVES];   // matches per (k, wave), then their prefix
  __shared__ unsigned long long s_base;
  __shared__ unsigned long long s_work[SEL_WAVES];
  constexpr int SEL_CHUNK = SEL_BLOCK * SEL_ITEMS;
  const int tid = threadIdx.x, lane = lane_id(), wave = tid / wave_size;
  unsigned long long work = 0;
  const std::size_t n_chunks = (n + SEL_CHUNK - 1) / SEL_CHUNK;
  for (std::size_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    unsigned long long mask[SEL_ITEMS];
#pragma unroll
    for (int k = 0; k < SEL_ITEMS; ++k) {
      const std::size_t i = chunk * SEL_CHUNK + (std::size_t)k * SEL_BLOCK + tid;
      if (i < n)
        each((vertex_t)i);
      if constexpr (!std::is_same<bit_t, select_no_bit_t>::value) {
        const unsigned long long word = __ballot(i < n && bit((vertex_t)i));
        if (lane == 0 && i < bit_limit)  // i is a multiple of 64 on lane 0; bit_limit is one too
          words[i / wave_size] = word;
      }
      // ids in [n_select, n) are visited for the side products only
      const bool keep = i < n_select && pred((vertex_t)i);
      mask[k] = __ballot(keep);
      if (keep)
        work += (unsigned long long)G.get_number_of_neighbors((vertex_t)i);
      if (lane == 0)
        s_count[k * SEL_WAVES + wave] = (unsigned)__popcll(mask[k]);
    }
    __syncthreads();
    if (wave == 0) {  // exclusive prefix over the (k, wave) groups, one or two per lane
      static_assert(SEL_ITEMS * SEL_WAVES == wave_size || SEL_ITEMS * SEL_WAVES == 2 * wave_size, "group scan");
      unsigned incl;
      if constexpr (SEL_ITEMS * SEL_WAVES == 2 * wave_size) {
        const unsigned a = s_count[2 * lane], b = s_count[2 * lane + 1];
        incl = wave_inclusive_sum(a + b);
        s_count[2 * lane] = incl - a - b;
        s_count[2 * lane + 1] = incl - b;
      } else {
        const unsigned a = s_count[lane];
        incl = wave_inclusive_sum(a);
        s_count[lane] = incl - a;
      }
      if (lane == wave_size - 1)
        s_base = incl ? atomicAdd(&counters[cursor_slot], (unsigned long long)incl) : 0ull;
    }
    __syncthreads();
    const unsigned long long base = s_base;
#pragma unroll
    for (int k = 0; k < SEL_ITEMS; ++k) {
      if ((mask[k] >> lane) & 1ull) {
        const unsigned long long at = base + s_count[k * SEL_WAVES + wave] + rank_in_mask(mask[k]);
        if (at < capacity)
          out[at] = (vertex_t)(chunk * SEL_CHUNK + (std::size_t)k * SEL_BLOCK + tid);
        else
          counters[overflow_slot] = 1ull;
      }
    }
    __syncthreads();  // s_count / s_base are rewritten by the next chunk
  }
  work = wave_sum(work);
  if (lane == 0)
    s_work[wave] = work;
  __syncthreads();
  if (tid == 0) {
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < SEL_WAVES; ++w)
      t += s_work[w];
    if (t)
      atomicAdd(&counters[work_slot], t);
  }
}

/// Number of tiles for n elements.
inline std::size_t compaction_tiles(std::size_t n) { return (n + CMP_TILE - 1) / CMP_TILE; }

}  // namespace kernels
}  // namespace hip
}  // namespace gunrock
