/**
 * @file exchange_kernels.hxx
 * @brief Device side of the frontier exchange between the ranks of a vertex-partitioned
 * traversal: pack a rank's finds as (vertex, label) pairs, and min-combine the
 * gathered pairs into the label replica while appending the OWNED, improved vertices to the next
 * frontier -- exactly once per superstep.  No reference counterpart (its operators throw for more
 * than one context, framework/operators/advance/advance.hxx:125-128).
 *
 * Appends go through a workgroup scan + ONE cursor atomic per 2048 items (append_tile): a single
 * device word retires ~90 atomics/us on MI355X, one per wavefront made these kernels cursor-bound.
 * Labels are 32-bit (int32 depth, float distance); a pair is one int64 word: low half vertex id,
 * high half the label's bit pattern.
 */
#pragma once

#include <cstdint>

#include <gunrock/hip/kernels/advance_kernels.hxx>
#include <gunrock/hip/primitives.hxx>
#include <gunrock/util/math.hxx>

namespace gunrock {
namespace hip {
namespace kernels {

constexpr int APPEND_ITEMS = 8;                   // items per thread and round
constexpr int APPEND_TILE = 256 * APPEND_ITEMS;   // items per workgroup and round

/// LDS of one workgroup-wide append: the kept items of a tile are ranked with a workgroup scan,
/// staged, and written with ONE cursor atomic per tile (a single-address atomic retires at
/// ~90/us on this part -- one per wavefront made these kernels cursor-bound).
template <typename T>
struct tile_append_t {
  T staged[APPEND_TILE];
  unsigned wave_totals[256 / wave_size + 1];
  unsigned long long base;
};

/// Every thread of the workgroup calls this with its (up to APPEND_ITEMS) kept values.
template <typename T>
__device__ __forceinline__ void append_tile(tile_append_t<T>& s, const T (&val)[APPEND_ITEMS],
                                            unsigned keep, T* out, unsigned long long capacity,
                                            unsigned long long* cursor, unsigned long long* overflow) {
  unsigned total = 0;
  unsigned at = block_exclusive_sum<256>((unsigned)__popc(keep), total, s.wave_totals);
  if (total == 0)  // workgroup-uniform
    return;
#pragma unroll
  for (int k = 0; k < APPEND_ITEMS; ++k)
    if (keep & (1u << k))
      s.staged[at++] = val[k];
  if (threadIdx.x == 0)
    s.base = atomicAdd(cursor, (unsigned long long)total);
  __syncthreads();
  const unsigned long long base = s.base;
  for (unsigned i = threadIdx.x; i < total; i += 256) {
    if (base + i < capacity)
      out[base + i] = s.staged[i];
    else
      *overflow = 1ull;
  }
  __syncthreads();  // staged[] is reused by the next tile
}

/// Pack the finds of one superstep (duplicate-free: a BFS level discovers a vertex once per rank,
/// the SSSP relax lambda keeps the first improver per superstep): send[1 + k] = (vertex | label
/// bits << 32), the label read AFTER the advance, i.e. the best this rank knows.
/// counters[C_SELECT] counts them.
template <typename label_t>
__global__ void __launch_bounds__(256)
    pack_pairs_kernel(const int32_t* found, int64_t count, const label_t* labels, int64_t* send,
                      int64_t send_capacity,
                      unsigned long long* counters, const unsigned long long* count_device = nullptr) {
  __shared__ tile_append_t<int64_t> lds;
  if (count_device)
    count = (int64_t)__hip_atomic_load(count_device, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int64_t stride = (int64_t)gridDim.x * APPEND_TILE;
  const int64_t rounds = (count + stride - 1) / stride;
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t tile = r * stride + (int64_t)blockIdx.x * APPEND_TILE;
    int64_t pair[APPEND_ITEMS];
    unsigned keep = 0;
#pragma unroll
    for (int k = 0; k < APPEND_ITEMS; ++k) {
      const int64_t i = tile + k * 256 + threadIdx.x;
      pair[k] = 0;
      if (i < count) {
        const int32_t v = found[i];
        const label_t l = labels[v];
        uint32_t bits;
        __builtin_memcpy(&bits, &l, 4);
        pair[k] = (int64_t)(((uint64_t)bits << 32) | (uint32_t)v);
        keep |= 1u << k;
      }
    }
    append_tile(lds, pair, keep, send + 1, (unsigned long long)(send_capacity - 1),
                counters + C_SELECT, counters + C_OVERFLOW);
  }
}


/// pack_pairs_kernel for an arbitrary client's output frontier: invalid entries (holes left by a
/// bypass filter, reference filter/bypass.hxx:29-34) are skipped and a vertex that occurs several
/// times is packed once (`stamp[v] <- tag` by exchange; the frontier of an unchanged sssp.hxx has
/// duplicates), so the send slot never needs more than V + 1 words.
template <typename label_t>
__global__ void __launch_bounds__(256)
    pack_unique_pairs_kernel(const int32_t* found, int64_t count, const label_t* labels,
                             int32_t* stamp, int32_t tag, int64_t* send, int64_t send_capacity,
                             unsigned long long* counters) {
  __shared__ tile_append_t<int64_t> lds;
  const int64_t stride = (int64_t)gridDim.x * APPEND_TILE;
  const int64_t rounds = (count + stride - 1) / stride;
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t tile = r * stride + (int64_t)blockIdx.x * APPEND_TILE;
    int64_t pair[APPEND_ITEMS];
    unsigned keep = 0;
#pragma unroll
    for (int k = 0; k < APPEND_ITEMS; ++k) {
      const int64_t i = tile + k * 256 + threadIdx.x;
      pair[k] = 0;
      if (i < count) {
        const int32_t v = found[i];
        if (v >= 0 && atomicExch(&stamp[v], tag) != tag) {
          const label_t l = labels[v];
          uint32_t bits;
          __builtin_memcpy(&bits, &l, 4);
          pair[k] = (int64_t)(((uint64_t)bits << 32) | (uint32_t)v);
          keep |= 1u << k;
        }
      }
    }
    append_tile(lds, pair, keep, send + 1, (unsigned long long)(send_capacity - 1),
                counters + C_SELECT, counters + C_OVERFLOW);
  }
}

template <typename label_t, bool DEDUPE>
__global__ void __launch_bounds__(256)
    admit_kernel(label_t* labels, int32_t* stamp, int32_t round, const int64_t* recv, int32_t world,
                 int64_t slot, int32_t me, int32_t lo, int32_t hi, int32_t* next,
                 unsigned long long next_capacity, unsigned long long* next_count,
                 unsigned long long* overflow) {
  // tiles over (rank, entry); ranks' slots are padded to `slot` words.  The trip count is
  // workgroup-uniform (append_tile synchronises).
  __shared__ tile_append_t<int32_t> lds;
  const int64_t per_rank = slot - 1;
  const int64_t total = (int64_t)world * per_rank;
  const int64_t stride = (int64_t)gridDim.x * APPEND_TILE;
  const int64_t rounds = (total + stride - 1) / stride;
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t tile = r * stride + (int64_t)blockIdx.x * APPEND_TILE;
    int32_t admitted[APPEND_ITEMS];
    unsigned keep = 0;
#pragma unroll
    for (int k = 0; k < APPEND_ITEMS; ++k) {
      const int64_t t = tile + k * 256 + threadIdx.x;
      admitted[k] = -1;
      if (t < total) {
        const int32_t p = (int32_t)(t / per_rank);
        const int64_t i = t - (int64_t)p * per_rank;
        const int64_t* seg = recv + (int64_t)p * slot;
        const int64_t cnt = seg[0] < per_rank ? seg[0] : per_rank;
        if (i < cnt) {
          const uint64_t word = (uint64_t)seg[1 + i];
          const int32_t v = (int32_t)(uint32_t)word;
          const uint32_t bits = (uint32_t)(word >> 32);
          label_t l;
          __builtin_memcpy(&l, &bits, 4);
          // this rank's own advance already improved its own finds
          const bool fresh = (p == me) ? true : (l < math::atomic::min(&labels[v], l));
          // exactly one copy per superstep (the reference's SSSP bypass predicate,
          // sssp.hxx:126-136, tolerates duplicates; here the frontier stays duplicate-free so
          // that its work is bounded by the rank's edge count)
          // BFS: every rank proposes the same label, so exactly one proposal is fresh (this
          // rank's own, or the first one to win the atomic::min) -- no stamp needed
          if (fresh && v >= lo && v < hi && (!DEDUPE || atomicExch(&stamp[v], round) != round)) {
            admitted[k] = v;
            keep |= 1u << k;
          }
        }
      }
    }
    append_tile(lds, admitted, keep, next, next_capacity, next_count, overflow);
  }
}

}  // namespace kernels
}  // namespace hip
}  // namespace gunrock
