/**
 * @file primitives.hxx
 * @brief Wavefront / workgroup building blocks for gfx950 (wave64) and the
 * device-wide rocPRIM wrappers (scan, reduce, radix sort) with caller-provided
 * temporary storage.
 *
 * These replace what the reference takes from CUB/thrust/ModernGPU
 * (cub::BlockScan in advance/block_mapped.hxx:52,85-86; thrust::transform_reduce
 * / transform_exclusive_scan in advance/helpers.hxx:67-76,135-143; thrust::sort in
 * algorithms/sort/radix_sort.hxx:40-51).  Wave-width constants are hard-coded to
 * 64: this engine targets gfx950 only.
 */
#pragma once

#include <cstring>  // must precede rocprim with this toolchain

#include <rocprim/rocprim.hpp>

#include <gunrock/hip/runtime.hxx>

namespace gunrock {
namespace hip {

constexpr int wave_size = 64;

// ---------------------------------------------------------------------------
// wavefront level
// ---------------------------------------------------------------------------

__device__ __forceinline__ int lane_id() {
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

/// Number of set bits of `mask` strictly below this lane (v_mbcnt pair).
__device__ __forceinline__ int rank_in_mask(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                        __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

template <typename T>
__device__ __forceinline__ T wave_inclusive_sum(T x) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < wave_size; d <<= 1) {
    T y = __shfl_up(x, d, wave_size);
    if (lane >= d)
      x += y;
  }
  return x;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
  for (int d = wave_size / 2; d > 0; d >>= 1)
    x += __shfl_xor(x, d, wave_size);
  return x;
}

template <typename T>
__device__ __forceinline__ T wave_max(T x) {
#pragma unroll
  for (int d = wave_size / 2; d > 0; d >>= 1) {
    T y = __shfl_xor(x, d, wave_size);
    x = y > x ? y : x;
  }
  return x;
}

// ---------------------------------------------------------------------------
// workgroup level (BLOCK threads = BLOCK/64 wavefronts)
// ---------------------------------------------------------------------------

/**
 * @brief Exclusive prefix sum over one value per thread.  `wave_totals` is LDS
 * scratch of BLOCK/64 + 1 elements.  Contains two barriers; every thread of the
 * workgroup must call it.
 */
template <int BLOCK, typename T>
__device__ __forceinline__ T block_exclusive_sum(T x, T& block_total, T* wave_totals) {
  constexpr int WAVES = BLOCK / wave_size;
  const int lane = lane_id();
  const int wave = threadIdx.x / wave_size;
  T incl = wave_inclusive_sum(x);
  if (lane == wave_size - 1)
    wave_totals[wave] = incl;
  __syncthreads();
  T offset = 0;
  T total = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    T t = wave_totals[w];
    if (w < wave)
      offset += t;
    total += t;
  }
  block_total = total;
  __syncthreads();
  return offset + incl - x;
}

/// Largest index i in [0, n) with keys[i] <= key (keys non-decreasing, keys[0] <= key).
template <typename key_t, typename K>
__device__ __forceinline__ int rightmost_le(const K* keys, key_t key, int n) {
  int lo = 0, hi = n;  // answer in [lo, hi)
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (keys[mid] <= key)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// ---------------------------------------------------------------------------
// device-wide (rocPRIM), temporary storage supplied by the caller
// ---------------------------------------------------------------------------

namespace detail {
template <typename T>
__global__ void __launch_bounds__(256) iota_kernel(T* p, std::size_t n) {
  for (std::size_t i = blockIdx.x * (std::size_t)256 + threadIdx.x; i < n;
       i += (std::size_t)gridDim.x * 256)
    p[i] = (T)i;
}
}  // namespace detail

/// p[i] = i
template <typename T>
void for_each_index_on(std::size_t n, T* p, hipStream_t stream) {
  if (!n)
    return;
  std::size_t g = (n + 255) / 256;
  detail::iota_kernel<<<(unsigned)(g > 4096 ? 4096 : g), 256, 0, stream>>>(p, n);
  GRX_HIP_CHECK(hipGetLastError());
}

enum class sort_order_t { ascending, descending };

template <typename key_t>
std::size_t radix_sort_temp_bytes(std::size_t n) {
  std::size_t bytes = 0;
  GRX_HIP_CHECK(rocprim::radix_sort_keys(nullptr, bytes, (key_t*)nullptr, (key_t*)nullptr, n, 0,
                                         8 * sizeof(key_t), nullptr));
  return bytes;
}

/// keys_in -> keys_out (distinct arrays), stable LSD radix sort.
template <typename key_t>
void radix_sort_keys(void* temp, std::size_t temp_bytes, const key_t* keys_in, key_t* keys_out,
                     std::size_t n, sort_order_t order, hipStream_t stream) {
  // rocPRIM reads a null temp as "tell me the size" and sorts nothing: fail loudly instead
  error::throw_if_exception(temp == nullptr, "radix_sort_keys: null temporary storage");
  if (order == sort_order_t::ascending)
    GRX_HIP_CHECK(rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, n, 0,
                                           8 * sizeof(key_t), stream));
  else
    GRX_HIP_CHECK(rocprim::radix_sort_keys_desc(temp, temp_bytes, keys_in, keys_out, n, 0,
                                                8 * sizeof(key_t), stream));
}

template <typename in_it, typename out_it, typename T>
std::size_t exclusive_sum_temp_bytes(in_it in, out_it out, T init, std::size_t n) {
  std::size_t bytes = 0;
  GRX_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, init, n, rocprim::plus<T>(), nullptr));
  return bytes;
}

template <typename in_it, typename out_it, typename T>
void exclusive_sum(void* temp, std::size_t temp_bytes, in_it in, out_it out, T init, std::size_t n,
                   hipStream_t stream) {
  error::throw_if_exception(temp == nullptr, "exclusive_sum: null temporary storage");
  GRX_HIP_CHECK(rocprim::exclusive_scan(temp, temp_bytes, in, out, init, n, rocprim::plus<T>(), stream));
}

}  // namespace hip
}  // namespace gunrock
