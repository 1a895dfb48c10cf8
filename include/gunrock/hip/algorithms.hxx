/**
 * @file algorithms.hxx
 * @brief Small device-wide helpers for algorithm code written against this
 * engine (fill, transform-reduce) -- hand-written kernels / rocPRIM, for clients
 * that do not want thrust.  (The reference's clients call thrust directly:
 * algorithms/pr.hxx:120-133,172-175.)
 */
#pragma once

#include <gunrock/hip/context.hxx>
#include <gunrock/hip/primitives.hxx>

namespace gunrock {
namespace hip {

namespace detail {
template <typename T>
__global__ void __launch_bounds__(256) fill_kernel(T* p, std::size_t n, T value) {
  for (std::size_t i = blockIdx.x * (std::size_t)blockDim.x + threadIdx.x; i < n;
       i += (std::size_t)gridDim.x * blockDim.x)
    p[i] = value;
}
template <typename op_t>
__global__ void __launch_bounds__(256) index_kernel(std::size_t n, op_t op) {
  for (std::size_t i = blockIdx.x * (std::size_t)blockDim.x + threadIdx.x; i < n;
       i += (std::size_t)gridDim.x * blockDim.x)
    op(i);
}
inline unsigned grid(std::size_t n) {
  std::size_t g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
}  // namespace detail

template <typename T>
void fill(T* p, std::size_t n, T value, hipStream_t stream) {
  if (!n)
    return;
  detail::fill_kernel<<<detail::grid(n), 256, 0, stream>>>(p, n, value);
  GRX_HIP_CHECK(hipGetLastError());
}

/// op(i) for i in [0, n)
template <typename op_t>
void for_each_index(std::size_t n, op_t op, hipStream_t stream) {
  if (!n)
    return;
  detail::index_kernel<<<detail::grid(n), 256, 0, stream>>>(n, op);
  GRX_HIP_CHECK(hipGetLastError());
}

/**
 * @brief result = reduce(f(0), f(1), ..., f(n-1)) with `combine`, deterministic
 * for a given n (rocPRIM tree).  Synchronises the stream and returns the value.
 */
template <typename T, typename f_t, typename combine_t>
T transform_reduce(std::size_t n, f_t f, T init, combine_t combine,
                   gcuda::standard_context_t& ctx) {
  auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<std::size_t>(0), f);
  auto& ws = ctx.workspace();
  T* d_out = reinterpret_cast<T*>(ws.counters() + 28);
  std::size_t bytes = 0;
  GRX_HIP_CHECK(rocprim::reduce(nullptr, bytes, in, d_out, init, n, combine, ctx.stream()));
  void* temp = ws.scratch(bytes);
  GRX_HIP_CHECK(rocprim::reduce(temp, bytes, in, d_out, init, n, combine, ctx.stream()));
  T* landing = reinterpret_cast<T*>(ws.mirror() + 28);
  GRX_HIP_CHECK(hipMemcpyAsync(landing, d_out, sizeof(T), hipMemcpyDeviceToHost, ctx.stream()));
  ctx.synchronize();
  return *landing;
}

}  // namespace hip
}  // namespace gunrock
