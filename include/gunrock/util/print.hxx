/**
 * @file print.hxx
 * @brief print::head -- the first k elements of a vector or of a device / host array, on
 * std::cout (reference util/print.hxx:30-66; the harnesses print their results with it,
 * examples/algorithms/bfs/bfs.cu:88-89).  Output format: `name[:k] = a b c \n`.
 */
#pragma once

#include <algorithm>
#include <iostream>
#include <string>
#include <vector>

#include <gunrock/hip/runtime.hxx>

namespace gunrock {
namespace print {

/// Any container with size() and operator[] (thrust::device_vector, thrust::host_vector,
/// std::vector).  Device vectors are read back in ONE copy, not element by element.
template <typename vector_t>
void head(vector_t& x, int k, std::string name = "") {
  using type_t = typename vector_t::value_type;
  const std::size_t m = std::min<std::size_t>(k < 0 ? 0 : (std::size_t)k, x.size());
  std::vector<type_t> h(m);
  for (std::size_t i = 0; i < m; ++i)
    h[i] = x[i];
  if (!name.empty())
    std::cout << name << "[:" << m << "] = ";
  for (std::size_t i = 0; i < m; ++i)
    std::cout << h[i] << " ";
  std::cout << std::endl;
}

/// Raw pointer to n elements in device OR host memory (the pointer's space is looked up).
template <typename type_t>
void head(type_t* x, int k, int n, std::string name = "") {
  const std::size_t m = (std::size_t)std::max(0, std::min(k, n));
  std::vector<type_t> h(m);
  if (m) {
    hipPointerAttribute_t attr{};
    const bool on_device =
        hipPointerGetAttributes(&attr, x) == hipSuccess && attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();  // an unregistered host pointer reports an error: it is not one
    if (on_device)
      GRX_HIP_CHECK(hipMemcpy(h.data(), x, m * sizeof(type_t), hipMemcpyDeviceToHost));
    else
      std::copy(x, x + m, h.begin());
  }
  if (!name.empty())
    std::cout << name << "[:" << m << "] = ";
  for (std::size_t i = 0; i < m; ++i)
    std::cout << h[i] << " ";
  std::cout << std::endl;
}

}  // namespace print
}  // namespace gunrock
