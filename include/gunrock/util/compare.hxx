/**
 * @file compare.hxx
 * @brief util::compare(device, host, n[, error_op, verbose]) -> number of
 * mismatching elements (reference util/compare.hxx:37-56).
 */
#pragma once

#include <cstddef>
#include <iostream>
#include <string>
#include <vector>

#include <gunrock/hip/runtime.hxx>

namespace gunrock {
namespace util {

namespace detail {
struct not_equal_t {
  template <typename A, typename B>
  bool operator()(A& a, B& b) const { return a != b; }
};
}  // namespace detail

template <typename type_t, typename comp_t = detail::not_equal_t>
std::size_t compare(const type_t* d_ptr, const type_t* h_ptr, const std::size_t n,
                    comp_t error_op = comp_t(), const bool verbose = false) {
  std::vector<type_t> d(n);
  if (n)
    GRX_HIP_CHECK(hipMemcpy(d.data(), d_ptr, n * sizeof(type_t), hipMemcpyDeviceToHost));
  std::size_t errors = 0;
  for (std::size_t i = 0; i < n; ++i) {
    type_t a = d[i], b = h_ptr[i];
    if (error_op(a, b)) {
      if (verbose)
        std::cout << "Error: " << a << " != " << b << std::endl;
      ++errors;
    }
  }
  return errors;
}

}  // namespace util

}  // namespace gunrock
