/**
 * @file type_limits.hxx
 * @brief The "invalid element" sentinel of frontiers and its test.
 *
 * Same contract as reference include/gunrock/util/type_limits.hxx:18-70:
 * invalid() is -1 for signed integers, max() for unsigned, NaN for floating
 * point; is_valid() is its negation.  Written as one constexpr function per
 * category instead of three partial specialisations.
 */
#pragma once

#include <hip/hip_runtime.h>

#include <limits>
#include <type_traits>

namespace gunrock {

template <typename type_t>
struct numeric_limits : std::numeric_limits<type_t> {
  static_assert(std::is_arithmetic<type_t>::value, "numeric_limits: arithmetic types only");

  __host__ __device__ static constexpr type_t invalid() {
    if constexpr (std::is_floating_point<type_t>::value)
      return std::numeric_limits<type_t>::quiet_NaN();
    else if constexpr (std::is_signed<type_t>::value)
      return static_cast<type_t>(-1);
    else
      return std::numeric_limits<type_t>::max();
  }
};

namespace util {
namespace limits {

template <typename type_t>
__host__ __device__ __forceinline__ constexpr bool is_valid(type_t value) {
  static_assert(std::is_arithmetic<type_t>::value, "is_valid: arithmetic types only");
  if constexpr (std::is_floating_point<type_t>::value)
    return value == value;  // NaN is the only value unequal to itself
  else
    return value != gunrock::numeric_limits<type_t>::invalid();
}

}  // namespace limits
}  // namespace util
}  // namespace gunrock
