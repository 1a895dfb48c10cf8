/**
 * @file array.hxx
 * @brief gunrock::array<T, N> -- a fixed-size aggregate usable in host AND device code
 * (reference container/array.hxx:46-260 models std::array with __host__ __device__ members).
 * An aggregate of a plain C array (N == 0: an empty struct), so it can be brace-initialised,
 * captured by value in a device lambda and passed as a kernel argument.
 */
#pragma once

#include <cstddef>
#include <iterator>
#include <type_traits>

#include <hip/hip_runtime.h>

namespace gunrock {

namespace detail {
template <typename T, std::size_t N>
struct array_storage_t {
  T items[N];
  __host__ __device__ constexpr T* ptr() noexcept { return items; }
  __host__ __device__ constexpr const T* ptr() const noexcept { return items; }
};
template <typename T>
struct array_storage_t<T, 0> {
  __host__ __device__ constexpr T* ptr() noexcept { return nullptr; }
  __host__ __device__ constexpr const T* ptr() const noexcept { return nullptr; }
};
}  // namespace detail

template <typename T, std::size_t NumElements>
struct array {
  using value_type = T;
  using pointer_t = T*;
  using const_pointer_t = const T*;
  using reference_t = T&;
  using const_reference_t = const T&;
  using iterator = T*;
  using const_iterator = const T*;
  using size_type = std::size_t;
  using difference_type = std::ptrdiff_t;
  using reverse_iterator = std::reverse_iterator<iterator>;
  using const_reverse_iterator = std::reverse_iterator<const_iterator>;

  detail::array_storage_t<T, NumElements> elements;  // public: keeps array<> an aggregate

  __host__ __device__ constexpr pointer_t data() noexcept { return elements.ptr(); }
  __host__ __device__ constexpr const_pointer_t data() const noexcept { return elements.ptr(); }
  __host__ __device__ constexpr size_type size() const noexcept { return NumElements; }
  __host__ __device__ constexpr size_type max_size() const noexcept { return NumElements; }
  __host__ __device__ constexpr bool empty() const noexcept { return NumElements == 0; }

  __host__ __device__ constexpr reference_t operator[](size_type n) noexcept { return data()[n]; }
  __host__ __device__ constexpr const_reference_t operator[](size_type n) const noexcept {
    return data()[n];
  }
  __host__ __device__ constexpr reference_t front() noexcept { return data()[0]; }
  __host__ __device__ constexpr const_reference_t front() const noexcept { return data()[0]; }
  __host__ __device__ constexpr reference_t back() noexcept {
    return data()[NumElements ? NumElements - 1 : 0];
  }
  __host__ __device__ constexpr const_reference_t back() const noexcept {
    return data()[NumElements ? NumElements - 1 : 0];
  }

  __host__ __device__ constexpr iterator begin() noexcept { return data(); }
  __host__ __device__ constexpr const_iterator begin() const noexcept { return data(); }
  __host__ __device__ constexpr iterator end() noexcept { return data() + NumElements; }
  __host__ __device__ constexpr const_iterator end() const noexcept { return data() + NumElements; }
  __host__ __device__ constexpr const_iterator cbegin() const noexcept { return data(); }
  __host__ __device__ constexpr const_iterator cend() const noexcept { return data() + NumElements; }

  __host__ __device__ void fill(const value_type& u) {
    for (size_type i = 0; i < NumElements; ++i)
      data()[i] = u;
  }
  __host__ __device__ void swap(array& other) {
    for (size_type i = 0; i < NumElements; ++i) {
      T t = data()[i];
      data()[i] = other.data()[i];
      other.data()[i] = t;
    }
  }
};

template <typename T, std::size_t N>
__host__ __device__ constexpr bool operator==(const array<T, N>& a, const array<T, N>& b) {
  for (std::size_t i = 0; i < N; ++i)
    if (!(a[i] == b[i]))
      return false;
  return true;
}
template <typename T, std::size_t N>
__host__ __device__ constexpr bool operator!=(const array<T, N>& a, const array<T, N>& b) {
  return !(a == b);
}

}  // namespace gunrock
