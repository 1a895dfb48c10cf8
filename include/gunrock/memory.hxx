/**
 * @file memory.hxx
 * @brief Reference include path gunrock/memory.hxx (memory.hxx:33-158): memory_space_t,
 * allocate / free and the raw-pointer cast live in hip/runtime.hxx (plain hipMalloc /
 * hipHostMalloc); this header adds the pieces that need rocThrust -- the device_ptr overload of
 * raw_pointer_cast (so `.data()` of a device OR host vector casts, memory.hxx:103-122) and the
 * shared_ptr helpers deleter_t / allocator_t (memory.hxx:130-156).
 */
#pragma once

#include <thrust/device_ptr.h>

#include <gunrock/hip/runtime.hxx>

namespace gunrock {
namespace memory {

/// In-place form (reference memory.hxx:44-58): `pointer` receives the allocation.
template <typename type_t>
inline void allocate(type_t*& pointer, std::size_t bytes, memory_space_t space) {
  pointer = allocate<type_t>(bytes, space);
}

template <typename type_t>
inline type_t* raw_pointer_cast(thrust::device_ptr<type_t> pointer) {
  return thrust::raw_pointer_cast(pointer);
}

/// std::shared_ptr<T>(allocate<T>(bytes), deleter_t<T>()) frees device memory when released.
template <typename type_t>
struct deleter_t {
  void operator()(type_t* pointer) const { memory::free(pointer); }
};

template <typename type_t>
struct allocator_t {
  type_t* operator()(std::size_t bytes) const { return allocate<type_t>(bytes); }
};

}  // namespace memory
}  // namespace gunrock
