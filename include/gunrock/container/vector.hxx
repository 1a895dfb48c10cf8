/**
 * @file vector.hxx
 * @brief vector_t<T, space>: the owning container of the LOAD path (formats, io,
 * harnesses).  As in reference container/vector.hxx:26-31 it is a thrust vector
 * (rocThrust here) so that harness code spelling `csr.row_offsets.data().get()`
 * (examples/algorithms/bfs/bfs.cu:52-54) keeps compiling.  Nothing on the
 * operator path uses it.
 */
#pragma once

#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

#include <gunrock/hip/runtime.hxx>

namespace gunrock {

template <typename type_t, memory::memory_space_t space>
using vector_t = std::conditional_t<space == memory::memory_space_t::host,
                                    thrust::host_vector<type_t>,
                                    thrust::device_vector<type_t>>;

}  // namespace gunrock
