/**
 * @file algorithms.hxx
 * @brief The single include of an algorithm header (reference
 * algorithms/algorithms.hxx:17-41): runtime, framework, operators, graph,
 * formats, loaders, utilities -- everything bfs.hxx / sssp.hxx / pr.hxx name.
 * The thrust symbols those clients call directly (device_vector, fill, fill_n,
 * copy_n, transform, transform_reduce, ...) come from ROCm's rocThrust.
 */
#pragma once

#include <limits>
#include <memory>

#include <thrust/copy.h>
#include <thrust/device_ptr.h>
#include <thrust/device_vector.h>
#include <thrust/execution_policy.h>
#include <thrust/fill.h>
#include <thrust/functional.h>
#include <thrust/host_vector.h>
#include <thrust/transform.h>
#include <thrust/transform_reduce.h>

// core (reference algorithms.hxx:17-18)
#include <gunrock/memory.hxx>
#include <gunrock/error.hxx>
#include <gunrock/hip/context.hxx>
#include <gunrock/hip/launch_box.hxx>

// framework (:21)
#include <gunrock/framework/framework.hxx>

// utilities (:24-26)
#include <gunrock/util/math.hxx>
#include <gunrock/util/type_limits.hxx>
#include <gunrock/util/print.hxx>
#include <gunrock/util/compare.hxx>

// formats, loaders, graph, containers (:29-41)
#include <gunrock/formats/formats.hxx>
#include <gunrock/io/matrix_market.hxx>
#include <gunrock/io/smtx.hxx>
#include <gunrock/io/sample.hxx>
#include <gunrock/graph/graph.hxx>
#include <gunrock/graph/transpose.hxx>
#include <gunrock/container/array.hxx>
#include <gunrock/container/vector.hxx>

namespace gunrock {
using memory::memory_space_t;
}  // namespace gunrock
