/**
 * @file sample.hxx
 * @brief io::sample::csr() -- the 4 x 4 sample matrix of the reference (io/sample.hxx:58-93),
 * used by its unit tests and as a fixture here (tests/golden: "sample4x4").
 *
 *   r/c  0 1 2 3          (i, j) [w]
 *   0  [ 0 0 0 0 ]        (1, 0) [5]
 *   1  [ 5 8 0 0 ]        (1, 1) [8]
 *   2  [ 0 0 3 0 ]        (2, 2) [3]
 *   3  [ 0 6 0 0 ]        (3, 1) [6]
 *
 *   ROW_OFFSETS = [0 0 2 3 4]   COLUMN_INDEX = [0 1 2 1]   VALUES = [5 8 3 6]
 */
#pragma once

#include <gunrock/formats/formats.hxx>
#include <gunrock/graph/graph.hxx>

namespace gunrock {
namespace io {
namespace sample {

using namespace memory;

template <memory_space_t space = memory_space_t::device,
          typename vertex_t = int,
          typename edge_t = int,
          typename weight_t = float>
format::csr_t<space, vertex_t, edge_t, weight_t> csr() {
  format::csr_t<memory_space_t::host, vertex_t, edge_t, weight_t> m(4, 4, 4);
  const edge_t offsets[5] = {0, 0, 2, 3, 4};
  const vertex_t columns[4] = {0, 1, 2, 1};
  const weight_t values[4] = {5, 8, 3, 6};
  for (int i = 0; i < 5; ++i)
    m.row_offsets[i] = offsets[i];
  for (int i = 0; i < 4; ++i) {
    m.column_indices[i] = columns[i];
    m.nonzero_values[i] = values[i];
  }
  // same space: a copy; device: one host-to-device copy per array (csr_t's converting constructor)
  return format::csr_t<space, vertex_t, edge_t, weight_t>(m);
}

}  // namespace sample
}  // namespace io
}  // namespace gunrock
