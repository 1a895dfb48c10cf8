/**
 * @file formats.hxx
 * @brief Owning sparse formats of the load path: coo_t, csr_t.
 *
 * Surface of reference formats/coo.hxx:21-46 and formats/csr.hxx:25-238
 * (number_of_rows/columns/nonzeros, row_offsets / column_indices /
 * nonzero_values, from_coo, read_binary, write_binary).  The on-disk ".csr"
 * layout is the reference's (csr.hxx:159-236): {rows, cols, nnz} then the three
 * arrays, raw.  from_coo is a stable counting sort on the row (csr.hxx:119-147)
 * with the arrays sized in both memory spaces (defect q5 not replicated).
 */
#pragma once

#include <cstdio>
#include <string>
#include <vector>

#include <gunrock/container/vector.hxx>
#include <gunrock/graph/graph.hxx>

namespace gunrock {
namespace format {

using memory::memory_space_t;

template <memory_space_t space, typename index_t, typename offset_t, typename value_t>
struct coo_t {
  using index_type = index_t;
  using offset_type = offset_t;
  using value_type = value_t;
  index_t number_of_rows{0};
  index_t number_of_columns{0};
  offset_t number_of_nonzeros{0};
  vector_t<index_t, space> row_indices;
  vector_t<index_t, space> column_indices;
  vector_t<value_t, space> nonzero_values;

  coo_t() = default;
  coo_t(index_t r, index_t c, offset_t nnz)
      : number_of_rows(r), number_of_columns(c), number_of_nonzeros(nnz),
        row_indices(nnz), column_indices(nnz), nonzero_values(nnz) {}
};

template <memory_space_t space, typename index_t, typename offset_t, typename value_t>
struct csr_t {
  using index_type = index_t;    // harness checkers name them (examples/algorithms/spmv/spmv_cpu.hxx:28-30)
  using offset_type = offset_t;
  using value_type = value_t;
  index_t number_of_rows{0};
  index_t number_of_columns{0};
  offset_t number_of_nonzeros{0};
  vector_t<offset_t, space> row_offsets;
  vector_t<index_t, space> column_indices;
  vector_t<value_t, space> nonzero_values;

  csr_t() = default;
  csr_t(index_t r, index_t c, offset_t nnz)
      : number_of_rows(r), number_of_columns(c), number_of_nonzeros(nnz),
        row_offsets(r + 1), column_indices(nnz), nonzero_values(nnz) {}
  /// Copy from another memory space.
  template <memory_space_t other>
  csr_t(const csr_t<other, index_t, offset_t, value_t>& rhs)
      : number_of_rows(rhs.number_of_rows), number_of_columns(rhs.number_of_columns),
        number_of_nonzeros(rhs.number_of_nonzeros), row_offsets(rhs.row_offsets),
        column_indices(rhs.column_indices), nonzero_values(rhs.nonzero_values) {}

  /// Stable row sort of a host COO; duplicates are kept.
  csr_t& from_coo(const coo_t<memory_space_t::host, index_t, offset_t, value_t>& coo) {
    number_of_rows = coo.number_of_rows;
    number_of_columns = coo.number_of_columns;
    number_of_nonzeros = coo.number_of_nonzeros;
    const std::size_t R = (std::size_t)number_of_rows, NZ = (std::size_t)number_of_nonzeros;
    std::vector<offset_t> Ap(R + 1, 0);
    std::vector<index_t> Aj(NZ);
    std::vector<value_t> Ax(NZ);
    for (std::size_t n = 0; n < NZ; ++n)
      ++Ap[(std::size_t)coo.row_indices[n] + 1];
    for (std::size_t r = 0; r < R; ++r)
      Ap[r + 1] += Ap[r];
    std::vector<offset_t> cursor(Ap.begin(), Ap.end() - 1);
    for (std::size_t n = 0; n < NZ; ++n) {
      const std::size_t dest = (std::size_t)cursor[(std::size_t)coo.row_indices[n]]++;
      Aj[dest] = coo.column_indices[n];
      Ax[dest] = coo.nonzero_values[n];
    }
    row_offsets = vector_t<offset_t, memory_space_t::host>(Ap.begin(), Ap.end());
    column_indices = vector_t<index_t, memory_space_t::host>(Aj.begin(), Aj.end());
    nonzero_values = vector_t<value_t, memory_space_t::host>(Ax.begin(), Ax.end());
    return *this;
  }

  void read_binary(std::string filename) {
    FILE* f = std::fopen(filename.c_str(), "rb");
    error::throw_if_exception(f == nullptr, "csr_t::read_binary: cannot open " + filename);
    bool ok = std::fread(&number_of_rows, sizeof(index_t), 1, f) == 1 &&
              std::fread(&number_of_columns, sizeof(index_t), 1, f) == 1 &&
              std::fread(&number_of_nonzeros, sizeof(offset_t), 1, f) == 1;
    std::vector<offset_t> Ap(ok ? (std::size_t)number_of_rows + 1 : 0);
    std::vector<index_t> Aj(ok ? (std::size_t)number_of_nonzeros : 0);
    std::vector<value_t> Ax(ok ? (std::size_t)number_of_nonzeros : 0);
    ok = ok && std::fread(Ap.data(), sizeof(offset_t), Ap.size(), f) == Ap.size() &&
         std::fread(Aj.data(), sizeof(index_t), Aj.size(), f) == Aj.size() &&
         std::fread(Ax.data(), sizeof(value_t), Ax.size(), f) == Ax.size();
    std::fclose(f);
    error::throw_if_exception(!ok, "csr_t::read_binary: short file " + filename);
    row_offsets = vector_t<offset_t, memory_space_t::host>(Ap.begin(), Ap.end());
    column_indices = vector_t<index_t, memory_space_t::host>(Aj.begin(), Aj.end());
    nonzero_values = vector_t<value_t, memory_space_t::host>(Ax.begin(), Ax.end());
  }

  void write_binary(std::string filename) {
    FILE* f = std::fopen(filename.c_str(), "wb");
    error::throw_if_exception(f == nullptr, "csr_t::write_binary: cannot open " + filename);
    vector_t<offset_t, memory_space_t::host> Ap(row_offsets);
    vector_t<index_t, memory_space_t::host> Aj(column_indices);
    vector_t<value_t, memory_space_t::host> Ax(nonzero_values);
    std::fwrite(&number_of_rows, sizeof(index_t), 1, f);
    std::fwrite(&number_of_columns, sizeof(index_t), 1, f);
    std::fwrite(&number_of_nonzeros, sizeof(offset_t), 1, f);
    std::fwrite(Ap.data(), sizeof(offset_t), Ap.size(), f);
    std::fwrite(Aj.data(), sizeof(index_t), Aj.size(), f);
    std::fwrite(Ax.data(), sizeof(value_t), Ax.size(), f);
    std::fclose(f);
  }
};

}  // namespace format

namespace graph {
namespace build {
/// Wrap an owning csr_t (reference graph/build.hxx:38-52).
template <memory_space_t space, view_t build_views, typename edge_t, typename vertex_t,
          typename weight_t>
auto from_csr(format::csr_t<space, vertex_t, edge_t, weight_t>& csr) {
  return from_csr<space, build_views>(csr.number_of_rows, csr.number_of_columns,
                                      csr.number_of_nonzeros,
                                      thrust::raw_pointer_cast(csr.row_offsets.data()),
                                      thrust::raw_pointer_cast(csr.column_indices.data()),
                                      thrust::raw_pointer_cast(csr.nonzero_values.data()));
}
}  // namespace build
}  // namespace graph
}  // namespace gunrock
