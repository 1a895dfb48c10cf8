/**
 * @file smtx.hxx
 * @brief io::smtx_t -- loader of the ".smtx" sparse-matrix text format into a host csr_t
 * (reference io/smtx.hxx:41-170).  Layout: any number of '%' comment lines, then three data
 * lines -- "rows columns nonzeros" (blank- or, with first_line_csv, comma-separated), the
 * rows+1 row offsets, the column indices.  The format carries no values.
 *
 * Difference from the reference, on purpose: the reference draws each value from an unseeded
 * random generator in [1, 10) (smtx.hxx:142-143), so two loads of one file differ.  Here the
 * value of entry k is a pure function of (seed, k) in the same range -- loads are reproducible
 * and a test can pin them.
 */
#pragma once

#include <cstdint>
#include <fstream>
#include <limits>
#include <sstream>
#include <string>

#include <gunrock/formats/formats.hxx>
#include <gunrock/io/matrix_market.hxx>

namespace gunrock {
namespace io {

using namespace memory;

template <typename vertex_t, typename edge_t, typename weight_t>
struct smtx_t {
  std::string filename;
  std::string dataset;
  std::uint64_t value_seed = 0x5eed;

  /// @return host CSR; throws error::exception_t on a missing or inconsistent file.
  format::csr_t<memory_space_t::host, vertex_t, edge_t, weight_t> load(std::string _filename,
                                                                        bool first_line_csv = false) {
    filename = _filename;
    dataset = util::extract_dataset(util::extract_filename(filename));
    std::ifstream file(filename);
    error::throw_if_exception(!file.is_open(), "smtx: unable to open " + filename);

    auto next_data_line = [&file, this]() {
      std::string line;
      do {
        error::throw_if_exception(!std::getline(file, line), "smtx: " + filename + " ends early");
      } while (!line.empty() && line[0] == '%');
      return line;
    };

    std::string head = next_data_line();
    if (first_line_csv)
      for (auto& c : head)
        if (c == ',')
          c = ' ';
    std::istringstream dims(head);
    unsigned long long rows = 0, columns = 0, nonzeros = 0;
    dims >> rows >> columns >> nonzeros;
    error::throw_if_exception(dims.fail(), "smtx: bad dimension line in " + filename);
    error::throw_if_exception(rows >= (unsigned long long)std::numeric_limits<vertex_t>::max() ||
                                  columns >= (unsigned long long)std::numeric_limits<vertex_t>::max(),
                              "vertex_t overflow");
    error::throw_if_exception(nonzeros >= (unsigned long long)std::numeric_limits<edge_t>::max(),
                              "edge_t overflow");

    format::csr_t<memory_space_t::host, vertex_t, edge_t, weight_t> csr(
        (vertex_t)rows, (vertex_t)columns, (edge_t)nonzeros);
    std::istringstream offsets(next_data_line());
    std::size_t n_offsets = 0;
    for (unsigned long long o; offsets >> o; ++n_offsets)
      if (n_offsets <= rows)
        csr.row_offsets[n_offsets] = (edge_t)o;
    error::throw_if_exception(n_offsets != rows + 1,
                              "smtx: " + filename + " has " + std::to_string(n_offsets) +
                                  " row offsets, its first line promises " +
                                  std::to_string(rows + 1));
    std::istringstream indices(next_data_line());
    std::size_t n_indices = 0;
    for (long long c; indices >> c; ++n_indices)
      if (n_indices < nonzeros) {
        csr.column_indices[n_indices] = (vertex_t)c;
        csr.nonzero_values[n_indices] = value_of(n_indices);
      }
    error::throw_if_exception(n_indices != nonzeros,
                              "smtx: " + filename + " has " + std::to_string(n_indices) +
                                  " column indices, its first line promises " +
                                  std::to_string(nonzeros));
    return csr;
  }

  /// Value of entry k: uniform in [1, 10), a function of (value_seed, k) only.
  weight_t value_of(std::size_t k) const {
    std::uint64_t z = value_seed + 0x9e3779b97f4a7c15ull * (std::uint64_t)(k + 1);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return (weight_t)(1.0 + 9.0 * (double)(z >> 11) * (1.0 / 9007199254740992.0));
  }
};

}  // namespace io
}  // namespace gunrock
