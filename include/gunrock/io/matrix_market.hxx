/**
 * @file matrix_market.hxx
 * @brief Matrix Market coordinate loader -> host coo_t.
 *
 * Behaviour of reference io/matrix_market.hxx:99-240: pattern entries get weight
 * 1 (:146-163), real/integer weights are read as double and narrowed (:164-187),
 * indices are 1-based, symmetric storage emits (i,j) directly followed by (j,i)
 * and diagonal entries once (:194-235).  Own parser (no NIST mmio.c); dense
 * "array" files and complex/hermitian/skew kinds are rejected with an exception
 * where the reference calls exit(1).
 */
#pragma once

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <limits>
#include <string>

#include <gunrock/formats/formats.hxx>

namespace gunrock {

namespace util {
inline std::string extract_filename(std::string path) {
  auto p = path.find_last_of("/\\");
  return p == std::string::npos ? path : path.substr(p + 1);
}
inline std::string extract_dataset(std::string filename) {
  auto p = filename.find_last_of('.');
  return p == std::string::npos ? filename : filename.substr(0, p);
}
inline bool has_extension(const std::string& f, const std::string& ext) {
  return f.size() >= ext.size() && f.compare(f.size() - ext.size(), ext.size(), ext) == 0;
}
inline bool is_market(std::string f) { return has_extension(f, ".mtx") || has_extension(f, ".mmio"); }
inline bool is_binary_csr(std::string f) { return has_extension(f, ".csr"); }
}  // namespace util

namespace io {

enum matrix_market_format_t { coordinate, array };
enum matrix_market_data_t { real, complex, pattern, integer };
enum matrix_market_storage_scheme_t { general, hermitian, symmetric, skew };

template <typename vertex_t, typename edge_t, typename weight_t>
struct matrix_market_t {
  std::string filename;
  std::string dataset;
  matrix_market_format_t format{coordinate};
  matrix_market_data_t data{real};
  matrix_market_storage_scheme_t scheme{general};

  auto load(std::string _filename) {
    using coo_type = format::coo_t<memory::memory_space_t::host, vertex_t, edge_t, weight_t>;
    filename = _filename;
    dataset = util::extract_dataset(util::extract_filename(filename));
    FILE* f = std::fopen(filename.c_str(), "r");
    error::throw_if_exception(f == nullptr, "File could not be opened: " + filename);
    char line[1100], banner[64], obj[64], fmt[64], field[64], sym[64];
    bool ok = std::fgets(line, sizeof line, f) != nullptr &&
              std::sscanf(line, "%63s %63s %63s %63s %63s", banner, obj, fmt, field, sym) == 5;
    auto lower = [](char* s) { for (; *s; ++s) *s = (char)std::tolower((unsigned char)*s); };
    if (ok) { lower(obj); lower(fmt); lower(field); lower(sym); }
    if (!ok || std::string(banner) != "%%MatrixMarket" || std::string(obj) != "matrix") {
      std::fclose(f);
      error::throw_if_exception(true, "Could not process Matrix Market banner");
    }
    if (std::string(fmt) != "coordinate") {
      std::fclose(f);
      error::throw_if_exception(true, "File is not a sparse matrix");
    }
    const std::string fld(field), s(sym);
    const bool is_pattern = fld == "pattern";
    if (!(is_pattern || fld == "real" || fld == "integer")) {
      std::fclose(f);
      error::throw_if_exception(true, "Unrecognized matrix market format type");
    }
    data = is_pattern ? pattern : (fld == "real" ? real : integer);
    scheme = s == "symmetric" ? symmetric : (s == "hermitian" ? hermitian : (s == "skew-symmetric" ? skew : general));

    std::size_t M = 0, N = 0, NZ = 0;
    for (;;) {
      if (!std::fgets(line, sizeof line, f)) {
        std::fclose(f);
        error::throw_if_exception(true, "Could not read file info (M, N, NNZ)");
      }
      if (line[0] == '%')
        continue;
      if (std::sscanf(line, "%zu %zu %zu", &M, &N, &NZ) == 3)
        break;
    }
    error::throw_if_exception(M >= (std::size_t)std::numeric_limits<vertex_t>::max() ||
                                  N >= (std::size_t)std::numeric_limits<vertex_t>::max(),
                              "vertex_t overflow");
    error::throw_if_exception(NZ >= (std::size_t)std::numeric_limits<edge_t>::max(),
                              "edge_t overflow");
    std::vector<vertex_t> I, J;
    std::vector<weight_t> V;
    I.reserve(NZ); J.reserve(NZ); V.reserve(NZ);
    for (std::size_t i = 0; i < NZ; ++i) {
      std::size_t r = 0, c = 0;
      double w = 1.0;
      int got = is_pattern ? std::fscanf(f, " %zu %zu \n", &r, &c)
                           : std::fscanf(f, " %zu %zu %lf \n", &r, &c, &w);
      if (got != (is_pattern ? 2 : 3) || r == 0 || c == 0) {
        std::fclose(f);
        error::throw_if_exception(true, "Could not read edge from market file");
      }
      I.push_back((vertex_t)(r - 1)); J.push_back((vertex_t)(c - 1));
      V.push_back(is_pattern ? (weight_t)1 : (weight_t)w);
      if (scheme == symmetric && r != c) {
        I.push_back((vertex_t)(c - 1)); J.push_back((vertex_t)(r - 1));
        V.push_back(V.back());
      }
    }
    std::fclose(f);
    coo_type coo((vertex_t)M, (vertex_t)N, (edge_t)I.size());
    std::copy(I.begin(), I.end(), coo.row_indices.begin());
    std::copy(J.begin(), J.end(), coo.column_indices.begin());
    std::copy(V.begin(), V.end(), coo.nonzero_values.begin());
    return coo;
  }
};

}  // namespace io
}  // namespace gunrock
