/**
 * @file rmat.hip
 * @brief GPU R-MAT generator + CSR build (no reference counterpart: the
 * reference has no graph generator).  Bit-exact with the oracle's
 * orc_rmat_pair / orc_rmat_csr (oracle/grx_oracle.c) -- same counter-based hash,
 * same integer quadrant thresholds, the Matrix-Market loader's symmetrisation
 * order ((u,v) then (v,u), a self loop once, duplicates kept: reference
 * io/matrix_market.hxx:213-230) and a STABLE row sort (formats/csr.hxx:119-147).
 *
 * Pipeline on the device: emit (row key, emission index) for 2 * pairs slots
 * (the unused second slot of a self loop / of a directed pair gets the sentinel
 * key 2^scale) -> rocPRIM stable radix sort by key over scale+1 bits -> regenerate
 * column and weight from the emission index -> row offsets by boundary detection.
 */
#include "capi_internal.hxx"

#include <algorithm>

using namespace essentials_amd;

namespace {

constexpr unsigned RMAT_TA = 2448131358u;    // floor(0.57 * 2^32)
constexpr unsigned RMAT_TAB = 3264175144u;   // floor(0.76 * 2^32)
constexpr unsigned RMAT_TABC = 4080218931u;  // floor(0.95 * 2^32)

__host__ __device__ inline unsigned long long mix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__host__ __device__ inline void rmat_pair(unsigned scale, unsigned long long seed,
                                          unsigned long long k, unsigned& u, unsigned& v) {
  const unsigned long long base = mix64(seed ^ mix64(k));
  u = 0;
  v = 0;
  for (unsigned l = 0; l < scale; ++l) {
    const unsigned r = (unsigned)(mix64(base + l) >> 32);
    const unsigned ub = r >= RMAT_TAB;
    const unsigned vb = (r >= RMAT_TA && r < RMAT_TAB) || (r >= RMAT_TABC);
    u = (u << 1) | ub;
    v = (v << 1) | vb;
  }
}

__host__ __device__ inline float rmat_weight(unsigned long long wseed, unsigned long long k) {
  if (wseed == 0)
    return 1.0f;
  return (float)(1 + (mix64(wseed ^ mix64(k ^ 0x5bd1e995ull)) & 63ull));
}

/// slot 2k = (u -> v); slot 2k+1 = (v -> u) when symmetrised and u != v, else sentinel.
__global__ void __launch_bounds__(256)
    emit_kernel(unsigned scale, unsigned long long seed, unsigned long long pairs, int symmetrize,
                unsigned* keys, unsigned* slots) {
  const unsigned sentinel = 1u << scale;
  for (unsigned long long k = blockIdx.x * 256ull + threadIdx.x; k < pairs;
       k += (unsigned long long)gridDim.x * 256ull) {
    unsigned u, v;
    rmat_pair(scale, seed, k, u, v);
    keys[2 * k] = u;
    slots[2 * k] = (unsigned)(2 * k);
    keys[2 * k + 1] = (symmetrize && u != v) ? v : sentinel;
    slots[2 * k + 1] = (unsigned)(2 * k + 1);
  }
}

__global__ void __launch_bounds__(256)
    fill_edges_kernel(unsigned scale, unsigned long long seed, unsigned long long wseed,
                      const unsigned* sorted_slots, unsigned long long nnz, int* col, float* val) {
  for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < nnz;
       i += (unsigned long long)gridDim.x * 256ull) {
    const unsigned slot = sorted_slots[i];
    const unsigned long long k = slot >> 1;
    unsigned u, v;
    rmat_pair(scale, seed, k, u, v);
    col[i] = (int)((slot & 1u) ? u : v);
    val[i] = rmat_weight(wseed, k);
  }
}

/// offsets[r] = first position whose key is >= r, for r in [0, n_keys] (keys sorted).
__global__ void __launch_bounds__(256)
    offsets_kernel(const unsigned* sorted_keys, unsigned long long total, unsigned n_rows,
                   int* offsets) {
  for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i <= total;
       i += (unsigned long long)gridDim.x * 256ull) {
    const unsigned hi = (i < total) ? sorted_keys[i] : n_rows + 1;  // one past the sentinel
    const unsigned lo = (i == 0) ? 0u : sorted_keys[i - 1] + 1;
    for (unsigned r = lo; r <= hi && r <= n_rows; ++r)
      offsets[r] = (int)i;
  }
}

}  // namespace

namespace essentials_amd {

/// Builds the full graph on the current device; returns an owning grx_graph_s.
std::unique_ptr<grx_graph_s> rmat_build(gcuda::standard_context_t& c, unsigned scale,
                                        unsigned edge_factor, unsigned long long seed,
                                        unsigned long long wseed, int symmetrize) {
  error::throw_if_exception(scale < 1 || scale > 30, "rmat: scale out of range");
  const unsigned long long pairs = (unsigned long long)edge_factor << scale;
  const unsigned long long slots = 2 * pairs;
  error::throw_if_exception(slots >= (1ull << 32), "rmat: more than 2^32 slots not supported yet");
  const unsigned n = 1u << scale;
  hipStream_t s = c.stream();
  hip::buffer_t<unsigned> keys(slots), idx(slots), keys2(slots), idx2(slots);
  const unsigned grid = (unsigned)c.compute_units() * 8;
  emit_kernel<<<grid, 256, 0, s>>>(scale, seed, pairs, symmetrize, keys.data(), idx.data());
  GRX_HIP_CHECK(hipGetLastError());
  std::size_t bytes = 0;
  GRX_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys.data(), keys2.data(), idx.data(),
                                          idx2.data(), slots, 0, scale + 1, s));
  hip::buffer_t<unsigned char> temp(bytes < 256 ? 256 : bytes);  // never null: rocPRIM would only size
  GRX_HIP_CHECK(rocprim::radix_sort_pairs(temp.data(), bytes, keys.data(), keys2.data(), idx.data(),
                                          idx2.data(), slots, 0, scale + 1, s));
  auto g = std::make_unique<grx_graph_s>();
  g->n_rows = (int32_t)n;
  g->n_cols = (int32_t)n;
  g->ap.resize((std::size_t)n + 2);  // [n] = nnz, [n+1] = slots (past the sentinel run)
  offsets_kernel<<<grid, 256, 0, s>>>(keys2.data(), slots, n, g->ap.data());
  GRX_HIP_CHECK(hipGetLastError());
  int nnz = 0;
  GRX_HIP_CHECK(hipMemcpyAsync(&nnz, g->ap.data() + n, sizeof(int), hipMemcpyDeviceToHost, s));
  GRX_HIP_CHECK(hipStreamSynchronize(s));
  error::throw_if_exception(nnz < 0, "rmat: edge count overflows int32");
  g->nnz = nnz;
  g->aj.resize((std::size_t)nnz);
  g->ax.resize((std::size_t)nnz);
  if (nnz) {
    fill_edges_kernel<<<grid, 256, 0, s>>>(scale, seed, wseed, idx2.data(), (unsigned long long)nnz,
                                           g->aj.data(), g->ax.data());
    GRX_HIP_CHECK(hipGetLastError());
  }
  GRX_HIP_CHECK(hipStreamSynchronize(s));
  g->adopt();
  return g;
}

}  // namespace essentials_amd

namespace {
/// key[e] = (row << 32) | col for every edge: rows from the offsets by binary search
__global__ void __launch_bounds__(256)
    row_col_keys_kernel(const int* ap, const int* aj, int n_rows, long long nnz,
                        unsigned long long* keys) {
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < nnz; e += (long long)gridDim.x * 256) {
    int lo = 0, hi = n_rows;  // ap[lo] <= e < ap[hi]
    while (hi - lo > 1) {
      const int mid = lo + (hi - lo) / 2;
      if ((long long)ap[mid] <= e)
        lo = mid;
      else
        hi = mid;
    }
    keys[e] = ((unsigned long long)(unsigned)lo << 32) | (unsigned)aj[e];
  }
}
__global__ void __launch_bounds__(256)
    keys_to_cols_kernel(const unsigned long long* keys, long long nnz, int* aj) {
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < nnz; e += (long long)gridDim.x * 256)
    aj[e] = (int)(unsigned)(keys[e] & 0xffffffffull);
}
}  // namespace

extern "C" int grx_graph_sorted_rows(grx_context_t ctx, grx_graph_t g, grx_graph_t* out) {
  if (!ctx || !g || !out)
    return invalid("grx_graph_sorted_rows: NULL argument");
  return guarded([&] {
    auto& c = ctx->single();
    hipStream_t s = c.stream();
    auto r = std::make_unique<grx_graph_s>();
    r->n_rows = g->n_rows;
    r->n_cols = g->n_cols;
    r->nnz = g->nnz;
    r->ap.resize((std::size_t)g->n_rows + 1);
    r->aj.resize((std::size_t)std::max<int64_t>(g->nnz, 1));
    r->ax.resize((std::size_t)std::max<int64_t>(g->nnz, 1));
    GRX_HIP_CHECK(hipMemcpyAsync(r->ap.data(), g->d_ap, ((std::size_t)g->n_rows + 1) * 4,
                                 hipMemcpyDeviceToDevice, s));
    if (g->nnz) {
      const std::size_t nnz = (std::size_t)g->nnz;
      hip::buffer_t<unsigned long long> keys(nnz), keys2(nnz);
      const unsigned grid = (unsigned)c.compute_units() * 8;
      row_col_keys_kernel<<<grid, 256, 0, s>>>(g->d_ap, g->d_aj, g->n_rows, (long long)nnz, keys.data());
      GRX_HIP_CHECK(hipGetLastError());
      std::size_t bytes = 0;
      GRX_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys.data(), keys2.data(), g->d_ax,
                                              r->ax.data(), nnz, 0, 64, s));
      hip::buffer_t<unsigned char> temp(bytes < 256 ? 256 : bytes);  // never null: rocPRIM would only size
      GRX_HIP_CHECK(rocprim::radix_sort_pairs(temp.data(), bytes, keys.data(), keys2.data(), g->d_ax,
                                              r->ax.data(), nnz, 0, 64, s));
      keys_to_cols_kernel<<<grid, 256, 0, s>>>(keys2.data(), (long long)nnz, r->aj.data());
      GRX_HIP_CHECK(hipGetLastError());
    }
    GRX_HIP_CHECK(hipStreamSynchronize(s));
    r->adopt();
    *out = r.release();
    hip::block_cache_t::instance().trim();  // the sort's key buffers, see grx_graph_rmat
    return (int)GRX_OK;
  });
}

extern "C" int grx_graph_rmat(grx_context_t ctx, uint32_t scale, uint32_t edge_factor,
                              uint64_t seed, uint64_t weight_seed, int symmetrize,
                              grx_graph_t* out) {
  if (!ctx || !out)
    return invalid("grx_graph_rmat: NULL argument");
  return guarded([&] {
    auto g = rmat_build(ctx->single(), scale, edge_factor, seed, weight_seed, symmetrize);
    // the generator knows: symmetrised = its own transpose; raw pairs = a directed graph
    g->symmetry = symmetrize ? grx_graph_s::symmetric : grx_graph_s::asymmetric;
    *out = g.release();
    // the generator's sort buffers (16 B per slot: 32 GiB at scale 26) are of no use to the
    // operators: do not let them fill the frontier block cache
    hip::block_cache_t::instance().trim();
    return (int)GRX_OK;
  });
}
