/**
 * @file capi_core.hip
 * @brief C ABI: library, context, graph construction / IO, bandwidth probe.
 * See include/essentials_amd.h for the reference interface each entry stands for.
 */
#include "capi_internal.hxx"

#include <gunrock/graph/reorder.hxx>

#include <cstdio>
#include <cstring>
#include <vector>

namespace essentials_amd {
std::string& last_error() {
  static thread_local std::string msg;
  return msg;
}
}  // namespace essentials_amd

using namespace essentials_amd;

namespace {
__global__ void __launch_bounds__(256)
    offsets_max_degree_kernel(const int32_t* ap, int32_t n, unsigned long long* out) {
  unsigned long long local = 0;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const unsigned long long d = (unsigned long long)(ap[i + 1] - ap[i]);
    local = d > local ? d : local;
  }
  local = gunrock::hip::wave_max(local);
  if ((threadIdx.x & 63) == 0 && local)
    atomicMax(out, local);
}
}  // namespace

unsigned long long essentials_amd::reduce_max_degree(const int32_t* d_row_offsets, int32_t n_rows) {
  if (!d_row_offsets || n_rows < 1)
    return 0;
  gunrock::hip::buffer_t<unsigned long long> out(1);
  GRX_HIP_CHECK(hipMemset(out.data(), 0, sizeof(unsigned long long)));
  const int64_t blocks = ((int64_t)n_rows + 255) / 256;
  offsets_max_degree_kernel<<<(unsigned)(blocks > 1024 ? 1024 : blocks), 256>>>(d_row_offsets, n_rows,
                                                                                 out.data());
  GRX_HIP_CHECK(hipGetLastError());
  unsigned long long md = 0;
  GRX_HIP_CHECK(hipMemcpy(&md, out.data(), sizeof md, hipMemcpyDeviceToHost));
  return md;
}

namespace {
__global__ void __launch_bounds__(256)
    count_differences_kernel(const int32_t* a, const int32_t* b, long long n, unsigned long long* out) {
  unsigned long long local = 0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    local += a[i] != b[i];
  local = gunrock::hip::wave_sum(local);
  if ((threadIdx.x & 63) == 0 && local)
    atomicAdd(out, local);
}
}  // namespace

int essentials_amd::ensure_can_pull(grx_context_s* ctx, grx_graph_s* g) {
  if (g->in_edges || g->symmetry == grx_graph_s::symmetric)
    return GRX_OK;
  if (g->symmetry == grx_graph_s::symmetry_unknown) {
    // the CSR is its own transpose iff its transpose (in-neighbours in ascending order) equals
    // the CSR with every row sorted by column: compare offsets, then indices
    auto& sc = ctx->single();
    graph_type G = g->view();
    auto T = graph::build::transpose(G, sc);
    grx_graph_t sorted = nullptr;
    int rc = grx_graph_sorted_rows(ctx, g, &sorted);
    error::throw_if_exception(rc != GRX_OK, "symmetry check: sorting the rows failed");
    std::unique_ptr<grx_graph_s> owner(sorted);
    hip::buffer_t<unsigned long long> diff(1);
    GRX_HIP_CHECK(hipMemsetAsync(diff.data(), 0, 8, sc.stream()));
    const unsigned grid = (unsigned)sc.compute_units() * 8;
    count_differences_kernel<<<grid, 256, 0, sc.stream()>>>(T.offsets.data(), g->d_ap,
                                                            (long long)g->n_rows + 1, diff.data());
    if (g->nnz)
      count_differences_kernel<<<grid, 256, 0, sc.stream()>>>(T.indices.data(), sorted->d_aj,
                                                              (long long)g->nnz, diff.data());
    GRX_HIP_CHECK(hipGetLastError());
    unsigned long long d = 0;
    GRX_HIP_CHECK(hipMemcpyAsync(&d, diff.data(), 8, hipMemcpyDeviceToHost, sc.stream()));
    sc.synchronize();
    g->symmetry = d == 0 ? grx_graph_s::symmetric : grx_graph_s::asymmetric;
  }
  if (g->symmetry == grx_graph_s::asymmetric)
    return unsupported("pull traversal of a DIRECTED graph needs its in-edges: call "
                       "grx_graph_build_in_edges first (the CSR is not its own transpose)");
  return GRX_OK;
}

grx_graph_s* essentials_amd::hot_copy(grx_context_s* ctx, grx_graph_s* g, bool csr_only) {
  std::lock_guard<std::mutex> lock(g->hot_mutex);
  if (g->in_edges && !csr_only)
    return nullptr;  // attached after the copy was made: the copy has no transpose
  if (g->hot) {
    g->hot->symmetry = g->symmetry;  // may have been verified since
    return g->hot_first == 0 ? nullptr : g->hot.get();
  }
  int want = g->hot_first;
  if (want < 0) {
    // automatic: graphs whose labels outgrow a CU's LDS image and whose traversal is worth the
    // second copy of the CSR; a graph with attached in-edges (directed) keeps its numbering -- the
    // transpose would have to be renumbered too
    want = g->n_rows == g->n_cols && g->n_rows >= (1 << 16) && g->nnz >= (1 << 20) && (!g->in_edges || csr_only);
    if (const char* e = std::getenv("GRX_HOT_FIRST"))
      want = std::atoi(e) != 0 && g->n_rows == g->n_cols && (!g->in_edges || csr_only);
  }
  if (!want || g->n_rows < 2)
    return nullptr;
  graph_type G = g->view();
  auto R = graph::build::hot_first(G, ctx->single(), g->max_degree);
  auto h = std::make_unique<grx_graph_s>();
  h->n_rows = g->n_rows;
  h->n_cols = g->n_cols;
  h->nnz = g->nnz;
  h->ap = std::move(R.offsets);
  h->aj = std::move(R.indices);
  h->ax = std::move(R.values);
  h->adopt();
  h->max_degree = g->max_degree;
  h->max_degree_known = true;
  h->symmetry = g->symmetry;  // renumbering keeps (a)symmetry
  h->hot_first = 0;           // a copy has no copy of its own
  {  // descending degree order: the vertices with edges come first -- how many are there?
    const int32_t* hap = h->d_ap;
    h->leading_connected = hip::transform_reduce(
        (std::size_t)g->n_rows,
        [hap] __device__(std::size_t i) -> unsigned long long { return hap[i + 1] > hap[i] ? 1ull : 0ull; },
        0ull, rocprim::plus<unsigned long long>(), ctx->single());
  }
  g->hot_rank_of.resize((std::size_t)g->n_rows);
  GRX_HIP_CHECK(hipMemcpy(g->hot_rank_of.data(), R.rank_of.data(), (std::size_t)g->n_rows * 4,
                          hipMemcpyDeviceToHost));
  g->hot_vertex_of = std::move(R.vertex_of);
  g->hot_rank_of_device = std::move(R.rank_of);
  g->hot = std::move(h);
  return g->hot.get();
}

extern "C" {

int grx_graph_hot_first(grx_context_t ctx, grx_graph_t g, int enable) {
  if (!ctx || !g)
    return invalid("grx_graph_hot_first: NULL argument");
  return guarded([&] {
    if (!enable) {
      std::lock_guard<std::mutex> lock(g->hot_mutex);
      g->hot_first = 0;
      g->hot.reset();
      g->hot_vertex_of = hip::device_array_t<int32_t>();
      g->hot_rank_of_device = hip::device_array_t<int32_t>();
      g->hot_rank_of.clear();
      g->hot_rank_of.shrink_to_fit();
      return (int)GRX_OK;
    }
    if (g->n_rows != g->n_cols)
      return unsupported("grx_graph_hot_first: the graph is not square");
    if (g->in_edges)
      return unsupported("grx_graph_hot_first: a graph with attached in-edges keeps its numbering");
    g->hot_first = 1;
    hot_copy(ctx, g);
    return (int)GRX_OK;
  });
}

int grx_abi_version(void) { return GRX_ABI_VERSION; }
const char* grx_last_error(void) { return last_error().c_str(); }

void grx_default_options(grx_options* opt) {
  if (!opt)
    return;
  std::memset(opt, 0, sizeof *opt);
  opt->load_balance = GRX_LB_BLOCK_MAPPED;
  opt->frontier_sizing_factor = 1.5f;
}

int grx_context_create(int device, void* stream, grx_context_t* out) {
  if (!out)
    return invalid("grx_context_create: out is NULL");
  return guarded([&] {
    int count = 0;
    GRX_HIP_CHECK(hipGetDeviceCount(&count));
    error::throw_if_exception(device < 0 || device >= count, "grx_context_create: no such device");
    auto* c = new grx_context_s;
    c->device = device;
    if (stream)
      c->mc = std::make_shared<gcuda::multi_context_t>(device, (hipStream_t)stream);
    else
      c->mc = std::make_shared<gcuda::multi_context_t>(device);
    *out = c;
    return (int)GRX_OK;
  });
}

int grx_context_destroy(grx_context_t ctx) {
  if (!ctx)
    return GRX_OK;
  return guarded([&] {
    delete ctx;
    return (int)GRX_OK;
  });
}

int grx_context_synchronize(grx_context_t ctx) {
  if (!ctx)
    return invalid("context is NULL");
  return guarded([&] {
    ctx->single().synchronize();
    return (int)GRX_OK;
  });
}

int grx_context_wait_stream(grx_context_t ctx, void* other_stream) {
  if (!ctx)
    return invalid("context is NULL");
  return guarded([&] {
    auto& c = ctx->single();
    if ((hipStream_t)other_stream == c.stream())
      return (int)GRX_OK;
    // device-side ordering only: nobody waits on the host
    GRX_HIP_CHECK(hipEventRecord(c.event(), (hipStream_t)other_stream));
    GRX_HIP_CHECK(hipStreamWaitEvent(c.stream(), c.event(), 0));
    return (int)GRX_OK;
  });
}

int grx_copy_to_host(grx_context_t ctx, void* h_dst, const void* d_src, size_t bytes) {
  if (!ctx || (bytes && (!h_dst || !d_src)))
    return invalid("grx_copy_to_host: bad arguments");
  return guarded([&] {
    auto& c = ctx->single();
    if (bytes)
      GRX_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c.stream()));
    c.synchronize();
    return (int)GRX_OK;
  });
}

int grx_copy_to_device(grx_context_t ctx, void* d_dst, const void* h_src, size_t bytes) {
  if (!ctx || (bytes && (!d_dst || !h_src)))
    return invalid("grx_copy_to_device: bad arguments");
  return guarded([&] {
    auto& c = ctx->single();
    if (bytes)
      GRX_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c.stream()));
    c.synchronize();  // pageable source: the staging copy AND the DMA have finished
    return (int)GRX_OK;
  });
}

int grx_trim_cache(void) {
  return guarded([&] {
    hip::block_cache_t::instance().trim();
    return (int)GRX_OK;
  });
}

int grx_context_device_info(grx_context_t ctx, int32_t* cus, int32_t* wave, int64_t* mem,
                            char* name, size_t name_len) {
  if (!ctx)
    return invalid("context is NULL");
  const hipDeviceProp_t& p = ctx->single().props();
  if (cus) *cus = p.multiProcessorCount;
  if (wave) *wave = p.warpSize;
  if (mem) *mem = (int64_t)p.totalGlobalMem;
  if (name && name_len) {
    std::strncpy(name, p.name, name_len - 1);
    name[name_len - 1] = 0;
  }
  return GRX_OK;
}

// ---- graphs ------------------------------------------------------------------

int grx_graph_from_device_csr(int32_t n_rows, int32_t n_cols, int32_t nnz, const int32_t* d_ap,
                              const int32_t* d_aj, const float* d_ax, grx_graph_t* out) {
  if (!out || n_rows < 0 || nnz < 0 || !d_ap || (nnz && (!d_aj || !d_ax)))
    return invalid("grx_graph_from_device_csr: bad arguments");
  return guarded([&] {
    auto* g = new grx_graph_s;
    g->n_rows = n_rows; g->n_cols = n_cols; g->nnz = nnz;
    g->d_ap = d_ap; g->d_aj = d_aj; g->d_ax = d_ax;
    *out = g;
    return (int)GRX_OK;
  });
}

int grx_graph_from_host_csr(int32_t n_rows, int32_t n_cols, int32_t nnz, const int32_t* h_ap,
                            const int32_t* h_aj, const float* h_ax, grx_graph_t* out) {
  if (!out || n_rows < 0 || nnz < 0 || !h_ap || (nnz && (!h_aj || !h_ax)))
    return invalid("grx_graph_from_host_csr: bad arguments");
  return guarded([&] {
    auto g = std::make_unique<grx_graph_s>();
    g->n_rows = n_rows; g->n_cols = n_cols; g->nnz = nnz;
    g->ap.assign(h_ap, (std::size_t)n_rows + 1);
    g->aj.assign(h_aj, (std::size_t)nnz);
    g->ax.assign(h_ax, (std::size_t)nnz);
    g->adopt();
    *out = g.release();
    return (int)GRX_OK;
  });
}

int grx_graph_from_mtx(const char* path, grx_graph_t* out) {
  if (!out || !path)
    return invalid("grx_graph_from_mtx: bad arguments");
  return guarded([&] {
    io::matrix_market_t<vertex_t, edge_t, weight_t> mm;
    format::csr_t<memory_space_t::host, vertex_t, edge_t, weight_t> csr;
    csr.from_coo(mm.load(path));
    int rc = grx_graph_from_host_csr(csr.number_of_rows, csr.number_of_columns, csr.number_of_nonzeros,
                                     csr.row_offsets.data(), csr.column_indices.data(),
                                     csr.nonzero_values.data(), out);
    // a "symmetric" file is expanded to both directions by the loader: its own transpose
    if (rc == GRX_OK && mm.scheme == io::symmetric)
      (*out)->symmetry = grx_graph_s::symmetric;
    return rc;
  });
}

int grx_graph_from_csr_file(const char* path, grx_graph_t* out) {
  if (!out || !path)
    return invalid("grx_graph_from_csr_file: bad arguments");
  return guarded([&] {
    format::csr_t<memory_space_t::host, vertex_t, edge_t, weight_t> csr;
    csr.read_binary(path);
    return grx_graph_from_host_csr(csr.number_of_rows, csr.number_of_columns, csr.number_of_nonzeros,
                                   csr.row_offsets.data(), csr.column_indices.data(),
                                   csr.nonzero_values.data(), out);
  });
}

int grx_graph_copy_to_host(grx_graph_t g, int32_t* h_ap, int32_t* h_aj, float* h_ax) {
  if (!g)
    return invalid("graph is NULL");
  return guarded([&] {
    if (h_ap)
      GRX_HIP_CHECK(hipMemcpy(h_ap, g->d_ap, ((std::size_t)g->n_rows + 1) * 4, hipMemcpyDeviceToHost));
    if (h_aj && g->nnz)
      GRX_HIP_CHECK(hipMemcpy(h_aj, g->d_aj, (std::size_t)g->nnz * 4, hipMemcpyDeviceToHost));
    if (h_ax && g->nnz)
      GRX_HIP_CHECK(hipMemcpy(h_ax, g->d_ax, (std::size_t)g->nnz * 4, hipMemcpyDeviceToHost));
    return (int)GRX_OK;
  });
}

int grx_graph_write_csr_file(grx_graph_t g, const char* path) {
  if (!g || !path)
    return invalid("grx_graph_write_csr_file: bad arguments");
  return guarded([&] {
    std::vector<int32_t> ap((std::size_t)g->n_rows + 1), aj((std::size_t)g->nnz);
    std::vector<float> ax((std::size_t)g->nnz);
    int rc = grx_graph_copy_to_host(g, ap.data(), aj.data(), ax.data());
    if (rc)
      return rc;
    FILE* f = std::fopen(path, "wb");
    error::throw_if_exception(f == nullptr, std::string("cannot open ") + path);
    int32_t nnz32 = (int32_t)g->nnz;
    std::fwrite(&g->n_rows, 4, 1, f);
    std::fwrite(&g->n_cols, 4, 1, f);
    std::fwrite(&nnz32, 4, 1, f);
    std::fwrite(ap.data(), 4, ap.size(), f);
    std::fwrite(aj.data(), 4, aj.size(), f);
    std::fwrite(ax.data(), 4, ax.size(), f);
    std::fclose(f);
    return (int)GRX_OK;
  });
}

int grx_graph_build_in_edges(grx_context_t ctx, grx_graph_t g) {
  if (!ctx || !g)
    return invalid("grx_graph_build_in_edges: NULL argument");
  return guarded([&] {
    if (g->in_edges)
      return (int)GRX_OK;
    graph_type G = g->view();
    using T = graph::transposed_t<vertex_t, edge_t, weight_t>;
    g->in_edges = std::make_unique<T>(graph::build::transpose(G, ctx->single()));
    return (int)GRX_OK;
  });
}

int grx_graph_destroy(grx_graph_t g) {
  if (!g)
    return GRX_OK;
  return guarded([&] {
    delete g;
    return (int)GRX_OK;
  });
}

int grx_graph_info(grx_graph_t g, int32_t* n_rows, int32_t* n_cols, int64_t* nnz,
                   const int32_t** d_ap, const int32_t** d_aj, const float** d_ax) {
  if (!g)
    return invalid("graph is NULL");
  if (n_rows) *n_rows = g->n_rows;
  if (n_cols) *n_cols = g->n_cols;
  if (nnz) *nnz = g->nnz;
  if (d_ap) *d_ap = g->d_ap;
  if (d_aj) *d_aj = g->d_aj;
  if (d_ax) *d_ax = g->d_ax;
  return GRX_OK;
}

// ---- achievable-HBM roof -----------------------------------------------------

namespace {
__global__ void __launch_bounds__(256) copy16_kernel(const uint4* __restrict__ in,
                                                     uint4* __restrict__ out, std::size_t n) {
  for (std::size_t i = blockIdx.x * (std::size_t)256 + threadIdx.x; i < n;
       i += (std::size_t)gridDim.x * 256)
    out[i] = in[i];
}
/// acc += table[columns[i]]: the access pattern every label-testing advance functor shares, with
/// none of the frontier logic, atomics or output (MODE 0 = plain L1-cached load, 1 = the
/// agent-scope relaxed load math::atomic::min pre-tests with).
__global__ void __launch_bounds__(256)
    gather_probe_kernel(const int32_t* columns, std::size_t n, const int32_t* table,
                        unsigned long long* sink, int MODE) {
  unsigned long long acc = 0;
  const std::size_t stride = (std::size_t)gridDim.x * 256;
  for (std::size_t i = blockIdx.x * (std::size_t)256 + threadIdx.x; i < n; i += stride) {
    const int32_t j = __builtin_nontemporal_load(columns + i);
    acc += (unsigned)(MODE == 0 ? table[j]
                                : __hip_atomic_load(table + j, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT));
  }
  if (acc == 0x5eed5eed5eed5eedull)  // never true for the all-ones table: keeps the loads alive
    *sink = acc;
}
}  // namespace

int grx_measure_gather_rate(grx_context_t ctx, grx_graph_t g, int mode, int repeats,
                            double* lookups_per_second) {
  if (!ctx || !g || !lookups_per_second || repeats < 1 || mode < 0 || mode > 1)
    return invalid("grx_measure_gather_rate: bad arguments");
  if (g->nnz < 1)
    return invalid("grx_measure_gather_rate: graph has no edges");
  return guarded([&] {
    auto& c = ctx->single();
    const std::size_t n = (std::size_t)g->nnz;
    const std::size_t cols = (std::size_t)std::max(g->n_cols, g->n_rows);
    hip::buffer_t<int32_t> table(cols);
    hip::buffer_t<unsigned long long> sink(1);
    GRX_HIP_CHECK(hipMemsetAsync(table.data(), 1, cols * 4, c.stream()));
    const unsigned grid = (unsigned)c.compute_units() * 8;
    auto launch = [&] {
      gather_probe_kernel<<<grid, 256, 0, c.stream()>>>(g->d_aj, n, table.data(), sink.data(), mode);
    };
    launch();
    c.synchronize();
    float best = 0;
    for (int r = 0; r < repeats; ++r) {
      util::timer_t t(c.stream());
      t.begin();
      launch();
      GRX_HIP_CHECK(hipGetLastError());
      const float ms = t.end();
      if (r == 0 || ms < best)
        best = ms;
    }
    *lookups_per_second = (double)n / ((double)best * 1e-3);
    return (int)GRX_OK;
  });
}

int grx_measure_copy_bandwidth(grx_context_t ctx, size_t bytes, int repeats, double* gbps) {
  if (!ctx || !gbps || bytes < 4096 || repeats < 1)
    return invalid("grx_measure_copy_bandwidth: bad arguments");
  return guarded([&] {
    auto& c = ctx->single();
    const std::size_t n = bytes / 16;
    hip::buffer_t<uint4> a(n), b(n);
    GRX_HIP_CHECK(hipMemsetAsync(a.data(), 1, n * 16, c.stream()));
    copy16_kernel<<<(unsigned)c.compute_units() * 8, 256, 0, c.stream()>>>(a.data(), b.data(), n);
    c.synchronize();
    util::timer_t t(c.stream());
    t.begin();
    for (int r = 0; r < repeats; ++r)
      copy16_kernel<<<(unsigned)c.compute_units() * 8, 256, 0, c.stream()>>>(a.data(), b.data(), n);
    GRX_HIP_CHECK(hipGetLastError());
    float ms = t.end();
    *gbps = (2.0 * (double)n * 16.0 * repeats) / ((double)ms * 1e-3) / 1e9;  // read + write
    return (int)GRX_OK;
  });
}

}  // extern "C"
