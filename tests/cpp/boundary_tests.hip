// boundary_tests.hip -- the include paths and harness symbols of the reference's public surface that
// sit beside the hot path: gunrock/memory.hxx, error.hxx, container/array.hxx,
// cuda/launch_box.hxx, io/sample.hxx, io/smtx.hxx, util/print.hxx -- each included by the path the
// reference spells.  Expectations follow the reference's own unit tests where it has them
// (unittests/cuda/launch_box.cuh:8-59, unittests/io/smtx.cuh:18-37, unittests/containers/array.cuh)
// and its documented values (io/sample.hxx:20-50).  Exit code 0 = pass.
#include <gunrock/memory.hxx>
#include <gunrock/error.hxx>
#include <gunrock/container/array.hxx>
#include <gunrock/container/vector.hxx>
#include <gunrock/cuda/cuda.hxx>
#include <gunrock/cuda/context.hxx>
#include <gunrock/cuda/launch_box.hxx>
#include <gunrock/cuda/atomic_functions.hxx>
#include <gunrock/formats/csr.hxx>
#include <gunrock/formats/coo.hxx>
#include <gunrock/framework/frontier/frontier.hxx>
#include <gunrock/framework/operators/advance/advance.hxx>
#include <gunrock/framework/operators/filter/filter.hxx>
#include <gunrock/framework/operators/uniquify/uniquify.hxx>
#include <gunrock/framework/operators/for/for.hxx>
#include <gunrock/framework/operators/batch/batch.hxx>
#include <gunrock/graph/build.hxx>
#include <gunrock/io/sample.hxx>
#include <gunrock/io/smtx.hxx>
#include <gunrock/util/filepath.hxx>
#include <gunrock/util/print.hxx>
#include <gunrock/util/timer.hxx>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <unistd.h>

using namespace gunrock;
using namespace gunrock::gcuda::launch_box;

static int failures = 0;
#define CHECK(...)                                                        \
  do {                                                                    \
    if (!(__VA_ARGS__)) {                                                 \
      std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #__VA_ARGS__);  \
      ++failures;                                                         \
    }                                                                     \
  } while (0)

// the reference's own example box (unittests/cuda/launch_box.cuh:8-14): CUDA rows + a fallback
typedef launch_box_t<launch_params_t<sm_86 | sm_80, dim3_t<16, 2, 2>, dim3_t<64, 1, 4>, 2>,
                     launch_params_t<sm_75 | sm_70, dim3_t<32, 2, 4>, dim3_t<64, 8, 8>>,
                     launch_params_t<sm_61 | sm_60, dim3_t<8, 4, 4>, dim3_t<32, 1, 4>, 2>,
                     launch_params_t<sm_35, dim3_t<64>, dim3_t<64>, 16>,
                     launch_params_t<fallback, dim3_t<16>, dim3_t<2>, 4>>
    reference_box_t;
// a box that names this engine's one target ahead of its fallback
typedef launch_box_t<launch_params_t<sm_80, dim3_t<128>, dim3_t<3>>,
                     launch_params_t<gfx950, dim3_t<256>, dim3_t<1024>, 2, 64>,
                     launch_params_t<fallback, dim3_t<32>, dim3_t<1>>>
    native_box_t;

__global__ void dummy_kernel() {}
__global__ void mark_kernel(int* out, int value) {
  out[blockIdx.x * blockDim.x + threadIdx.x] = value + (int)blockIdx.x;
}
__global__ void array_kernel(gunrock::array<int, 4> a, int* out) {
  int s = 0;
  for (auto x : a)
    s += x;
  out[0] = s;
  out[1] = (int)a.size();
  out[2] = a.back();
}

static std::string captured(std::function<void()> body) {
  std::ostringstream os;
  auto* old = std::cout.rdbuf(os.rdbuf());
  body();
  std::cout.rdbuf(old);
  return os.str();
}

int main() {
  auto mc = std::make_shared<gcuda::multi_context_t>(0);
  auto& ctx = *mc->get_context(0);

  // ---- launch_box: selection ------------------------------------------------------------
  {
    static_assert(reference_box_t::block_dimensions_t::x == 16, "fallback row expected");
    CHECK(reference_box_t::block_dimensions_t::x == 16 && reference_box_t::grid_dimensions_t::x == 2 &&
          reference_box_t::items_per_thread == 4 && reference_box_t::shared_memory_bytes == 0);
    CHECK(native_box_t::block_dimensions_t::x == 256 && native_box_t::grid_dimensions_t::x == 1024 &&
          native_box_t::items_per_thread == 2 && native_box_t::shared_memory_bytes == 64);
    dimensions_t b = reference_box_t::block_dimensions_t::dimensions();
    dim3 conv = b;
    CHECK(conv.x == 16 && conv.y == 1 && conv.z == 1 && b.size() == 16);
    CHECK((dim3_t<8, 4, 4>::size() == 128));
    CHECK(occupancy<reference_box_t>(dummy_kernel) > 0.0f);
    CHECK(occupancy<native_box_t>(dummy_kernel) <= 1.0f);
  }
  // ---- launch_box: dynamic grids, strided / blocked / plain launches ----------------------
  {
    using dyn_t = launch_box_t<launch_params_dynamic_grid_t<fallback, dim3_t<128>, 3>>;
    dyn_t box;
    box.calculate_grid_dimensions_strided(1000);
    CHECK(box.grid_dimensions.x == 8);
    box.calculate_grid_dimensions_blocked(1000);
    CHECK(box.grid_dimensions.x == 3);  // ceil(1000 / (128 * 3))
    const std::size_t n = 100003;
    hip::device_array_t<int> hits(n), who(n);
    hits.zero(ctx.stream());
    int* h = hits.data();
    int* w = who.data();
    auto f = [h, w] __device__(int const& tid, int const& bid) {
      atomicAdd(&h[tid], 1);
      w[tid] = bid;
    };
    box.launch_strided(ctx, f, n);
    ctx.synchronize();
    CHECK(box.grid_dimensions.x == (n + 127) / 128);
    auto hh = hits.to_host();
    auto ww = who.to_host();
    bool once = true, owner = true;
    for (std::size_t i = 0; i < n; ++i) {
      once = once && hh[i] == 1;
      owner = owner && ww[i] == (int)(i / 128);
    }
    CHECK(once);
    CHECK(owner);
    hits.zero(ctx.stream());
    box.launch_blocked(ctx, f, n);
    ctx.synchronize();
    CHECK(box.grid_dimensions.x == (n + 383) / 384);
    hh = hits.to_host();
    once = true;
    for (std::size_t i = 0; i < n; ++i)
      once = once && hh[i] == 1;
    CHECK(once);
    // extra kernel arguments are forwarded
    auto g = [h] __device__(int const& tid, int const& bid, int add) { h[tid] = add; };
    box.launch_strided(ctx, g, 10, 7);
    ctx.synchronize();
    hh = hits.to_host();
    CHECK(hh[0] == 7 && hh[9] == 7 && hh[10] == 1);
    // static box + a __global__ function: launch(context, kernel, args...)
    using small_t = launch_box_t<launch_params_t<gfx950, dim3_t<64>, dim3_t<5>>>;
    small_t sbox;
    hip::device_array_t<int> out(64 * 5);
    sbox.launch(ctx, mark_kernel, out.data(), 100);
    ctx.synchronize();
    auto oo = out.to_host();
    CHECK(oo[0] == 100 && oo[63] == 100 && oo[64] == 101 && oo[64 * 5 - 1] == 104);
    // cooperative launch of the same kernel over 64 * 5 elements
    out.zero(ctx.stream());
    using coop_t = launch_box_t<launch_params_dynamic_grid_t<fallback, dim3_t<64>>>;
    coop_t cbox;
    int* optr = out.data();
    int value = 200;
    cbox.launch_cooperative(ctx, mark_kernel, 64 * 5, optr, value);
    ctx.synchronize();
    oo = out.to_host();
    CHECK(oo[0] == 200 && oo[64 * 5 - 1] == 204);
  }
  // ---- container/array.hxx ------------------------------------------------------------------
  {
    gunrock::array<int, 4> a = {{{1, 2, 3, 4}}};
    CHECK(a.size() == 4 && !a.empty() && a.front() == 1 && a.back() == 4 && a[2] == 3);
    gunrock::array<int, 4> b = a;
    CHECK(a == b);
    b.fill(9);
    CHECK(a != b && b[0] == 9 && b[3] == 9);
    a.swap(b);
    CHECK(a[0] == 9 && b[3] == 4);
    gunrock::array<float, 0> z;
    CHECK(z.empty() && z.size() == 0 && z.data() == nullptr);
    hip::device_array_t<int> out(3);
    array_kernel<<<1, 1, 0, ctx.stream()>>>(b, out.data());
    ctx.synchronize();
    auto o = out.to_host();
    CHECK(o[0] == 10 && o[1] == 4 && o[2] == 4);
  }
  // ---- memory.hxx ---------------------------------------------------------------------------
  {
    using namespace memory;
    float* p = nullptr;
    allocate(p, 64 * sizeof(float), memory_space_t::device);
    CHECK(p != nullptr);
    std::shared_ptr<float> owner(p, deleter_t<float>());
    CHECK(raw_pointer_cast(owner.get()) == p);
    thrust::device_vector<int> dv(8, 3);
    CHECK(raw_pointer_cast(dv.data()) == dv.data().get());
    int* hp = allocate<int>(16 * sizeof(int), memory_space_t::host);
    hp[15] = 1;
    memory::free(hp, memory_space_t::host);
    CHECK(allocate<int>(0) == nullptr);
    bool threw = false;
    try {
      error::throw_if_exception(hipErrorInvalidValue, "message");
    } catch (error::exception_t& e) {
      threw = std::string(e.what()).find("message") != std::string::npos;
    }
    CHECK(threw);
  }
  // ---- io/sample.hxx: the documented 4 x 4 matrix, in both memory spaces -----------------------
  {
    auto h = io::sample::csr<memory_space_t::host>();
    CHECK(h.number_of_rows == 4 && h.number_of_columns == 4 && h.number_of_nonzeros == 4);
    const int ap[5] = {0, 0, 2, 3, 4}, aj[4] = {0, 1, 2, 1};
    const float ax[4] = {5, 8, 3, 6};
    bool same = true;
    for (int i = 0; i < 5; ++i) same = same && h.row_offsets[i] == ap[i];
    for (int i = 0; i < 4; ++i) same = same && h.column_indices[i] == aj[i] && h.nonzero_values[i] == ax[i];
    CHECK(same);
    auto d = io::sample::csr();  // device, int / int / float
    CHECK(d.number_of_nonzeros == 4);
    same = true;
    for (int i = 0; i < 5; ++i) same = same && d.row_offsets[i] == ap[i];
    for (int i = 0; i < 4; ++i) same = same && d.column_indices[i] == aj[i] && d.nonzero_values[i] == ax[i];
    CHECK(same);
    auto G = graph::build::from_csr<memory_space_t::device, graph::view_t::csr>(d);
    CHECK(G.get_number_of_vertices() == 4 && G.get_number_of_edges() == 4);
  }
  // ---- io/smtx.hxx ------------------------------------------------------------------------------
  {
    char path[] = "/tmp/grx_smtx_XXXXXX";
    int fd = mkstemp(path);
    CHECK(fd >= 0);
    close(fd);
    const std::string file = std::string(path) + ".smtx";
    {
      std::ofstream f(file);
      f << "% Sparse matrix file format .smtx\n%\n% comment\n%\n3 4 5\n0 2 2 5\n1 3 0 1 2\n";
    }
    io::smtx_t<int, int, float> loader;
    auto csr = loader.load(file);
    CHECK(csr.number_of_rows == 3 && csr.number_of_columns == 4 && csr.number_of_nonzeros == 5);
    CHECK(csr.row_offsets[0] == 0 && csr.row_offsets[1] == 2 && csr.row_offsets[3] == 5);
    CHECK(csr.column_indices[0] == 1 && csr.column_indices[4] == 2);
    bool ranged = true;
    for (int i = 0; i < 5; ++i) ranged = ranged && csr.nonzero_values[i] >= 1.0f && csr.nonzero_values[i] < 10.0f;
    CHECK(ranged);
    auto again = io::smtx_t<int, int, float>().load(file);
    bool repeat = true;
    for (int i = 0; i < 5; ++i) repeat = repeat && again.nonzero_values[i] == csr.nonzero_values[i];
    CHECK(repeat);  // values are a function of (seed, position): the reference's are not
    CHECK(loader.dataset == util::extract_dataset(util::extract_filename(file)));
    format::csr_t<memory_space_t::device, int, int, float> on_device(csr);  // unittests/io/smtx.cuh:29
    CHECK(on_device.number_of_nonzeros == 5 && on_device.column_indices[3] == 1);
    {
      std::ofstream f(file);
      f << "3, 4, 5\n0 2 2 5\n1 3 0 1 2\n";
    }
    CHECK(io::smtx_t<int, int, float>().load(file, true).number_of_nonzeros == 5);
    {
      std::ofstream f(file);
      f << "3 4 5\n0 2 2\n1 3 0 1 2\n";  // one row offset short
    }
    bool threw = false;
    try { io::smtx_t<int, int, float>().load(file); } catch (error::exception_t&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { io::smtx_t<int, int, float>().load(file + ".missing"); } catch (error::exception_t&) { threw = true; }
    CHECK(threw);
    std::remove(file.c_str());
    std::remove(path);
    CHECK(util::is_market("a/b.mtx") && util::is_market("x.mmio") && util::is_binary_csr("g.csr") &&
          !util::is_market("g.csr"));
  }
  // ---- util/print.hxx ---------------------------------------------------------------------------
  {
    thrust::device_vector<int> dv(50);
    thrust::host_vector<float> hv(3);
    for (int i = 0; i < 50; ++i) dv[i] = i * 2;
    hv[0] = 1.5f; hv[1] = 2.5f; hv[2] = 3.5f;
    CHECK(captured([&] { print::head(dv, 4, "GPU distances"); }) == "GPU distances[:4] = 0 2 4 6 \n");
    CHECK(captured([&] { print::head(hv, 40, "h"); }) == "h[:3] = 1.5 2.5 3.5 \n");
    CHECK(captured([&] { print::head(hv, 2); }) == "1.5 2.5 \n");
    CHECK(captured([&] { print::head(dv.data().get(), 3, 50, "ptr"); }) == "ptr[:3] = 0 2 4 \n");
    int on_host[4] = {7, 8, 9, 10};
    CHECK(captured([&] { print::head(on_host, 10, 4, "host"); }) == "host[:4] = 7 8 9 10 \n");
  }
  // ---- util/timer.hxx through its reference path ----------------------------------------------------
  {
    util::timer_t t(ctx.stream());
    t.begin();
    dummy_kernel<<<1, 64, 0, ctx.stream()>>>();
    CHECK(t.end() >= 0.0f && t.milliseconds() == t.time);
  }

  std::printf(failures ? "boundary_tests: %d FAILURES\n" : "boundary_tests: all passed\n", failures);
  return failures ? 1 : 0;
}
