/**
 * @file launch_box.hxx
 * @brief gcuda::launch_box -- compile-time kernel launch geometry.
 *
 * Surface of reference cuda/launch_box.hxx:116-141,194-335 (+ cuda/sm.hxx, cuda/detail/*):
 * dim3_t / dimensions_t, launch_params_t (static grid), launch_params_dynamic_grid_t
 * (calculate_grid_dimensions_strided / _blocked), launch_box_t<params...> that resolves to the
 * FIRST parameter set matching the architecture being compiled for, its launch / launch_strided /
 * launch_blocked / launch_cooperative members on the context's stream, and occupancy<box>(kernel).
 *
 * There is ONE architecture here: gfx950 (MI355X).  The flag set therefore has `gfx950` and
 * `fallback`; the reference's sm_XX enumerators are kept as flags that never match, so a box
 * written for the reference (sm_XX rows + a fallback row) compiles and resolves to its fallback
 * row, and a box that names gfx950 resolves to that row.  Selection is a constexpr search over
 * the pack (no tuple machinery); a box without any matching row is a static_assert, as in the
 * reference (cuda/detail/launch_box.hxx:58-62).
 *
 * The engine's own operators do not use a launch box: their persistent grids are sized from the
 * device's CU count at run time (advance.hxx).  The box is the documented way for CLIENT code to
 * launch its own kernels on the engine's stream (reference block_mapped.hxx:181-203 is its only
 * in-tree user).
 */
#pragma once

#include <cstddef>
#include <type_traits>
#include <utility>

#include <gunrock/hip/context.hxx>

namespace gunrock {
namespace gcuda {
namespace launch_box {

/// Architecture flags.  Bit 31 is the one real target; `fallback` matches everything.
enum sm_flag_t : unsigned {
  fallback = ~0u,
  gfx950 = 1u << 31,
  // the reference's CUDA targets: valid spellings, never selected on this engine
  sm_30 = 1u << 0, sm_35 = 1u << 1, sm_37 = 1u << 2, sm_50 = 1u << 3, sm_52 = 1u << 4,
  sm_53 = 1u << 5, sm_60 = 1u << 6, sm_61 = 1u << 7, sm_62 = 1u << 8, sm_70 = 1u << 9,
  sm_72 = 1u << 10, sm_75 = 1u << 11, sm_80 = 1u << 12, sm_86 = 1u << 13
};
constexpr sm_flag_t operator|(sm_flag_t a, sm_flag_t b) {
  return static_cast<sm_flag_t>(static_cast<unsigned>(a) | static_cast<unsigned>(b));
}
constexpr sm_flag_t operator&(sm_flag_t a, sm_flag_t b) {
  return static_cast<sm_flag_t>(static_cast<unsigned>(a) & static_cast<unsigned>(b));
}
/// The architecture this translation unit is compiled for.
constexpr sm_flag_t target = gfx950;

struct dimensions_t {
  unsigned int x, y, z;
  __host__ __device__ constexpr dimensions_t(unsigned int _x = 1, unsigned int _y = 1,
                                             unsigned int _z = 1)
      : x(_x), y(_y), z(_z) {}
  __host__ __device__ constexpr unsigned int size() const { return x * y * z; }
  __host__ __device__ operator dim3() const { return dim3(x, y, z); }
};

/// dim3 as a type (a dim3 value cannot be a template argument).
template <unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1>
struct dim3_t {
  enum : unsigned int { x = x_, y = y_, z = z_ };
  static constexpr unsigned int size() { return x_ * y_ * z_; }
  static constexpr dimensions_t dimensions() { return dimensions_t(x_, y_, z_); }
  constexpr operator dimensions_t() const { return dimensions_t(x_, y_, z_); }
};

namespace detail {
template <sm_flag_t flags_, std::size_t items_per_thread_, std::size_t shared_memory_bytes_>
struct launch_params_base_t {
  static constexpr sm_flag_t sm_flags = flags_;
  static constexpr std::size_t items_per_thread = items_per_thread_;
  static constexpr std::size_t shared_memory_bytes = shared_memory_bytes_;
};

/// Index of the first parameter set whose flags include the target (sizeof...(lp) if none).
template <typename... lp_v>
constexpr std::size_t first_match() {
  constexpr bool hit[] = {((static_cast<unsigned>(lp_v::sm_flags) & static_cast<unsigned>(target)) != 0)..., false};
  std::size_t i = 0;
  while (i < sizeof...(lp_v) && !hit[i])
    ++i;
  return i;
}

template <std::size_t i, typename... lp_v>
struct pick_t;
template <std::size_t i, typename head_t, typename... tail_v>
struct pick_t<i, head_t, tail_v...> : pick_t<i - 1, tail_v...> {};
template <typename head_t, typename... tail_v>
struct pick_t<0, head_t, tail_v...> {
  using type = head_t;
};
template <std::size_t i>
struct pick_t<i> {
  static_assert(i != i, "Launch box could not find valid launch parameters");
};

template <typename func_t, typename... args_t>
__global__ void strided_kernel(func_t f, const std::size_t bound, args_t... args) {
  const std::size_t stride = (std::size_t)blockDim.x * gridDim.x;
  for (std::size_t i = (std::size_t)blockIdx.x * blockDim.x + threadIdx.x; i < bound; i += stride)
    f((int)i, (int)blockIdx.x, args...);
}

template <unsigned int items_per_thread, typename func_t, typename... args_t>
__global__ void blocked_kernel(func_t f, const std::size_t bound, args_t... args) {
  const std::size_t stride = (std::size_t)blockDim.x * gridDim.x;
  for (std::size_t i = (std::size_t)blockIdx.x * blockDim.x + threadIdx.x; i < bound;
       i += stride * items_per_thread) {
#pragma unroll
    for (unsigned int j = 0; j < items_per_thread; ++j)
      if (i + stride * j < bound)
        f((int)(i + stride * j), (int)blockIdx.x, args...);
  }
}
}  // namespace detail

/// Static block AND grid dimensions.
template <sm_flag_t flags_, typename block_dimensions_, typename grid_dimensions_,
          std::size_t items_per_thread_ = 1, std::size_t shared_memory_bytes_ = 0>
struct launch_params_t : detail::launch_params_base_t<flags_, items_per_thread_, shared_memory_bytes_> {
  typedef block_dimensions_ block_dimensions_t;
  typedef grid_dimensions_ grid_dimensions_t;
  static constexpr dimensions_t block_dimensions = block_dimensions_t::dimensions();
  static constexpr dimensions_t grid_dimensions = grid_dimensions_t::dimensions();
};

/// Static block dimensions, grid computed from the element count at run time.
template <sm_flag_t flags_, typename block_dimensions_, std::size_t items_per_thread_ = 1,
          std::size_t shared_memory_bytes_ = 0>
struct launch_params_dynamic_grid_t
    : detail::launch_params_base_t<flags_, items_per_thread_, shared_memory_bytes_> {
  typedef detail::launch_params_base_t<flags_, items_per_thread_, shared_memory_bytes_> base_t;
  typedef block_dimensions_ block_dimensions_t;
  static constexpr dimensions_t block_dimensions = block_dimensions_t::dimensions();
  dimensions_t grid_dimensions;

  /// one element per thread: ceil(n / block)
  void calculate_grid_dimensions_strided(std::size_t num_elements) {
    grid_dimensions = dimensions_t(
        (unsigned)((num_elements + block_dimensions.x - 1) / block_dimensions.x), 1, 1);
  }
  /// items_per_thread elements per thread: ceil(n / (block * items))
  void calculate_grid_dimensions_blocked(std::size_t num_elements) {
    const std::size_t per_block = (std::size_t)block_dimensions.x * base_t::items_per_thread;
    grid_dimensions = dimensions_t((unsigned)((num_elements + per_block - 1) / per_block), 1, 1);
  }
};

template <typename... lp_v>
using select_launch_params_t = typename detail::pick_t<detail::first_match<lp_v...>(), lp_v...>::type;

/**
 * @brief A pack of launch parameter sets; the box IS the first set that matches the target.
 */
template <typename... lp_v>
struct launch_box_t : public select_launch_params_t<lp_v...> {
  typedef select_launch_params_t<lp_v...> params_t;
  launch_box_t() {}

  /// f(thread id, block id, args...) for every id < num_elements, one per thread, grid-strided.
  template <typename func_t, typename... args_t>
  void launch_strided(gcuda::standard_context_t& context, func_t& f, const std::size_t num_elements,
                      args_t&&... args) {
    params_t::calculate_grid_dimensions_strided(num_elements);
    if (num_elements == 0)
      return;
    detail::strided_kernel<<<dim3(params_t::grid_dimensions), dim3(params_t::block_dimensions),
                             params_t::shared_memory_bytes, context.stream()>>>(
        f, num_elements, std::forward<args_t>(args)...);
    GRX_HIP_CHECK(hipGetLastError());
  }

  /// The same with items_per_thread ids per thread.
  template <typename func_t, typename... args_t>
  void launch_blocked(gcuda::standard_context_t& context, func_t& f, const std::size_t num_elements,
                      args_t&&... args) {
    params_t::calculate_grid_dimensions_blocked(num_elements);
    if (num_elements == 0)
      return;
    detail::blocked_kernel<(unsigned)params_t::items_per_thread>
        <<<dim3(params_t::grid_dimensions), dim3(params_t::block_dimensions),
           params_t::shared_memory_bytes, context.stream()>>>(f, num_elements,
                                                              std::forward<args_t>(args)...);
    GRX_HIP_CHECK(hipGetLastError());
  }

  /// Cooperative launch of a __global__ function (grid = ceil(num_elements / block)).
  template <typename func_t, typename... args_t>
  void launch_cooperative(gcuda::standard_context_t& context, const func_t& f,
                          const std::size_t num_elements, args_t&&... args) {
    params_t::calculate_grid_dimensions_strided(num_elements);
    void* argument_ptrs[sizeof...(args_t) == 0 ? 1 : sizeof...(args_t)] = {
        const_cast<void*>(static_cast<const void*>(&args))...};
    GRX_HIP_CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(f),
                                             dim3(params_t::grid_dimensions),
                                             dim3(params_t::block_dimensions), argument_ptrs,
                                             (unsigned)params_t::shared_memory_bytes,
                                             context.stream()));
  }

  /// kernel<<<grid, block, smem, context.stream()>>>(args...) with the box's geometry.
  template <typename func_t, typename... args_t>
  void launch(gcuda::standard_context_t& context, const func_t& f, args_t&&... args) {
    f<<<dim3(params_t::grid_dimensions), dim3(params_t::block_dimensions),
        params_t::shared_memory_bytes, context.stream()>>>(std::forward<args_t>(args)...);
    GRX_HIP_CHECK(hipGetLastError());
  }
};

/// Resident wavefronts of `kernel` under the box's block size / dynamic LDS, as a fraction of
/// the CU's 32 wavefront slots (reference launch_box.hxx:337-360, warps of an SM there).
template <typename launch_box_type, typename func_t>
inline float occupancy(func_t kernel) {
  int blocks_per_cu = 0;
  GRX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(
      &blocks_per_cu, kernel, (int)launch_box_type::block_dimensions_t::size(),
      launch_box_type::shared_memory_bytes));
  const int waves_per_block =
      ((int)launch_box_type::block_dimensions_t::size() + gfx950::wavefront_size - 1) / gfx950::wavefront_size;
  constexpr int wave_slots_per_cu = 32;
  return (float)(blocks_per_cu * waves_per_block) / (float)wave_slots_per_cu;
}

}  // namespace launch_box
}  // namespace gcuda
}  // namespace gunrock
