/**
 * @file frontier.hxx
 * @brief The vertex / edge frontier: a device array of ids plus a host-side
 * element count, double-buffered by the enactor.
 *
 * Surface of reference framework/frontier/frontier.hxx:33-148 and
 * framework/frontier/vector_frontier.hxx:28-256 (push_back, reserve, resize,
 * data/begin/end, get/set_number_of_elements, get_capacity, fill, sequence, sort,
 * is_empty, get/set_element_at, print).  Elements equal to
 * numeric_limits<type_t>::invalid() are holes every operator skips.
 *
 * Own design: storage is a shared hip::buffer_t (plain hipMalloc), the object is
 * a small handle that kernels never receive -- kernels take (pointer, count).
 * fill / sequence / sort are hand-written kernels and rocPRIM, not thrust.
 */
#pragma once

#include <cstdio>
#include <type_traits>

#include <gunrock/hip/primitives.hxx>
#include <gunrock/hip/runtime.hxx>
#include <gunrock/util/math.hxx>
#include <gunrock/util/type_limits.hxx>

namespace gunrock {

namespace sort {
using order_t = hip::sort_order_t;
}

namespace frontier {

enum frontier_view_t { vector, bitmap, boolmap };
enum frontier_kind_t { vertex_frontier, edge_frontier, vertex_edge_frontier };

namespace detail {

template <typename T>
__global__ void fill_kernel(T* p, std::size_t n, T value) {
  for (std::size_t i = blockIdx.x * (std::size_t)blockDim.x + threadIdx.x; i < n;
       i += (std::size_t)gridDim.x * blockDim.x)
    p[i] = value;
}

template <typename T>
__global__ void sequence_kernel(T* p, std::size_t n, T first) {
  for (std::size_t i = blockIdx.x * (std::size_t)blockDim.x + threadIdx.x; i < n;
       i += (std::size_t)gridDim.x * blockDim.x)
    p[i] = first + static_cast<T>(i);
}

inline unsigned grid_for(std::size_t n, unsigned block = 256, unsigned cap = 2048) {
  std::size_t g = (n + block - 1) / block;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace detail

template <typename vertex_t,
          typename edge_t,
          frontier_kind_t _kind = frontier_kind_t::vertex_frontier,
          frontier_view_t _view = frontier_view_t::vector>
class frontier_t {
 public:
  using vertex_type = vertex_t;
  using edge_type = edge_t;
  using type_t = std::conditional_t<_kind == frontier_kind_t::vertex_frontier, vertex_t, edge_t>;
  using offset_t = std::conditional_t<_kind == frontier_kind_t::vertex_frontier, edge_t, vertex_t>;
  using frontier_type = frontier_t<vertex_t, edge_t, _kind, _view>;

  frontier_t() : storage_(std::make_shared<hip::buffer_t<type_t>>()) {}

  explicit frontier_t(std::size_t size, float frontier_resizing_factor = 1.0f)
      : storage_(std::make_shared<hip::buffer_t<type_t>>(size)),
        num_elements_(size),
        resizing_factor_(frontier_resizing_factor) {}

  /// Frontier over caller-owned device memory holding `size` elements (cannot grow).
  static frontier_t wrap(type_t* external, std::size_t size, std::size_t capacity) {
    frontier_t f;
    f.storage_ = std::make_shared<hip::buffer_t<type_t>>(external, capacity);
    f.num_elements_ = size;
    return f;
  }

  static constexpr frontier_kind_t get_kind() { return _kind; }
  static constexpr frontier_view_t get_view() { return _view; }

  // --- size bookkeeping -----------------------------------------------------
  std::size_t get_number_of_elements(hipStream_t = nullptr) const { return num_elements_; }
  /// Changing the length from outside invalidates the work hint (see work_hint()).
  void set_number_of_elements(std::size_t const& n) {
    num_elements_ = n;
    work_hint_ = unknown_work;
    ascending_ = false;
  }

  /// The elements ascend (in long runs): operators::filter::select_range wrote them.  On a hot-first
  /// numbered graph ascending ids are descending degrees, so 64 CONSECUTIVE elements make a tile of
  /// near-equal, possibly all heavy, rows; the wide-level advance then deals its tiles across the
  /// frontier instead (lane l of tile t takes element l * tiles + t).  Any other writer clears it.
  bool ascending() const { return ascending_; }
  void set_ascending(bool a) { ascending_ = a; }

  /**
   * @brief Upper bound of the sum of the degrees of the valid elements, when an
   * operator of this engine produced the contents (advance sums the degrees of
   * what it emits; filters and uniquify only remove elements, so they pass the
   * bound on).  unknown_work otherwise.  Lets the next advance size its output
   * without the reference's reduction pass (advance/helpers.hxx:112-146).  Code
   * that rewrites elements in place through data() must call invalidate_work_hint().
   */
  static constexpr unsigned long long unknown_work = ~0ull;
  unsigned long long work_hint() const { return work_hint_; }
  void set_work_hint(unsigned long long w) { work_hint_ = w; }
  void invalidate_work_hint() { work_hint_ = unknown_work; }
  bool is_empty() const { return num_elements_ == 0; }
  std::size_t get_capacity() const { return storage_->capacity(); }
  float get_resizing_factor() const { return resizing_factor_; }
  void set_resizing_factor(float f) { resizing_factor_ = f; }

  // --- storage --------------------------------------------------------------
  type_t* data() const { return storage_->data(); }
  type_t* get() const { return storage_->data(); }
  type_t* begin() const { return data(); }
  type_t* end() const { return data() + num_elements_; }

  /// Capacity for at least size * resizing_factor elements; contents are kept.
  void reserve(std::size_t const& size) {
    std::size_t want = (std::size_t)((double)size * (double)resizing_factor_);
    if (want < size)
      want = size;
    storage_->reserve(want, num_elements_);
  }

  void resize(std::size_t const& size,
              type_t const default_value = gunrock::numeric_limits<type_t>::invalid()) {
    if (size > get_capacity())
      storage_->reserve(size, num_elements_);
    if (size > num_elements_) {
      std::size_t extra = size - num_elements_;
      detail::fill_kernel<<<detail::grid_for(extra), 256>>>(data() + num_elements_, extra,
                                                            default_value);
      GRX_HIP_CHECK(hipDeviceSynchronize());
    }
    num_elements_ = size;
    work_hint_ = unknown_work;
  }

  /// Host-side append of one element (used by prepare_frontier, bfs.hxx:77).
  void push_back(type_t const& value) {
    if (num_elements_ + 1 > get_capacity())
      storage_->reserve(get_capacity() ? 2 * get_capacity() : 64, num_elements_);
    GRX_HIP_CHECK(hipMemcpy(data() + num_elements_, &value, sizeof(type_t), hipMemcpyHostToDevice));
    // a small pageable H2D copy may return before it lands; operators run on a non-blocking
    // stream that is not ordered after the null stream
    GRX_HIP_CHECK(hipStreamSynchronize(nullptr));
    ++num_elements_;
    work_hint_ = unknown_work;
  }

  /// Device-side append of one element, ENQUEUED on `stream` (no host copy, nothing awaited):
  /// for callers whose next operator runs on that stream anyway.
  void push_back(type_t const& value, hipStream_t stream) {
    if (num_elements_ + 1 > get_capacity())
      storage_->reserve(get_capacity() ? 2 * get_capacity() : 64, num_elements_, stream);
    detail::fill_kernel<<<1, 64, 0, stream>>>(data() + num_elements_, std::size_t(1), value);
    GRX_HIP_CHECK(hipGetLastError());
    ++num_elements_;
    work_hint_ = unknown_work;
  }

  void fill(type_t const value, hipStream_t stream = nullptr) {
    work_hint_ = unknown_work;
    if (!num_elements_)
      return;
    detail::fill_kernel<<<detail::grid_for(num_elements_), 256, 0, stream>>>(data(), num_elements_,
                                                                             value);
    GRX_HIP_CHECK(hipGetLastError());
  }

  void sequence(type_t const initial_value, std::size_t const& size, hipStream_t stream = nullptr) {
    if (get_capacity() < size)
      reserve(size);
    num_elements_ = size;
    work_hint_ = unknown_work;
    if (!size)
      return;
    detail::sequence_kernel<<<detail::grid_for(size), 256, 0, stream>>>(data(), size,
                                                                        initial_value);
    GRX_HIP_CHECK(hipGetLastError());
  }

  /// Radix sort of the live elements (rocPRIM).  Allocates its own temporaries;
  /// the uniquify operator uses the context workspace instead.
  void sort(sort::order_t order = sort::order_t::ascending, hipStream_t stream = nullptr) {
    if (num_elements_ < 2)
      return;
    std::size_t bytes = hip::radix_sort_temp_bytes<type_t>(num_elements_);
    // never a null temp: rocPRIM reads "temp == nullptr" as a size query and sorts NOTHING, and the
    // copy below would then publish the uninitialised `sorted` buffer (round-2 fault)
    hip::buffer_t<unsigned char> temp(bytes < 256 ? 256 : bytes);
    hip::buffer_t<type_t> sorted(num_elements_);
    hip::radix_sort_keys(temp.data(), bytes, data(), sorted.data(), num_elements_, order, stream);
    GRX_HIP_CHECK(hipMemcpyAsync(data(), sorted.data(), num_elements_ * sizeof(type_t),
                                 hipMemcpyDeviceToDevice, stream));
    GRX_HIP_CHECK(hipStreamSynchronize(stream));
  }

  /// Swap storage handles with another frontier (O(1)).
  void swap(frontier_t& other) {
    std::swap(storage_, other.storage_);
    std::swap(num_elements_, other.num_elements_);
    std::swap(resizing_factor_, other.resizing_factor_);
    std::swap(work_hint_, other.work_hint_);
    std::swap(ascending_, other.ascending_);
  }

  std::vector<type_t> to_host() const {
    std::vector<type_t> h(num_elements_);
    if (num_elements_)
      GRX_HIP_CHECK(hipMemcpy(h.data(), data(), num_elements_ * sizeof(type_t), hipMemcpyDeviceToHost));
    return h;
  }

  void print() const {
    auto h = to_host();
    std::printf("Frontier = ");
    for (auto x : h)
      std::printf("%lld ", (long long)x);
    std::printf("\n");
  }

 private:
  std::shared_ptr<hip::buffer_t<type_t>> storage_;
  std::size_t num_elements_ = 0;
  float resizing_factor_ = 1.0f;
  unsigned long long work_hint_ = ~0ull;
  bool ascending_ = false;
};

}  // namespace frontier
}  // namespace gunrock
