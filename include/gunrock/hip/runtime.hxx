/**
 * @file runtime.hxx
 * @brief HIP runtime layer of the MI355X-native frontier engine: error
 * convention, memory spaces, owning device / pinned-host buffers.
 *
 * Replaces (own implementation, gfx950 only): reference include/gunrock/error.hxx:21-46
 * (exception_t / throw_if_exception), include/gunrock/memory.hxx:33-122
 * (memory_space_t, allocate/free/raw_pointer_cast) and the thrust::device_vector
 * storage behind container/vector.hxx:26-31.  No thrust, no CUB: storage is plain
 * hipMalloc / hipHostMalloc so nothing on the operator path allocates implicitly.
 */
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace gunrock {

namespace error {

using error_t = hipError_t;

/// Exceptions are the error convention of the C++ surface (reference error.hxx:21-46).
struct exception_t : std::exception {
  std::string report;
  explicit exception_t(error_t status, std::string message = "")
      : report(std::string(hipGetErrorString(status)) + "\t: " + message) {}
  explicit exception_t(std::string message = "") : report(std::move(message)) {}
  const char* what() const noexcept override { return report.c_str(); }
};

inline void throw_if_exception(error_t status, std::string message = "") {
  if (status != hipSuccess)
    throw exception_t(status, std::move(message));
}

inline void throw_if_exception(bool is_exception, std::string message = "") {
  if (is_exception)
    throw exception_t(std::move(message));
}

}  // namespace error

/// Every HIP call on the engine's path is checked.
#define GRX_HIP_CHECK(expr) \
  ::gunrock::error::throw_if_exception((expr), #expr " @ " __FILE__ ":" + std::to_string(__LINE__))

namespace memory {

enum memory_space_t { device, host };

namespace detail {
/// Hook the device block cache (hip::block_cache_t, below) installs: frees everything it parks.
inline void (*&trim_hook())() {
  static void (*hook)() = nullptr;
  return hook;
}
}  // namespace detail

template <typename type_t>
inline type_t* allocate(std::size_t bytes, memory_space_t space = memory_space_t::device) {
  void* p = nullptr;
  if (bytes) {
    if (space == memory_space_t::device) {
      hipError_t st = hipMalloc(&p, bytes);
      if (st == hipErrorOutOfMemory && detail::trim_hook()) {
        // blocks parked for reuse are the engine's to give back before anybody sees an OOM
        (void)hipGetLastError();
        detail::trim_hook()();
        st = hipMalloc(&p, bytes);
      }
      GRX_HIP_CHECK(st);
    } else {
      GRX_HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    }
  }
  return reinterpret_cast<type_t*>(p);
}

template <typename type_t>
inline void free(type_t* p, memory_space_t space = memory_space_t::device) {
  if (!p)
    return;
  if (space == memory_space_t::device)
    (void)hipFree(p);
  else
    (void)hipHostFree(p);
}

template <typename type_t>
__host__ __device__ inline type_t* raw_pointer_cast(type_t* p) {
  return p;
}

}  // namespace memory

using memory::memory_space_t;  // spelt unqualified by harnesses and unit tests (unittests/io/smtx.cuh:26)

namespace hip {

/**
 * @brief Process-wide cache of large device blocks.  An enactor reserves two frontiers of
 * 1.5 * max(|E|,|V|) elements (reference framework/enactor.hxx:181-192) and frees them when the
 * run ends; hipFree costs ~0.2 ms and synchronises the device, which is 20 % of an RMAT-22 BFS.
 * Blocks of >= 1 MiB are therefore parked here, keyed by (device, exact size), and handed back to
 * the next run ON THAT DEVICE.  At most `limit_bytes` are parked; beyond that blocks are really
 * freed; memory::allocate trims the cache and retries before it reports an out-of-memory.
 * Thread-safe.
 */
class block_cache_t {
 public:
  static block_cache_t& instance() {
    static block_cache_t cache;
    return cache;
  }
  /// A parked block of exactly `bytes` that lives on the CURRENT device, or nullptr.
  void* take(std::size_t bytes) {
    if (bytes < min_bytes)
      return nullptr;
    const int device = current_device();
    std::lock_guard<std::mutex> lock(mutex_);
    for (std::size_t i = 0; i < blocks_.size(); ++i) {
      if (blocks_[i].bytes == bytes && blocks_[i].device == device) {
        void* p = blocks_[i].ptr;
        blocks_[i] = blocks_.back();
        blocks_.pop_back();
        parked_ -= bytes;
        return p;
      }
    }
    return nullptr;
  }
  /// Returns true when the block was parked (caller must not free it).  `device` = the device the
  /// block was allocated on (a block is only ever handed back to an allocation on that device).
  bool give(void* p, std::size_t bytes, int device) {
    if (bytes < min_bytes)
      return false;
    std::lock_guard<std::mutex> lock(mutex_);
    if (parked_ + bytes > limit_bytes)
      return false;
    blocks_.push_back(block_t{p, bytes, device});
    parked_ += bytes;
    return true;
  }
  /// Really free everything parked (all devices).  Also what memory::allocate calls before it
  /// reports an out-of-memory, and what grx_trim_cache() exports for hosts that share the device
  /// with another allocator (torch).
  void trim() {
    std::lock_guard<std::mutex> lock(mutex_);
    for (auto& b : blocks_)
      (void)hipFree(b.ptr);
    blocks_.clear();
    parked_ = 0;
  }
  std::size_t parked_bytes() {
    std::lock_guard<std::mutex> lock(mutex_);
    return parked_;
  }
  static int current_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
  }
  static constexpr std::size_t min_bytes = 1ull << 20;
  static constexpr std::size_t limit_bytes = 64ull << 30;  // 64 GiB of a 288 GB part (an RMAT-26
                                                           // enactor holds 2 x 12.9 GB of frontiers)

 private:
  block_cache_t() {
    memory::detail::trim_hook() = [] { block_cache_t::instance().trim(); };
  }
  struct block_t {
    void* ptr;
    std::size_t bytes;
    int device;
  };
  std::mutex mutex_;
  std::vector<block_t> blocks_;
  std::size_t parked_ = 0;
};

/**
 * @brief Owning, growable device array.  Growth preserves contents (needed by
 * frontier_t::push_back / reserve); the engine sizes buffers up front so that no
 * allocation happens inside the BSP loop in the steady state.
 */
template <typename type_t>
class buffer_t {
 public:
  buffer_t() = default;
  explicit buffer_t(std::size_t n) { reserve(n); }
  /// Non-owning view of caller memory: never freed, cannot grow.
  buffer_t(type_t* external, std::size_t capacity) : ptr_(external), cap_(capacity), owns_(false) {}
  buffer_t(const buffer_t&) = delete;
  buffer_t& operator=(const buffer_t&) = delete;
  buffer_t(buffer_t&& o) noexcept
      : ptr_(o.ptr_), cap_(o.cap_), owns_(o.owns_), park_(o.park_), device_(o.device_) {
    o.ptr_ = nullptr; o.cap_ = 0;
  }
  buffer_t& operator=(buffer_t&& o) noexcept {
    if (this != &o) {
      release();
      ptr_ = o.ptr_; cap_ = o.cap_; owns_ = o.owns_; park_ = o.park_; device_ = o.device_;
      o.ptr_ = nullptr; o.cap_ = 0;
    }
    return *this;
  }
  /// false: storage goes straight back to the device when released (long-lived, odd-sized
  /// arrays such as a graph's CSR would only clog the reuse cache).
  void set_parking(bool park) { park_ = park; }
  ~buffer_t() { release(); }

  type_t* data() const { return ptr_; }
  std::size_t capacity() const { return cap_; }

  /// Grow to at least n elements, keeping the first `keep` elements.
  void reserve(std::size_t n, std::size_t keep = 0, hipStream_t stream = nullptr) {
    if (n <= cap_)
      return;
    error::throw_if_exception(!owns_, "caller-provided frontier storage is too small: need " +
                                          std::to_string(n) + " elements, have " +
                                          std::to_string(cap_));
    type_t* fresh = park_ ? reinterpret_cast<type_t*>(block_cache_t::instance().take(n * sizeof(type_t)))
                          : nullptr;
    if (!fresh)
      fresh = memory::allocate<type_t>(n * sizeof(type_t));
    const int device = block_cache_t::current_device();
    if (ptr_ && keep) {
      GRX_HIP_CHECK(hipMemcpyAsync(fresh, ptr_, keep * sizeof(type_t), hipMemcpyDeviceToDevice, stream));
      GRX_HIP_CHECK(hipStreamSynchronize(stream));
    }
    release();
    ptr_ = fresh;
    cap_ = n;
    device_ = device;
  }

  void release() {
    if (ptr_ && owns_ &&
        !(park_ && block_cache_t::instance().give(ptr_, cap_ * sizeof(type_t), device_)))
      (void)hipFree(ptr_);
    ptr_ = nullptr;
    cap_ = 0;
    owns_ = true;
  }

 private:
  type_t* ptr_ = nullptr;
  std::size_t cap_ = 0;
  bool owns_ = true;
  bool park_ = true;
  int device_ = 0;  // device the storage was allocated on
};

/**
 * @brief Sized device array (the engine's stand-in for a device vector where a
 * size matters: the enactor's scan workspace, owning CSR arrays).
 */
template <typename type_t>
class device_array_t {
 public:
  device_array_t() = default;
  explicit device_array_t(std::size_t n) { resize(n); }
  device_array_t(const type_t* host, std::size_t n) { assign(host, n); }

  std::size_t size() const { return size_; }
  std::size_t capacity() const { return buf_.capacity(); }
  type_t* data() const { return buf_.data(); }
  bool empty() const { return size_ == 0; }

  void resize(std::size_t n) {
    if (n > buf_.capacity())
      buf_.reserve(n, size_);
    size_ = n;
  }
  void assign(const type_t* host, std::size_t n) {
    resize(n);
    if (n) {
      GRX_HIP_CHECK(hipMemcpy(buf_.data(), host, n * sizeof(type_t), hipMemcpyHostToDevice));
      GRX_HIP_CHECK(hipStreamSynchronize(nullptr));  // see frontier_t::push_back
    }
  }
  void assign(const std::vector<type_t>& host) { assign(host.data(), host.size()); }
  std::vector<type_t> to_host() const {
    std::vector<type_t> h(size_);
    if (size_)
      GRX_HIP_CHECK(hipMemcpy(h.data(), buf_.data(), size_ * sizeof(type_t), hipMemcpyDeviceToHost));
    return h;
  }
  /// With a stream: enqueued there.  Without one: complete when the call returns -- a memset on
  /// the null stream is NOT ordered before work on the engine's non-blocking streams, and it may
  /// still be in flight when hipMemsetAsync returns.
  void zero(hipStream_t stream = nullptr) {
    if (!size_)
      return;
    GRX_HIP_CHECK(hipMemsetAsync(buf_.data(), 0, size_ * sizeof(type_t), stream));
    if (!stream)
      GRX_HIP_CHECK(hipStreamSynchronize(nullptr));
  }
  void set_parking(bool park) { buf_.set_parking(park); }

 private:
  buffer_t<type_t> buf_;
  std::size_t size_ = 0;
};

/// Pinned host words the device copies counters into (one D2H per operator).
template <typename type_t>
class pinned_t {
 public:
  explicit pinned_t(std::size_t n = 1) : n_(n) {
    ptr_ = memory::allocate<type_t>(n * sizeof(type_t), memory::memory_space_t::host);
    std::memset(ptr_, 0, n * sizeof(type_t));
  }
  pinned_t(const pinned_t&) = delete;
  pinned_t& operator=(const pinned_t&) = delete;
  ~pinned_t() { if (ptr_) (void)hipHostFree(ptr_); }
  type_t* data() const { return ptr_; }
  type_t& operator[](std::size_t i) const { return ptr_[i]; }
  std::size_t size() const { return n_; }

 private:
  type_t* ptr_ = nullptr;
  std::size_t n_ = 0;
};

}  // namespace hip
}  // namespace gunrock
