/**
 * @file context.hxx
 * @brief Device context of the engine: one HIP stream + event timer + a
 * persistent operator workspace per GPU.
 *
 * Replaces reference include/gunrock/cuda/context.hxx:54-206 (standard_context_t,
 * multi_context_t) and include/gunrock/util/timer.hxx:16-49.  The namespace keeps
 * the name `gcuda` because the client headers spell it (algorithms/bfs.hxx:44,156).
 *
 * MI355X-first differences:
 *  - the context owns a workspace (device counters, pinned mirror, scratch) so an
 *    operator call performs no hipMalloc/hipFree (the reference allocates a
 *    1-element device_vector per advance: advance/block_mapped.hxx:200);
 *  - the timer records on the context's stream (the reference records on the
 *    null stream while work runs on a non-blocking stream);
 *  - multi-GPU is one process per GPU: a multi_context_t holds the local device's
 *    context plus the communicator_t {rank, world_size, collectives} of the RCCL job (the reference
 *    only holds a vector of local contexts and never uses more than the first:
 *    framework/enactor.hxx:243-254).
 */
#pragma once

#include <gunrock/hip/communicator.hxx>
#include <gunrock/hip/runtime.hxx>

#include <cstdio>
#include <cstdlib>

#include <thrust/execution_policy.h>
#include <thrust/system/hip/execution_policy.h>

namespace gunrock {

namespace util {

/// Event-pair timer on a stream; enact() returns its milliseconds.
struct timer_t {
  float time = 0.0f;

  explicit timer_t(hipStream_t stream = nullptr) : stream_(stream) {
    GRX_HIP_CHECK(hipEventCreate(&start_));
    GRX_HIP_CHECK(hipEventCreate(&stop_));
  }
  timer_t(const timer_t&) = delete;
  timer_t& operator=(const timer_t&) = delete;
  ~timer_t() {
    (void)hipEventDestroy(start_);
    (void)hipEventDestroy(stop_);
  }

  void begin() { GRX_HIP_CHECK(hipEventRecord(start_, stream_)); }
  void start() { begin(); }
  float end() {
    GRX_HIP_CHECK(hipEventRecord(stop_, stream_));
    GRX_HIP_CHECK(hipEventSynchronize(stop_));
    GRX_HIP_CHECK(hipEventElapsedTime(&time, start_, stop_));
    return time;
  }
  float stop() { return end(); }
  float seconds() const { return time * 1e-3f; }
  float milliseconds() const { return time; }
  void set_stream(hipStream_t s) { stream_ = s; }

 private:
  hipEvent_t start_{}, stop_{};
  hipStream_t stream_ = nullptr;
};

}  // namespace util

namespace gcuda {

using device_id_t = int;
using stream_t = hipStream_t;
using event_t = hipEvent_t;

/// Hardware constants of the one target (MI355X / gfx950, CDNA4).
struct gfx950 {
  static constexpr int wavefront_size = 64;
  static constexpr int compute_units = 256;
  static constexpr int xcds = 8;
  static constexpr int lds_bytes_per_cu = 160 * 1024;
};

/**
 * @brief Per-context operator workspace.  All operator state lives here (no
 * globals) so that independent contexts on independent host threads do not
 * interfere (reference operators::batch runs N run() calls on N threads).
 */
class workspace_t {
 public:
  static constexpr std::size_t n_counters = 32;

  /// 64-bit device counters (output cursor, hub-queue cursor, tile cursor, ...).
  unsigned long long* counters() {
    if (!counters_.data()) {
      counters_.reserve(n_counters);
      GRX_HIP_CHECK(hipMemset(counters_.data(), 0, n_counters * sizeof(unsigned long long)));
      // hipMemset on device memory may still be in flight when it returns, and the context's
      // stream is non-blocking (not ordered after the null stream): wait once, here
      GRX_HIP_CHECK(hipDeviceSynchronize());
    }
    return counters_.data();
  }
  /// Pinned host mirror of the counters.
  unsigned long long* mirror() { return mirror_.data(); }
  /// Pinned words a client's own kernels may leave facts of a run in (the host reads them after the
  /// run's final synchronisation: no hand-off kernel, no extra wait): [0] degree of the source,
  /// [1] vertices reached, [2] sum of their degrees, [3] set to 1 when [1] and [2] are valid.
  unsigned long long* run_facts() { return run_facts_.data(); }
  /// Slot of the mirror that carries the hand-off sequence number, and the next number.
  static constexpr std::size_t sequence_slot = 31;
  unsigned long long next_sequence() { return ++sequence_; }

  /// Growable untyped scratch (rocPRIM temp storage, block counts, flag words).
  /// Never null, also for a request of 0 bytes: rocPRIM reads a null temporary-storage pointer
  /// as a size query and then does NOTHING.
  void* scratch(std::size_t bytes) {
    if (bytes < 256)
      bytes = 256;
    if (bytes > scratch_.capacity())
      scratch_.reserve(bytes + bytes / 4);
    return scratch_.data();
  }
  /// A second, independent scratch region (e.g. the hub chunk queue).
  void* queue(std::size_t bytes) {
    if (bytes > queue_.capacity())
      queue_.reserve(bytes);
    return queue_.data();
  }
  std::size_t queue_capacity_bytes() const { return queue_.capacity(); }

  /// Per-graph facts an operator needs on the host (keyed by the offsets pointer).
  struct graph_facts_t {
    const void* key = nullptr;
    std::size_t vertices = 0;
    unsigned long long max_degree = 0;
    std::size_t edges = 0;
  };
  graph_facts_t* find_graph(const void* key, std::size_t vertices, std::size_t edges) {
    for (auto& g : graphs_)
      if (g.key == key && g.vertices == vertices && g.edges == edges)
        return &g;
    return nullptr;
  }
  /// Call when graph memory is released or rewritten: remembered facts are keyed by address.
  void forget_graphs() { graphs_.clear(); }
  graph_facts_t* remember_graph(const graph_facts_t& g) {
    graphs_.push_back(g);
    return &graphs_.back();
  }

  /// The edges of ONE graph ordered by destination (operators/by_destination.hxx), kept between
  /// operator calls.  Identified by the addresses AND a 64-bit fingerprint of the three CSR
  /// arrays' contents: memory released and reused for another graph never matches.
  struct by_destination_t {
    const void* offsets = nullptr;
    const void* indices = nullptr;
    const void* values = nullptr;
    std::size_t vertices = 0, edges = 0;
    unsigned long long fingerprint = 0;
    unsigned long long checked_for = 0;  ///< enactor (bsp.hxx unique_id) the fingerprint was last compared in
    unsigned long long current_run = 0;  ///< enactor whose advance is being dispatched (0: none)
    unsigned calls = 0;                  ///< whole-graph advances without an output seen on this graph
    bool built = false;
    bool refused = false;                ///< device memory did not allow the list: not tried again for this graph
    unsigned long long alternating_in = 0;  ///< enactor that keeps switching graphs (0: none) ...
    unsigned switches = 0;                  ///< ... and how often it did: no list for that enactor
    hip::buffer_t<unsigned char> items;  ///< [edges] {source, destination, edge, weight}
  };
  by_destination_t& by_destination() { return by_destination_; }

 private:
  hip::buffer_t<unsigned long long> counters_;
  hip::pinned_t<unsigned long long> mirror_{n_counters};
  hip::pinned_t<unsigned long long> run_facts_{8};
  unsigned long long sequence_ = 0;
  hip::buffer_t<unsigned char> scratch_;
  hip::buffer_t<unsigned char> queue_;
  std::vector<graph_facts_t> graphs_;
  by_destination_t by_destination_;
};

/// Run-time switches of the operators (per context).
struct operator_options_t {
  /// true: advance writes one output slot per traversed edge, invalid where the
  /// functor said no (the reference's layout, advance/block_mapped.hxx:142-145);
  /// false (default): only accepted neighbours are written, packed.
  bool holes_layout = false;
  /// Neighbour lists at least this long are cut into chunks spread over the GPU.
  unsigned hub_threshold = 256;
  /// Edges per chunk of such a list (one persistent workgroup step).
  unsigned chunk_edges = 1024;
  /// Persistent workgroups per CU for the tile / chunk kernels.
  unsigned tile_blocks_per_cu = 8;
  unsigned chunk_blocks_per_cu = 4;
  /// Input slots per tile of the block_mapped kernel (power of two <= 256); 0 = chosen per call.
  unsigned tile_width = 0;
  /// Hub chunks are taken by single wavefronts instead of whole workgroups.
  bool wave_chunks = false;
  /// Test hook: cap the hub chunk queue (0 = sized per call) to force the overflow path, in
  /// which a tile expands its hubs in place.
  unsigned long long chunk_queue_limit = 0;
  /// block_mapped: frontiers of at least this many slots take the fused form (hub pre-pass + one
  /// kernel for tiles and dynamically claimed hub chunks); 0 = never.
  std::size_t fused_min_slots = 32768;
  /// Persistent workgroups per CU of the fused kernel (what is resident at its register use).
  unsigned fused_blocks_per_cu = 6;
  /// Honour the settled-destination hint a client attached to its functor (operators/settled.hxx)
  /// on wide block_mapped levels; false: the functor is called for every edge.
  bool settled_filter = true;
  /// ... from this many edges of work on (a level of a few hubs is as wide as one of 1 M slots).
  unsigned long long settled_min_work = 1ull << 20;
  /// Clients that can name "what this level found" as a predicate of the vertex (BFS: depth ==
  /// level) run a level of at least this many edges of work WITHOUT an output frontier and build the
  /// next one by one pass over the labels (operators::filter::select_range: sorted runs, degree sum
  /// for free); 0 = never.  Below it the advance packs its output as before.  (RMAT-22, mean of 6
  /// sources, BFS + SSSP enact: 3.21 ms from 16 M, 3.15 from 8 M, 3.10-3.15 from 2 M, 1 M and 512 K.)
  unsigned long long label_scan_min_work = 2ull << 20;
  /// An advance WITHOUT an output frontier normally waits for its kernels like every operator.
  /// true (set by a client around ONE such call, when an operator that fetches the counters follows
  /// at once on the same stream -- operators::filter::select_range): it only enqueues; the next
  /// operator's hand-off reports its overflow flags and closes its kernel-time interval.
  bool defer_sync_of_none_output = false;
  /// A whole-graph advance WITHOUT an output (advance_io_type_t::graph -> none: `pr.hxx`'s push)
  /// on a graph of at least this many edges walks the edges grouped by DESTINATION from its second
  /// call on the same graph on (operators/by_destination.hxx); 0 = never (env GRX_BY_DESTINATION).
  unsigned long long by_destination_min_edges = 1ull << 20;
  /// Event-time the advance expansion kernels (two events per operator call).
  bool time_kernels = false;
};

/// Accumulated device time of the advance expansion kernels (when enabled).
struct kernel_clock_t {
  float total_ms = 0.0f;
  int launches = 0;
  bool pending = false;
  hipEvent_t begin_{}, end_{};
  bool created = false;
  void ensure() {
    if (!created) {
      GRX_HIP_CHECK(hipEventCreate(&begin_));
      GRX_HIP_CHECK(hipEventCreate(&end_));
      created = true;
    }
  }
  bool running = false;  // started, not yet stopped: a deferred operator left its interval open
  void start(hipStream_t s) {
    if (running)
      return;  // the interval of a deferred operator goes on through this one
    ensure();
    GRX_HIP_CHECK(hipEventRecord(begin_, s));
    running = true;
  }
  void stop(hipStream_t s) {
    GRX_HIP_CHECK(hipEventRecord(end_, s));
    pending = true;
    running = false;
  }
  /// Call after the stream has been synchronised.
  void collect() {
    if (!pending)
      return;
    float ms = 0;
    GRX_HIP_CHECK(hipEventSynchronize(end_));
    GRX_HIP_CHECK(hipEventElapsedTime(&ms, begin_, end_));
    total_ms += ms;
    ++launches;
    pending = false;
  }
  void reset() { total_ms = 0; launches = 0; pending = false; }
  ~kernel_clock_t() {
    if (created) {
      (void)hipEventDestroy(begin_);
      (void)hipEventDestroy(end_);
    }
  }
};

class standard_context_t {
 public:
  explicit standard_context_t(device_id_t device = 0) : ordinal_(device), owns_stream_(true) {
    GRX_HIP_CHECK(hipSetDevice(ordinal_));
    GRX_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    init();
  }
  standard_context_t(stream_t stream, device_id_t device = 0)
      : ordinal_(device), stream_(stream), owns_stream_(false) {
    GRX_HIP_CHECK(hipSetDevice(ordinal_));
    init();
  }
  standard_context_t(const standard_context_t&) = delete;
  standard_context_t& operator=(const standard_context_t&) = delete;
  ~standard_context_t() {
    timer_.reset();
    (void)hipEventDestroy(event_);
    if (owns_stream_)
      (void)hipStreamDestroy(stream_);
  }

  stream_t stream() { return stream_; }
  event_t event() { return event_; }
  device_id_t ordinal() const { return ordinal_; }
  util::timer_t& timer() { return *timer_; }
  workspace_t& workspace() { return workspace_; }
  operator_options_t& options() { return options_; }
  kernel_clock_t& kernel_clock() { return clock_; }
  const hipDeviceProp_t& props() const { return props_; }
  int compute_units() const { return props_.multiProcessorCount; }

  void synchronize() {
    GRX_HIP_CHECK(stream_ ? hipStreamSynchronize(stream_) : hipDeviceSynchronize());
  }

  /// Policy for the clients' own thrust calls (sssp.hxx:57, pr.hxx:66); the
  /// engine's operators do not use thrust.
  auto execution_policy() { return thrust::hip::par_nosync.on(stream_); }

  void print_properties() const {
    std::printf("%s: %d CUs, wavefront %d, %zu MiB L2, %.1f GiB\n", props_.name,
                props_.multiProcessorCount, props_.warpSize,
                (std::size_t)props_.l2CacheSize >> 20,
                (double)props_.totalGlobalMem / (1ull << 30));
  }

 private:
  void init() {
    // tuning knobs for experiments (defaults are the measured best)
    if (const char* e = std::getenv("GRX_TILE_BLOCKS_PER_CU"))
      options_.tile_blocks_per_cu = (unsigned)std::atoi(e);
    if (const char* e = std::getenv("GRX_CHUNK_BLOCKS_PER_CU"))
      options_.chunk_blocks_per_cu = (unsigned)std::atoi(e);
    if (const char* e = std::getenv("GRX_TILE_WIDTH"))
      options_.tile_width = (unsigned)std::atoi(e);
    if (const char* e = std::getenv("GRX_WAVE_CHUNKS"))
      options_.wave_chunks = std::atoi(e) != 0;
    if (const char* e = std::getenv("GRX_FUSED_MIN_SLOTS"))
      options_.fused_min_slots = (std::size_t)std::atoll(e);
    if (const char* e = std::getenv("GRX_FUSED_BLOCKS_PER_CU"))
      options_.fused_blocks_per_cu = (unsigned)std::atoi(e);
    if (const char* e = std::getenv("GRX_SETTLED_FILTER"))
      options_.settled_filter = std::atoi(e) != 0;
    if (const char* e = std::getenv("GRX_SETTLED_MIN_WORK"))
      options_.settled_min_work = (unsigned long long)std::atoll(e);
    if (const char* e = std::getenv("GRX_LABEL_SCAN_MIN_WORK"))
      options_.label_scan_min_work = (unsigned long long)std::atoll(e);
    if (const char* e = std::getenv("GRX_BY_DESTINATION"))
      options_.by_destination_min_edges = (unsigned long long)std::atoll(e);  // 0: never, 1: always
    if (const char* e = std::getenv("GRX_CHUNK_QUEUE_LIMIT"))
      options_.chunk_queue_limit = (unsigned long long)std::atoll(e);
    GRX_HIP_CHECK(hipEventCreateWithFlags(&event_, hipEventDisableTiming));
    GRX_HIP_CHECK(hipGetDeviceProperties(&props_, ordinal_));
    timer_ = std::make_unique<util::timer_t>(stream_);
  }

  device_id_t ordinal_ = 0;
  stream_t stream_ = nullptr;
  bool owns_stream_ = false;
  event_t event_{};
  hipDeviceProp_t props_{};
  std::unique_ptr<util::timer_t> timer_;
  workspace_t workspace_;
  operator_options_t options_;
  kernel_clock_t clock_;
};

/**
 * @brief The context handed to problem_t / enactor_t / operators.  size() is the
 * number of LOCAL device contexts (1 in the process-per-GPU model); world_size()
 * is the number of ranks of the job the context is attached to.
 */
class multi_context_t {
 public:
  static constexpr std::size_t MAX_NUMBER_OF_GPUS = 1024;

  std::vector<standard_context_t*> contexts;
  std::vector<device_id_t> devices;

  explicit multi_context_t(device_id_t device) : devices(1, device) {
    contexts.push_back(new standard_context_t(device));
  }
  multi_context_t(device_id_t device, stream_t stream) : devices(1, device) {
    contexts.push_back(new standard_context_t(stream, device));
  }
  explicit multi_context_t(std::vector<device_id_t> _devices) : devices(std::move(_devices)) {
    for (auto d : devices)
      contexts.push_back(new standard_context_t(d));
    if (!devices.empty())
      GRX_HIP_CHECK(hipSetDevice(devices[0]));
  }
  multi_context_t(const multi_context_t&) = delete;
  multi_context_t& operator=(const multi_context_t&) = delete;
  ~multi_context_t() {
    for (auto* c : contexts)
      delete c;
  }

  standard_context_t* get_context(device_id_t i) { return contexts[(std::size_t)i]; }
  std::size_t size() const { return contexts.size(); }

  /// Peer access between the local devices (reference cuda/context.hxx:188-205).
  void enable_peer_access() {
    int n = (int)size();
    for (int i = 0; i < n; ++i) {
      GRX_HIP_CHECK(hipSetDevice(contexts[i]->ordinal()));
      for (int j = 0; j < n; ++j) {
        if (i == j)
          continue;
        hipError_t st = hipDeviceEnablePeerAccess(contexts[j]->ordinal(), 0);
        if (st != hipSuccess && st != hipErrorPeerAccessAlreadyEnabled)
          error::throw_if_exception(st, "hipDeviceEnablePeerAccess");
      }
    }
    if (n)
      GRX_HIP_CHECK(hipSetDevice(contexts[0]->ordinal()));
  }

  // --- process-per-GPU job attachment (RCCL over xGMI) ----------------------
  /// Join a job of `world_size` ranks (one process per GPU).  `table` is the collective transport:
  /// rccl::make(rank, world, id, device) in production, host callbacks in the test rigs.  A rank
  /// owns the contiguous vertex range it is given with set_owned_rows(); the graph it traverses is
  /// its slice of the global graph (graph::build::partition / grx_graph_partition: global vertex
  /// ids, rows outside the range empty).  With a job attached, enactor_t::enact() exchanges the
  /// frontiers between supersteps (framework/partitioned.hxx).
  void attach_job(int rank, int world_size, collective_table_t table) {
    comm_.attach(rank, world_size, table);
  }
  void detach_job() { comm_.detach(); }
  int rank() const { return comm_.rank(); }
  int world_size() const { return comm_.world_size(); }
  communicator_t& communicator() { return comm_; }

  /// Vertex range [begin, end) this rank owns (the whole graph when no job is attached).
  void set_owned_rows(long long begin, long long end) {
    owned_begin_ = begin;
    owned_end_ = end;
  }
  long long owned_begin() const { return owned_begin_; }
  long long owned_end() const { return owned_end_; }
  bool owns(long long v) const { return owned_end_ < 0 || (v >= owned_begin_ && v < owned_end_); }

 private:
  communicator_t comm_;
  long long owned_begin_ = 0;
  long long owned_end_ = -1;  // -1: everything
};

}  // namespace gcuda
}  // namespace gunrock
