/**
 * @file partitioned_run.hip
 * @brief C ABI: a context's membership in a multi-GPU job (RCCL or host callbacks) and the
 * vertex-partitioned traversals as ONE call each -- the BSP superstep loop in C++ with the
 * collectives issued directly on the context's stream (gcuda::communicator_t, hip/communicator.hxx).
 *
 * No reference counterpart (its operators throw for more than one context,
 * framework/operators/advance/advance.hxx:125-128; enactor.hxx:243-254 uses context 0 only);
 * design per SURVEY.md 8(e).  The device side of a superstep is partitioned.hip's
 * grx_partitioned_step (admit -> advance -> pack, enqueue-only); this file owns the buffers, the
 * loop, the choice of exchange format per superstep and the termination test.
 */
#include "capi_internal.hxx"

#include <algorithm>
#include <chrono>
#include <vector>

using namespace essentials_amd;

struct grx_partitioned_s {
  grx_context_t ctx = nullptr;
  grx_graph_t local = nullptr;
  int32_t lo = 0, hi = 0;
  int64_t n = 0;
  grx_options opts{};
  int64_t slot0 = 0;              // int64 words per rank in a superstep's first all-gather
  int64_t dense_threshold = 0;    // BFS: level bitmaps above this many finds on the busiest rank
  int64_t replica_threshold = 0;  // SSSP: replica all-reduce above it
  int64_t words = 0;              // ceil(V / 64)
  hip::device_array_t<int32_t> stamp, sent, frontier, scratch;
  hip::device_array_t<int64_t> send, recv, recv_big, bits, recv_bits;
  hip::device_array_t<unsigned long long> fcount;
  hip::device_array_t<float> snapshot;
  // plan over a slice of a renumbered copy (grx_graph_partition_hot_first): the replica the
  // supersteps run on; the caller's array is written once, at the end of a run
  hip::device_array_t<int32_t> renumbered_labels;
  // PageRank
  hip::device_array_t<float> scale, partial;
  // gathered per-rank counts land here (pinned): [0, world) counts, [world] sequence number
  std::unique_ptr<hip::pinned_t<unsigned long long>> heads;
  unsigned long long head_sequence = 0;
};

namespace {

/// Everything a run starts from, in one launch: labels <- unreached (source <- 0), both stamp
/// arrays <- -1, the owner of the source gets a frontier of one.
template <typename label_t>
__global__ void __launch_bounds__(256)
    reset_run_kernel(label_t* labels, int32_t* stamp, int32_t* sent, int64_t n, label_t unreached,
                     int32_t source, int32_t lo, int32_t hi, int32_t* frontier,
                     unsigned long long* fcount) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    labels[i] = i == source ? label_t(0) : unreached;
    stamp[i] = -1;
    sent[i] = -1;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const bool mine = source >= lo && source < hi;
    if (mine)
      frontier[0] = source;
    *fcount = mine ? 1ull : 0ull;
  }
}

/// End of a run on a renumbered slice: out[vertex_of[r]] = labels[r] (4-byte labels of either kind).
__global__ void __launch_bounds__(256)
    deliver_kernel(const int32_t* labels, const int32_t* vertex_of, int64_t n, int32_t* out) {
  for (int64_t r = blockIdx.x * 256ll + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256)
    out[vertex_of[r]] = labels[r];
}

/// Hand the gathered per-rank find counts (word 0 of every rank's slot) to the host through pinned
/// memory, then stamp the sequence number the host spins on: no memcpy command, no stream
/// synchronisation call on the superstep's critical path.
__global__ void publish_heads_kernel(const int64_t* recv, int64_t slot, int32_t world,
                                     unsigned long long* pinned, unsigned long long sequence) {
  const int p = threadIdx.x;
  if (p < world)
    pinned[p] = (unsigned long long)__hip_atomic_load(recv + (int64_t)p * slot, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
  __threadfence_system();
  __syncthreads();
  if (p == 0)
    __hip_atomic_store(&pinned[world], sequence, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

/// p <- partial + base; *error <- max |p_new - p_old| (float bits: non-negative floats order as
/// unsigned integers).
__global__ void __launch_bounds__(256)
    pagerank_update_kernel(float* p, const float* partial, int64_t n, float one_minus_alpha,
                           unsigned long long* error_bits) {
  __shared__ float s_part[256 / hip::wave_size];
  const float base = (one_minus_alpha + partial[n]) / (float)n;
  float worst = 0.0f;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float fresh = partial[i] + base;
    const float d = fresh - p[i];
    worst = fmaxf(worst, d < 0 ? -d : d);
    p[i] = fresh;
  }
  worst = hip::wave_max(worst);
  if ((threadIdx.x & 63) == 0)
    s_part[threadIdx.x / hip::wave_size] = worst;
  __syncthreads();
  if (threadIdx.x == 0) {
    float w = 0;
    for (int k = 0; k < 256 / hip::wave_size; ++k)
      w = fmaxf(w, s_part[k]);
    atomicMax(error_bits, (unsigned long long)__float_as_uint(w));
  }
}

/// The gathered counts did not arrive in time: a peer is gone or stuck inside the collective.
struct superstep_timeout_t : std::runtime_error {
  using std::runtime_error::runtime_error;
};

long long superstep_timeout_ms() {
  if (const char* e = std::getenv("GRX_PARTITIONED_TIMEOUT_MS"))
    return std::atoll(e);
  return 30000;
}

/// A rank whose own step failed still takes part in the superstep's gather, with this in its
/// slot's header word: every rank sees it in the same gather and returns an error.
constexpr unsigned long long FAILED_HEAD = ~0ull;

__global__ void mark_failed_kernel(int64_t* send) { send[0] = -1; }

/// The ONE host wait of a superstep: spins on the sequence word the publish kernel stamps behind
/// the gathered counts.  Bounded: a collective whose peer never arrives keeps the stream "not
/// ready" for ever, so the wall clock decides.
unsigned long long* await_heads(grx_partitioned_s& p, gcuda::standard_context_t& sc) {
  volatile unsigned long long* flag = p.heads->data() + (std::size_t)p.ctx->mc->world_size();
  unsigned spins = 0;
  const long long limit_ms = superstep_timeout_ms();
  const auto started = std::chrono::steady_clock::now();
  while (*flag < p.head_sequence) {
    __builtin_ia32_pause();
    if ((++spins & 0xFFFFu) == 0) {
      hipError_t st = hipStreamQuery(sc.stream());
      if (st != hipSuccess && st != hipErrorNotReady)
        error::throw_if_exception(st, "partitioned run: a superstep's kernels or collective failed");
      if (limit_ms > 0 &&
          std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - started)
                  .count() > limit_ms)
        throw superstep_timeout_t("partitioned run: the gathered counts of a superstep did not arrive "
                                  "within " + std::to_string(limit_ms) + " ms (a peer rank is gone or stuck)");
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return p.heads->data();
}

template <typename label_t>
int run_as(grx_partitioned_s& p, int32_t edge_op, int32_t source, label_t* caller_labels, label_t unreached,
           grx_partitioned_stats* stats) {
  auto& mc = *p.ctx->mc;
  auto& sc = p.ctx->single();
  auto& comm = mc.communicator();
  const int world = comm.world_size(), rank = comm.rank();
  hipStream_t stream = sc.stream();
  const unsigned grid = (unsigned)sc.compute_units() * 8;
  // a slice of a renumbered copy: source in, labels out in the CALLER's numbering
  static_assert(sizeof(label_t) == 4, "replicas hold 4-byte labels");
  label_t* labels = caller_labels;
  if (p.local->renumbered_slice) {
    source = p.local->hot_rank_of[(std::size_t)source];
    labels = reinterpret_cast<label_t*>(p.renumbered_labels.data());
  }

  reset_run_kernel<label_t><<<grid, 256, 0, stream>>>(labels, p.stamp.data(), p.sent.data(), p.n,
                                                      unreached, source, p.lo, p.hi,
                                                      p.frontier.data(), p.fcount.data());
  GRX_HIP_CHECK(hipGetLastError());
  sc.synchronize();  // the clock below starts from a drained device, like enact()'s timer
  const auto traffic0 = comm.traffic();
  const auto t0 = std::chrono::steady_clock::now();

  int rounds = 0, dense = 0, reduced = 0, large = 0;
  long long found_total = 0;
  const int64_t* recv_prev = nullptr;
  int64_t slot_prev = 0;
  int32_t fmt_prev = GRX_RECV_PAIRS;
  void* snapshot = edge_op == GRX_OP_SSSP ? (void*)p.snapshot.data() : nullptr;
  // A failure on THIS rank (a buffer of its step overflowed, a kernel launch was refused, ...) must
  // not leave the others inside a collective: the rank keeps taking part in the protocol without
  // computing, announces the failure in its next send-slot header, and every rank leaves the loop
  // in that same superstep with an error.
  int failed = GRX_OK;
  std::string why;
  auto attempt = [&](auto&& body) {
    if (failed != GRX_OK)
      return;
    try {
      const int rc = body();
      if (rc != GRX_OK) {
        failed = rc;
        why = last_error();
      }
    } catch (const superstep_timeout_t&) {
      throw;
    } catch (const std::exception& e) {
      failed = GRX_ERR_RUNTIME;
      why = e.what();
    }
  };
  // test hook: GRX_PARTITIONED_FAIL_AT="<rank>:<superstep>" makes that rank's step fail there
  int fail_rank = -1, fail_round = -1;
  if (const char* e = std::getenv("GRX_PARTITIONED_FAIL_AT"))
    std::sscanf(e, "%d:%d", &fail_rank, &fail_round);
  p.ctx->superstep_finds_hint = -1;
  p.ctx->superstep_found_so_far = 0;
  for (;;) {
    attempt([&] {
      if (rank == fail_rank && rounds == fail_round) {
        last_error() = "injected failure (GRX_PARTITIONED_FAIL_AT)";
        return (int)GRX_ERR_RUNTIME;
      }
      return grx_partitioned_step(p.ctx, p.local, &p.opts, edge_op, labels, p.stamp.data(),
                                  p.sent.data(), rounds, recv_prev, fmt_prev, world, slot_prev, rank,
                                  p.lo, p.hi, p.frontier.data(), (int64_t)p.frontier.size(),
                                  reinterpret_cast<uint64_t*>(p.fcount.data()), p.scratch.data(),
                                  (int64_t)p.scratch.size(), p.send.data(), (int64_t)p.send.size(),
                                  snapshot);
    });
    if (failed != GRX_OK) {
      mark_failed_kernel<<<1, 1, 0, stream>>>(p.send.data());
      GRX_HIP_CHECK(hipGetLastError());
    }
    // every rank's [count | first pairs] -> every rank, on the stream that packed them
    comm.all_gather(p.send.data(), p.recv.data(), (std::size_t)p.slot0 * 8, stream);
    publish_heads_kernel<<<1, 64 * ((world + 63) / 64), 0, stream>>>(p.recv.data(), p.slot0, world,
                                                                     p.heads->data(),
                                                                     ++p.head_sequence);
    GRX_HIP_CHECK(hipGetLastError());
    const unsigned long long* counts = await_heads(p, sc);  // the ONE host wait of the superstep
    long long most = 0, sum = 0;
    int failed_peer = -1;
    for (int r = 0; r < world; ++r) {
      if (counts[r] == FAILED_HEAD) {
        failed_peer = failed_peer < 0 ? r : failed_peer;
        continue;
      }
      most = std::max<long long>(most, (long long)counts[r]);
      sum += (long long)counts[r];
    }
    if (failed_peer >= 0) {  // the same verdict on every rank, from the same gather
      sc.synchronize();
      p.ctx->pending_sequence = 0;
      if (failed != GRX_OK) {
        last_error() = "partitioned run: superstep " + std::to_string(rounds) + " failed on this rank (" +
                       std::to_string(rank) + "): " + why;
        return failed;
      }
      last_error() = "partitioned run: superstep " + std::to_string(rounds) + " failed on rank " +
                     std::to_string(failed_peer) + "; this rank (" + std::to_string(rank) + ") stops with it";
      return (int)GRX_ERR_PEER;
    }
    if (most == 0)
      break;  // no rank improved anything: every replica is final
    // every rank takes the same branch (same gathered counts)
    const int64_t* recv = p.recv.data();
    int64_t slot = p.slot0;
    int32_t fmt = GRX_RECV_PAIRS;
    if (edge_op == GRX_OP_BFS && p.dense_threshold >= 0 && most > p.dense_threshold) {
      // dense level: V/8 bytes per rank instead of 8 bytes per discovery
      attempt([&] {
        return grx_partitioned_level_bitmap(p.ctx, (const int32_t*)labels, p.n, rounds + 1,
                                            p.bits.data(), (int64_t)p.bits.size());
      });
      comm.all_gather(p.bits.data(), p.recv_bits.data(), (std::size_t)p.words * 8, stream);
      recv = p.recv_bits.data();
      slot = p.words;
      fmt = GRX_RECV_LEVEL_BITMAP;
      ++dense;
    } else if (edge_op == GRX_OP_SSSP && p.replica_threshold >= 0 && most > p.replica_threshold) {
      // the replicas already hold each rank's own improvements: combine them in place; the next
      // step admits what fell below its pre-advance snapshot
      comm.all_reduce(labels, (std::size_t)p.n, gcuda::collective_dtype_t::float32,
                      gcuda::collective_op_t::min, stream);
      recv = reinterpret_cast<const int64_t*>(p.snapshot.data());
      slot = 0;
      fmt = GRX_RECV_REPLICA_MIN;
      ++reduced;
    } else if (most > p.slot0 - 1) {
      // recv_big was sized for the largest slot when the plan was made: nothing can fail here
      slot = std::min<int64_t>(((most + 1 + 4095) / 4096) * 4096, (int64_t)p.send.size());
      comm.all_gather(p.send.data(), p.recv_big.data(), (std::size_t)slot * 8, stream);
      recv = p.recv_big.data();
      ++large;
    }
    found_total += sum;
    p.ctx->superstep_finds_hint = sum;  // what the next step's owned frontiers add up to, at most
    p.ctx->superstep_found_so_far = found_total;
    recv_prev = recv;
    slot_prev = slot;
    fmt_prev = fmt;
    ++rounds;
  }
  p.ctx->superstep_finds_hint = -1;
  if (p.local->renumbered_slice) {  // inside the timed run, like finalize() of the single-GPU clients
    deliver_kernel<<<grid, 256, 0, stream>>>(reinterpret_cast<const int32_t*>(labels),
                                             p.local->hot_vertex_of.data(), p.n,
                                             reinterpret_cast<int32_t*>(caller_labels));
    GRX_HIP_CHECK(hipGetLastError());
  }
  sc.synchronize();
  const auto t1 = std::chrono::steady_clock::now();
  // the last step's hand-off has landed: report a buffer overflow of that step here
  if (p.ctx->pending_sequence) {
    unsigned long long* m = operators::advance::detail::await_counters(sc, p.ctx->pending_sequence);
    p.ctx->pending_sequence = 0;
    error::throw_if_exception(m[hip::kernels::C_OVERFLOW] != 0,
                              "grx_partitioned_run: a buffer of the last superstep overflowed");
  }
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    stats->elapsed_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    stats->supersteps = rounds + 1;
    const auto& tr = comm.traffic();
    stats->collectives = (int32_t)((tr.all_gathers - traffic0.all_gathers) +
                                   (tr.all_reduces - traffic0.all_reduces));
    stats->bytes_sent = (int64_t)(tr.bytes_sent - traffic0.bytes_sent);
    stats->bitmap_supersteps = dense;
    stats->allreduce_supersteps = reduced;
    stats->large_gather_supersteps = large;
    stats->pairs_exchanged = found_total;
  }
  return (int)GRX_OK;
}

}  // namespace

extern "C" {

// ---- the job ------------------------------------------------------------------------------------

int grx_job_unique_id(void* id128) {
  if (!id128)
    return invalid("grx_job_unique_id: NULL argument");
  return guarded([&] {
    gcuda::rccl::unique_id(id128);
    return (int)GRX_OK;
  });
}

int grx_context_attach_rccl(grx_context_t ctx, int rank, int world, const void* id128) {
  if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world)
    return invalid("grx_context_attach_rccl: bad arguments");
  return guarded([&] {
    ctx->mc->attach_job(rank, world, gcuda::rccl::make(rank, world, id128, ctx->device));
    return (int)GRX_OK;
  });
}

namespace {
struct hooks_state_t {
  grx_all_gather_fn all_gather;
  grx_all_reduce_fn all_reduce;
  void* user;
};
}  // namespace

int grx_context_attach_collectives(grx_context_t ctx, int rank, int world,
                                   grx_all_gather_fn all_gather, grx_all_reduce_fn all_reduce,
                                   void* user) {
  if (!ctx || world < 1 || rank < 0 || rank >= world || (world > 1 && (!all_gather || !all_reduce)))
    return invalid("grx_context_attach_collectives: bad arguments");
  return guarded([&] {
    gcuda::collective_table_t t;
    t.state = new hooks_state_t{all_gather, all_reduce, user};
    t.name = "hooks";
    t.stream_ordered = false;
    t.all_gather = [](void* state, const void* d_send, void* d_recv, std::size_t bytes,
                      hipStream_t stream) -> int {
      auto* h = static_cast<hooks_state_t*>(state);
      return h->all_gather(h->user, d_send, d_recv, (uint64_t)bytes, (void*)stream);
    };
    t.all_reduce = [](void* state, void* d_buffer, std::size_t count, int dtype, int op,
                      hipStream_t stream) -> int {
      auto* h = static_cast<hooks_state_t*>(state);
      return h->all_reduce(h->user, d_buffer, (uint64_t)count, dtype, op, (void*)stream);
    };
    t.destroy = [](void* state) { delete static_cast<hooks_state_t*>(state); };
    ctx->mc->attach_job(rank, world, t);
    return (int)GRX_OK;
  });
}

int grx_context_detach(grx_context_t ctx) {
  if (!ctx)
    return invalid("context is NULL");
  return guarded([&] {
    ctx->mc->detach_job();
    return (int)GRX_OK;
  });
}

int grx_context_job_info(grx_context_t ctx, int32_t* rank, int32_t* world, char* backend,
                         size_t backend_len) {
  if (!ctx)
    return invalid("context is NULL");
  if (rank) *rank = ctx->mc->rank();
  if (world) *world = ctx->mc->world_size();
  if (backend && backend_len) {
    std::strncpy(backend, ctx->mc->communicator().backend(), backend_len - 1);
    backend[backend_len - 1] = 0;
  }
  return GRX_OK;
}

// ---- the plan -----------------------------------------------------------------------------------

int grx_partitioned_create(grx_context_t ctx, grx_graph_t local, int32_t row_begin, int32_t row_end,
                           const grx_options* opt, int64_t small_slot, int64_t dense_threshold,
                           int64_t replica_threshold, grx_partitioned_t* out) {
  if (!ctx || !local || !out || row_begin < 0 || row_end < row_begin || row_end > local->n_rows ||
      local->n_rows < 1)
    return invalid("grx_partitioned_create: bad arguments");
  return guarded([&] {
    auto p = std::make_unique<grx_partitioned_s>();
    p->ctx = ctx;
    p->local = local;
    p->lo = row_begin;
    p->hi = row_end;
    p->n = local->n_rows;
    grx_default_options(&p->opts);
    if (opt)
      p->opts = *opt;
    p->opts.holes_layout = 0;
    const int world = ctx->mc->world_size();
    const int64_t n = p->n;
    // every rank derives the same slot sizes from (small_slot, V)
    p->slot0 = std::max<int64_t>(2, std::min<int64_t>(small_slot > 0 ? small_slot : (1 << 15), n + 2));
    p->dense_threshold =
        dense_threshold != 0 ? dense_threshold : std::max<int64_t>(n / 64, p->slot0 - 1);
    p->replica_threshold = replica_threshold != 0
                               ? replica_threshold
                               : std::max<int64_t>(n / std::max(world, 1), p->slot0 - 1);
    p->words = (n + 63) / 64;
    const std::size_t own = (std::size_t)std::max<int32_t>(row_end - row_begin, 1);
    p->stamp.resize((std::size_t)n);
    p->sent.resize((std::size_t)n);
    p->frontier.resize(own + 64);
    p->scratch.resize((std::size_t)std::max<int64_t>(local->nnz, 1) + (std::size_t)n + 64);
    p->send.resize((std::size_t)n + 2);
    p->send.zero();
    p->fcount.resize(1);
    p->fcount.zero();
    p->recv.resize((std::size_t)((int64_t)world * p->slot0));
    p->recv.zero();
    // the second, larger pair gather of a superstep takes at most a whole send buffer per rank:
    // sized here so that the loop never allocates (an allocation failing on ONE rank in the middle
    // of a superstep would leave the others inside the collective)
    p->recv_big.resize((std::size_t)((int64_t)world * (int64_t)p->send.size()));
    p->bits.resize((std::size_t)p->words);
    p->recv_bits.resize((std::size_t)((int64_t)world * p->words));
    p->snapshot.resize((std::size_t)n);
    if (local->renumbered_slice)
      p->renumbered_labels.resize((std::size_t)n);
    p->heads = std::make_unique<hip::pinned_t<unsigned long long>>((std::size_t)world + 1);
    GRX_HIP_CHECK(hipDeviceSynchronize());
    *out = p.release();
    return (int)GRX_OK;
  });
}

int grx_partitioned_destroy(grx_partitioned_t plan) {
  if (!plan)
    return GRX_OK;
  return guarded([&] {
    delete plan;
    return (int)GRX_OK;
  });
}

int grx_partitioned_run(grx_partitioned_t plan, int32_t edge_op, int32_t source, void* d_labels,
                        grx_partitioned_stats* stats) {
  if (!plan || !d_labels)
    return invalid("grx_partitioned_run: NULL argument");
  if (edge_op != GRX_OP_BFS && edge_op != GRX_OP_SSSP)
    return unsupported("grx_partitioned_run: edge_op must be GRX_OP_BFS or GRX_OP_SSSP");
  if (source < 0 || source >= plan->n)
    return invalid("grx_partitioned_run: source out of range");
  if (plan->ctx->mc->world_size() * plan->slot0 != (int64_t)plan->recv.size())
    return invalid("grx_partitioned_run: the context's job changed since the plan was created");
  return guarded([&] {
    try {
      if (edge_op == GRX_OP_BFS)
        return run_as<int32_t>(*plan, edge_op, source, (int32_t*)d_labels, INT32_MAX, stats);
      return run_as<float>(*plan, edge_op, source, (float*)d_labels, FLT_MAX, stats);
    } catch (const superstep_timeout_t& e) {
      last_error() = e.what();
      return (int)GRX_ERR_TIMEOUT;
    }
  });
}

int grx_partitioned_pagerank(grx_partitioned_t plan, float alpha, float tol, int32_t max_iterations,
                             float* d_p, grx_partitioned_stats* stats) {
  if (!plan || !d_p)
    return invalid("grx_partitioned_pagerank: NULL argument");
  return guarded([&] {
    auto& p = *plan;
    auto& sc = p.ctx->single();
    auto& comm = p.ctx->mc->communicator();
    hipStream_t stream = sc.stream();
    const std::size_t n = (std::size_t)p.n;
    p.scale.resize(n);
    p.partial.resize(n + 1);
    // a slice of a renumbered copy: the iterations run on a renumbered vector, the caller's is
    // written once at the end (ranks are per vertex: the numbering does not change them)
    float* const caller_p = d_p;
    if (p.local->renumbered_slice)
      d_p = reinterpret_cast<float*>(p.renumbered_labels.data());
    hip::fill(d_p, n, 1.0f / (float)n, stream);
    sc.synchronize();
    const auto traffic0 = comm.traffic();
    const auto t0 = std::chrono::steady_clock::now();
    int it = 0;
    float err = 0;
    const unsigned grid = (unsigned)sc.compute_units() * 8;
    unsigned long long* counters = sc.workspace().counters();
    for (;;) {
      int rc = grx_pagerank_partitioned_scatter(p.ctx, p.local, alpha, d_p, p.scale.data(), it == 0,
                                                p.partial.data(), p.lo, p.hi, &p.opts);
      if (rc != GRX_OK)
        return rc;
      comm.all_reduce(p.partial.data(), n + 1, gcuda::collective_dtype_t::float32,
                      gcuda::collective_op_t::sum, stream);
      // identical update on every rank: the replicas stay identical, the stop test needs no
      // further collective
      pagerank_update_kernel<<<grid, 256, 0, stream>>>(d_p, p.partial.data(), (int64_t)n,
                                                       1.0f - alpha,
                                                       counters + hip::kernels::C_SELECT);
      GRX_HIP_CHECK(hipGetLastError());
      unsigned long long* m = operators::advance::detail::fetch_counters(sc);
      const unsigned bits = (unsigned)m[hip::kernels::C_SELECT];
      std::memcpy(&err, &bits, 4);
      ++it;
      if (err < tol || (max_iterations && it >= max_iterations))
        break;
    }
    if (p.local->renumbered_slice) {
      deliver_kernel<<<grid, 256, 0, stream>>>(reinterpret_cast<const int32_t*>(d_p),
                                               p.local->hot_vertex_of.data(), (int64_t)n,
                                               reinterpret_cast<int32_t*>(caller_p));
      GRX_HIP_CHECK(hipGetLastError());
    }
    sc.synchronize();
    const auto t1 = std::chrono::steady_clock::now();
    if (stats) {
      std::memset(stats, 0, sizeof *stats);
      stats->elapsed_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
      stats->iterations = it;
      stats->last_error = err;
      const auto& tr = comm.traffic();
      stats->collectives = (int32_t)((tr.all_gathers - traffic0.all_gathers) +
                                     (tr.all_reduces - traffic0.all_reduces));
      stats->bytes_sent = (int64_t)(tr.bytes_sent - traffic0.bytes_sent);
    }
    return (int)GRX_OK;
  });
}

}  // extern "C"
