/**
 * @file communicator.hxx
 * @brief The collective transport of a multi-GPU job: one process per GPU, ranks exchange
 * frontiers / label replicas between BSP supersteps.
 *
 * The reference has nothing here: its multi_context_t holds several LOCAL device contexts and
 * every operator throws for more than one (cuda/context.hxx:136-206,
 * framework/operators/advance/advance.hxx:125-128).  This engine's model is one rank per MI355X
 * with RCCL over xGMI: communicator_t = {rank, world size, a table of two collectives}.  The
 * production table (rccl::make) calls ncclAllGather / ncclAllReduce directly on the stream the
 * engine's kernels run on -- the collective is ordered between the kernel that packs a send slot
 * and the kernel that admits the gathered slots by stream order alone, no event, no host wait.
 * A second table carries host callbacks (hooks): the same supersteps then run over any transport
 * the host provides (the test rigs use torch.distributed/gloo with ranks sharing one GPU).
 *
 * rccl::make is a function template on purpose: a translation unit that never attaches a job to
 * RCCL references no nccl symbol and needs no -lrccl (the reference's harnesses link unchanged).
 */
#pragma once

#include <rccl/rccl.h>

#include <gunrock/hip/runtime.hxx>

namespace gunrock {
namespace gcuda {

enum class collective_dtype_t : int { int32 = 0, float32 = 1, int64 = 2 };
enum class collective_op_t : int { min = 0, sum = 1, max = 2 };

/// Two collectives + teardown, as plain function pointers (crosses the C ABI unchanged).
struct collective_table_t {
  /// recv[world * bytes_per_rank] <- concatenation over ranks of send[bytes_per_rank]; both device.
  int (*all_gather)(void* state, const void* d_send, void* d_recv, std::size_t bytes_per_rank,
                    hipStream_t stream) = nullptr;
  /// buffer[count] <- elementwise op over ranks, in place; device.
  int (*all_reduce)(void* state, void* d_buffer, std::size_t count, int dtype, int op,
                    hipStream_t stream) = nullptr;
  void (*destroy)(void* state) = nullptr;
  void* state = nullptr;
  const char* name = "single";
  /// true: the collective is enqueued on `stream` and returns at once (RCCL); false: it has
  /// completed, on the host, when the call returns (host-staged transports)
  bool stream_ordered = true;
};

class communicator_t {
 public:
  communicator_t() = default;
  communicator_t(const communicator_t&) = delete;
  communicator_t& operator=(const communicator_t&) = delete;
  ~communicator_t() { detach(); }

  /// A transport attached to a job of ONE rank is still used (the collectives then run through
  /// RCCL / the callbacks with a single participant): the production call sequence can be
  /// exercised on a one-GPU box.
  void attach(int rank, int world, collective_table_t table) {
    error::throw_if_exception(world < 1 || rank < 0 || rank >= world, "communicator: bad rank / world size");
    error::throw_if_exception(world > 1 && (!table.all_gather || !table.all_reduce),
                              "communicator: a job of several ranks needs both collectives");
    detach();
    rank_ = rank;
    world_ = world;
    table_ = table;
  }
  void detach() {
    if (table_.destroy)
      table_.destroy(table_.state);
    table_ = collective_table_t();
    rank_ = 0;
    world_ = 1;
  }

  int rank() const { return rank_; }
  int world_size() const { return world_; }
  /// A job is attached (possibly of one rank): enactors exchange frontiers between supersteps.
  bool attached() const { return table_.all_gather != nullptr; }
  const char* backend() const { return table_.name; }
  bool stream_ordered() const { return table_.stream_ordered; }

  /// Counters of what went over the wire (reported by the benchmarks).
  struct traffic_t {
    unsigned long long all_gathers = 0, all_reduces = 0, bytes_sent = 0;
  };
  const traffic_t& traffic() const { return traffic_; }
  void reset_traffic() { traffic_ = traffic_t(); }

  void all_gather(const void* d_send, void* d_recv, std::size_t bytes_per_rank, hipStream_t stream) {
    ++traffic_.all_gathers;
    traffic_.bytes_sent += bytes_per_rank;
    if (!table_.all_gather) {  // no transport attached (a job of one): the rank's own slot
      if (d_send != d_recv && bytes_per_rank)
        GRX_HIP_CHECK(hipMemcpyAsync(d_recv, d_send, bytes_per_rank, hipMemcpyDeviceToDevice, stream));
      return;
    }
    const int rc = table_.all_gather(table_.state, d_send, d_recv, bytes_per_rank, stream);
    error::throw_if_exception(rc != 0, std::string("all_gather failed on the '") + table_.name +
                                           "' transport (" + std::to_string(rc) + ")");
  }

  void all_reduce(void* d_buffer, std::size_t count, collective_dtype_t dtype, collective_op_t op,
                  hipStream_t stream) {
    ++traffic_.all_reduces;
    traffic_.bytes_sent += count * (dtype == collective_dtype_t::int64 ? 8 : 4);
    if (!table_.all_reduce)  // no transport attached (a job of one)
      return;
    const int rc = table_.all_reduce(table_.state, d_buffer, count, (int)dtype, (int)op, stream);
    error::throw_if_exception(rc != 0, std::string("all_reduce failed on the '") + table_.name +
                                           "' transport (" + std::to_string(rc) + ")");
  }

 private:
  int rank_ = 0;
  int world_ = 1;
  collective_table_t table_;
  traffic_t traffic_;
};

// ---------------------------------------------------------------------------------------------
// RCCL (production): direct calls on the engine's stream.
// ---------------------------------------------------------------------------------------------
namespace rccl {

constexpr std::size_t unique_id_bytes = NCCL_UNIQUE_ID_BYTES;  // 128

inline void check(ncclResult_t r, const char* what) {
  error::throw_if_exception(r != ncclSuccess, std::string(what) + ": " + ncclGetErrorString(r));
}

/// Rank 0 creates the job's id and hands the 128 bytes to every other rank (any side channel).
template <int header_only = 0>
void unique_id(void* out128) {
  ncclUniqueId id;
  check(ncclGetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out128, &id, unique_id_bytes);
}

/// Collective over all ranks of the job: every rank calls it with the same id.
template <int header_only = 0>
collective_table_t make(int rank, int world, const void* id128, int device) {
  GRX_HIP_CHECK(hipSetDevice(device));
  ncclUniqueId id;
  std::memcpy(&id, id128, unique_id_bytes);
  ncclComm_t comm = nullptr;
  check(ncclCommInitRank(&comm, world, id, rank), "ncclCommInitRank");
  collective_table_t t;
  t.state = comm;
  t.name = "rccl";
  t.stream_ordered = true;
  t.all_gather = [](void* state, const void* d_send, void* d_recv, std::size_t bytes_per_rank,
                    hipStream_t stream) -> int {
    return (int)ncclAllGather(d_send, d_recv, bytes_per_rank, ncclInt8, (ncclComm_t)state, stream);
  };
  t.all_reduce = [](void* state, void* d_buffer, std::size_t count, int dtype, int op,
                    hipStream_t stream) -> int {
    const ncclDataType_t ty = dtype == (int)collective_dtype_t::float32 ? ncclFloat32
                              : dtype == (int)collective_dtype_t::int64 ? ncclInt64
                                                                        : ncclInt32;
    const ncclRedOp_t ro = op == (int)collective_op_t::sum ? ncclSum
                           : op == (int)collective_op_t::max ? ncclMax
                                                             : ncclMin;
    return (int)ncclAllReduce(d_buffer, d_buffer, count, ty, ro, (ncclComm_t)state, stream);
  };
  t.destroy = [](void* state) {
    if (state)
      (void)ncclCommDestroy((ncclComm_t)state);
  };
  return t;
}

}  // namespace rccl
}  // namespace gcuda
}  // namespace gunrock
