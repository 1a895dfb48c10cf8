/**
 * @file partitioned.hxx
 * @brief Vertex-partitioned (multi-GPU) execution of an UNCHANGED client: when the enactor's
 * context is attached to a job (gcuda::multi_context_t::attach_job, one process per GPU),
 * enactor_t::enact() exchanges the frontier between the ranks after every loop() -- the client's
 * problem / enactor / lambdas are not touched.
 *
 * The reference offers nothing here (every operator throws for more than one context,
 * framework/operators/advance/advance.hxx:125-128; enact() uses context 0 only,
 * framework/enactor.hxx:243-254); design per SURVEY.md 8(e):
 *   - rank r owns the vertex range [owned_begin, owned_end) and traverses its SLICE of the graph
 *     (global vertex ids; rows it does not own are empty), so the client's advance expands owned
 *     frontier vertices only;
 *   - the label array the client's lambdas test and update (BFS depth, SSSP distance) is
 *     REPLICATED; which array that is and how replicas combine is declared OUTSIDE the client
 *     headers by exchange_traits<problem_t> -- below for every problem that has
 *     `result.distances` (reference algorithms/bfs.hxx:40-47, sssp.hxx:44-53): combine = MIN;
 *   - after loop() the output frontier holds what this rank discovered (global ids, any owner,
 *     duplicates allowed).  exchange(): pack (vertex, label) pairs -> all-gather of the counts ->
 *     all-gather of the pairs (communicator_t: ncclAllGather on the context's stream) -> every
 *     rank min-combines every other rank's pairs into its replica (replicas stay coherent) and
 *     appends the OWNED vertices whose label improved -- or that it discovered itself -- to its
 *     next input frontier, once per superstep;
 *   - the job has converged when no rank discovered anything (sum of the gathered counts).
 * BSP levels are globally synchronised, so BFS depths equal the single-GPU ones; the SSSP fix point
 * is unique, so distances are bit-identical whatever the exchange order.
 * A problem without a declared combiner cannot be partitioned ("replicas only"): enact() throws.
 *
 * The C ABI's grx_partitioned_run is the tuned form of the same protocol for the two built-in
 * functors (fused enqueue-only supersteps, level bitmaps, replica all-reduce); this header is the
 * general one.
 */
#pragma once

#include <type_traits>
#include <vector>

#include <gunrock/framework/frontier.hxx>
#include <gunrock/framework/operators/advance.hxx>
#include <gunrock/hip/algorithms.hxx>
#include <gunrock/hip/context.hxx>
#include <gunrock/hip/kernels/exchange_kernels.hxx>

namespace gunrock {
namespace partitioned {

/// How a problem's state is replicated and combined.  Specialise for a problem type to make it
/// partitionable; the primary template says "not declared".
template <typename problem_t, typename = void>
struct exchange_traits {
  static constexpr bool enabled = false;
  using label_t = int;
  static label_t* labels(problem_t&) { return nullptr; }
};

/// Any problem with `result.distances` (the reference's bfs::problem_t and sssp::problem_t):
/// replicas combine with MIN.
template <typename problem_t>
struct exchange_traits<problem_t,
                       std::void_t<decltype(std::declval<problem_t&>().result.distances)>> {
  static constexpr bool enabled = true;
  using label_t = std::remove_pointer_t<decltype(std::declval<problem_t&>().result.distances)>;
  static label_t* labels(problem_t& p) { return p.result.distances; }
};

/// Buffers of one enactor's exchanges (allocated on first use, reused every superstep).
struct exchange_state_t {
  hip::device_array_t<int64_t> send;  // [count | pairs...]
  hip::device_array_t<int64_t> recv;  // world x slot
  hip::device_array_t<int64_t> heads; // world counts (gathered)
  hip::device_array_t<int32_t> stamp; // superstep in which a vertex last entered a frontier
  std::vector<int64_t> counts;
  int round = 0;
  unsigned long long supersteps = 0, pairs = 0;
};

/// Keep the owned elements of a (tiny) initial frontier: prepare_frontier pushed the source on
/// every rank, only its owner starts from it.
template <typename frontier_t>
void keep_owned(frontier_t& f, gcuda::multi_context_t& context) {
  // a client may have pushed its source with the STREAMED push_back (a fill kernel on the
  // context's non-blocking stream, nothing awaited): to_host() copies on the null stream, which
  // is not ordered after it
  context.get_context(0)->synchronize();
  auto h = f.to_host();
  std::vector<typename frontier_t::type_t> mine;
  for (auto v : h)
    if (context.owns((long long)v))
      mine.push_back(v);
  f.set_number_of_elements(0);
  for (auto v : mine)
    f.push_back(v);
}

/**
 * @brief One exchange: `found` (this rank's discoveries) -> `next` (the owned, improved vertices of
 * ALL ranks' discoveries).  Returns the number of discoveries of the whole job (0 = converged).
 */
template <typename label_t, typename frontier_t>
unsigned long long exchange(frontier_t& found, frontier_t& next, label_t* labels, std::size_t n_vertices,
                            exchange_state_t& st, gcuda::multi_context_t& context) {
  static_assert(sizeof(label_t) == 4 && sizeof(typename frontier_t::type_t) == 4,
                "the exchange packs a 32-bit vertex id and a 32-bit label into one word");
  namespace k = hip::kernels;
  auto& sc = *context.get_context(0);
  auto& comm = context.communicator();
  hipStream_t stream = sc.stream();
  const int world = comm.world_size(), rank = comm.rank();
  unsigned long long* counters = sc.workspace().counters();
  if (st.send.size() < n_vertices + 2 || st.heads.size() != (std::size_t)world) {
    st.send.resize(((n_vertices + 2 + 1023) / 1024) * 1024 + 1024);  // slots are cut in 1024-word steps
    st.heads.resize((std::size_t)world);
    st.stamp.resize(n_vertices);
    hip::fill(st.stamp.data(), n_vertices, int32_t(-1), stream);
    st.counts.assign((std::size_t)world, 0);
  }
  // 1. pack (vertex, label) of every find; duplicates and foreign vertices included
  const int64_t count = (int64_t)found.get_number_of_elements();
  int64_t packed = 0;
  if (count) {
    // a frontier may hold more entries than V (duplicates): pack at most what the slot takes;
    // duplicates carry the same label, so dropping the surplus loses nothing once deduplicated
    const unsigned grid = (unsigned)std::min<int64_t>((count + k::APPEND_TILE - 1) / k::APPEND_TILE,
                                                      (int64_t)sc.compute_units() * 8);
    k::pack_unique_pairs_kernel<label_t><<<grid, 256, 0, stream>>>(
        reinterpret_cast<const int32_t*>(found.data()), count, labels, st.stamp.data(),
        2 * st.round + 1, st.send.data(), (int64_t)st.send.size(), counters);
    GRX_HIP_CHECK(hipGetLastError());
  }
  {
    unsigned long long* m = operators::advance::detail::await_counters(
        sc, operators::advance::detail::publish_counters(sc, reinterpret_cast<long long*>(st.send.data()),
                                                         k::C_SELECT));
    error::throw_if_exception(m[k::C_OVERFLOW] != 0, "partitioned exchange: send slot overflow");
    packed = (int64_t)m[k::C_SELECT];
  }
  // 2. counts of every rank (8 bytes each), then the pairs in slots sized by the busiest rank
  comm.all_gather(st.send.data(), st.heads.data(), 8, stream);
  GRX_HIP_CHECK(hipMemcpyAsync(st.counts.data(), st.heads.data(), (std::size_t)world * 8,
                               hipMemcpyDeviceToHost, stream));
  sc.synchronize();
  int64_t most = 0;
  unsigned long long total = 0;
  for (int r = 0; r < world; ++r) {
    most = std::max<int64_t>(most, st.counts[(std::size_t)r]);
    total += (unsigned long long)st.counts[(std::size_t)r];
  }
  (void)packed;
  ++st.supersteps;
  st.pairs += total;
  if (total == 0) {
    next.set_number_of_elements(0);
    ++st.round;
    return 0;
  }
  const int64_t slot = ((most + 1 + 1023) / 1024) * 1024;
  if ((int64_t)st.recv.size() < (int64_t)world * slot)
    st.recv.resize((std::size_t)((int64_t)world * slot));
  comm.all_gather(st.send.data(), st.recv.data(), (std::size_t)slot * 8, stream);
  // 3. min-combine + owner admission (once per vertex and superstep)
  const std::size_t owned = (std::size_t)std::max<long long>(
      (context.owned_end() < 0 ? (long long)n_vertices : context.owned_end()) - context.owned_begin(), 1);
  if (next.get_capacity() < owned)
    next.reserve(owned);
  const int32_t lo = (int32_t)context.owned_begin();
  const int32_t hi = (int32_t)(context.owned_end() < 0 ? (long long)n_vertices : context.owned_end());
  const int64_t entries = (int64_t)world * (slot - 1);
  const unsigned grid = (unsigned)std::min<int64_t>(
      std::max<int64_t>((entries + k::APPEND_TILE - 1) / k::APPEND_TILE, 1), (int64_t)sc.compute_units() * 8);
  k::admit_kernel<label_t, true><<<grid, 256, 0, stream>>>(
      labels, st.stamp.data(), 2 * st.round + 2, st.recv.data(), world, slot, rank, lo, hi,
      reinterpret_cast<int32_t*>(next.data()), (unsigned long long)next.get_capacity(),
      counters + k::C_OUT, counters + k::C_OVERFLOW);
  GRX_HIP_CHECK(hipGetLastError());
  unsigned long long* m = operators::advance::detail::fetch_counters(sc);
  error::throw_if_exception(m[k::C_OVERFLOW] != 0, "partitioned exchange: next frontier overflow");
  next.set_number_of_elements((std::size_t)m[k::C_OUT]);  // work hint unknown: the advance sizes itself
  ++st.round;
  return total;
}

}  // namespace partitioned
}  // namespace gunrock
