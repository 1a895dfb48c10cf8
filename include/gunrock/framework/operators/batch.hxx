/**
 * @file batch.hxx
 * @brief operators::batch::execute -- run `number_of_jobs` independent host
 * functions (typically whole algorithm runs, each with its own context) on host
 * threads sharing the GPU; total_elapsed[0] = sum of the values they return.
 *
 * Same signature as reference framework/operators/batch/batch.hxx:61-79.  The
 * reference starts one std::thread per job at once; here a fixed pool
 * (GRX_BATCH_THREADS, default min(jobs, 8)) pulls job indices from an atomic
 * counter, so 10^4 seeds do not mean 10^4 threads / streams.  An exception in a
 * job is re-thrown on the caller's thread after every worker has stopped.
 */
#pragma once

#include <atomic>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include <gunrock/hip/runtime.hxx>

namespace gunrock {
namespace operators {
namespace batch {

template <typename function_t, typename... args_t>
void execute(function_t f, std::size_t number_of_jobs, float* total_elapsed, args_t&... args) {
  std::vector<float> elapsed(number_of_jobs, 0.0f);
  std::size_t workers = number_of_jobs < 8 ? number_of_jobs : 8;
  if (const char* e = std::getenv("GRX_BATCH_THREADS"))
    workers = (std::size_t)std::max(1, std::atoi(e));
  if (workers > number_of_jobs)
    workers = number_of_jobs;
  int device = 0;
  (void)hipGetDevice(&device);
  std::atomic<std::size_t> next{0};
  std::exception_ptr failure;
  std::mutex failure_mutex;
  std::vector<std::thread> pool;
  for (std::size_t w = 0; w < workers; ++w) {
    pool.emplace_back([&]() {
      (void)hipSetDevice(device);
      for (;;) {
        const std::size_t j = next.fetch_add(1);
        if (j >= number_of_jobs)
          break;
        try {
          elapsed[j] = f(j);
        } catch (...) {
          std::lock_guard<std::mutex> lock(failure_mutex);
          if (!failure)
            failure = std::current_exception();
          next.store(number_of_jobs);
        }
      }
    });
  }
  for (auto& t : pool)
    t.join();
  if (failure)
    std::rethrow_exception(failure);
  float total = 0.0f;
  for (float e : elapsed)
    total += e;
  total_elapsed[0] = total;
}

}  // namespace batch
}  // namespace operators
}  // namespace gunrock
