/**
 * @file math.hxx
 * @brief Integer helpers and the device atomics facade used by client lambdas.
 *
 * Surface of reference include/gunrock/util/math.hxx:24-129 and
 * include/gunrock/cuda/atomic_functions.hxx:36-123 (atomic::{add,min,max,cas,exch}
 * return the OLD value).  gfx950 implementation notes:
 *  - float min/max are ONE integer atomic on the IEEE bit pattern (sign-split
 *    trick) instead of the reference's compare-and-swap loop;
 *  - min/max first read the word with a relaxed agent-scope load and skip the
 *    read-modify-write when it cannot change the word.  Device-scope RMWs execute
 *    at the memory side on MI355X (they bypass the per-XCD L2), so a BFS/SSSP
 *    relaxation that loses -- the overwhelming majority on a power-law graph --
 *    costs one L2-served load instead of a fabric atomic.  The value returned in
 *    that case was in memory during the call, which is all a relaxed atomic
 *    promises.  Define GRX_ATOMIC_NO_PRETEST to disable.
 *  - the host side of these functions really updates memory (the reference's host
 *    branch returns without storing, SURVEY.md 8a' q4).
 */
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace gunrock {
namespace math {

template <typename type_t>
__host__ __device__ __forceinline__ constexpr type_t divide_round_up(type_t const& a,
                                                                      type_t const& b) {
  return (a + b - 1) / b;
}

template <typename type_t>
constexpr type_t log2(const type_t& n) {
  return (n < 2) ? 0 : 1 + log2(n / 2);
}

template <typename type_t>
__host__ __device__ constexpr const type_t& max(const type_t& a, const type_t& b) {
  return (a < b) ? b : a;
}

template <typename type_t>
__host__ __device__ constexpr const type_t& min(const type_t& a, const type_t& b) {
  return (b < a) ? b : a;
}

namespace atomic {

namespace detail {
#if defined(__HIP_DEVICE_COMPILE__)
template <typename type_t>
__device__ __forceinline__ type_t peek(type_t* address) {
#ifdef GRX_ATOMIC_PRETEST_PLAIN
  return __builtin_nontemporal_load(address);  // experiment only
#else
  return __hip_atomic_load(address, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ float fmin_rmw(float* address, float value) {
  // IEEE-754 order equals signed-int order for non-negative values and reversed
  // unsigned order for negative ones.
  return (value >= 0.0f)
             ? __int_as_float(::atomicMin(reinterpret_cast<int*>(address), __float_as_int(value)))
             : __uint_as_float(
                   ::atomicMax(reinterpret_cast<unsigned int*>(address), __float_as_uint(value)));
}
__device__ __forceinline__ float fmax_rmw(float* address, float value) {
  return (value >= 0.0f)
             ? __int_as_float(::atomicMax(reinterpret_cast<int*>(address), __float_as_int(value)))
             : __uint_as_float(
                   ::atomicMin(reinterpret_cast<unsigned int*>(address), __float_as_uint(value)));
}
__device__ __forceinline__ double dmin_rmw(double* address, double value) {
  return (value >= 0.0)
             ? __longlong_as_double(::atomicMin(reinterpret_cast<long long*>(address),
                                                __double_as_longlong(value)))
             : __longlong_as_double((long long)::atomicMax(
                   reinterpret_cast<unsigned long long*>(address),
                   (unsigned long long)__double_as_longlong(value)));
}
__device__ __forceinline__ double dmax_rmw(double* address, double value) {
  return (value >= 0.0)
             ? __longlong_as_double(::atomicMax(reinterpret_cast<long long*>(address),
                                                __double_as_longlong(value)))
             : __longlong_as_double((long long)::atomicMin(
                   reinterpret_cast<unsigned long long*>(address),
                   (unsigned long long)__double_as_longlong(value)));
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
/// atomic add of the active lanes of a wave in which NEIGHBOURING lanes that add to the SAME word
/// (a run) issue ONE read-modify-write carrying the run's sum.  A wave that walks edges grouped by
/// destination (operators/by_destination.hxx) and calls `add(&p[dst], x)` per edge -- `pr.hxx`'s
/// push -- then sends one RMW per destination and wave instead of one per edge; the hottest
/// destination of a directed R-MAT-24 receives 370 K of them per iteration, and one 128-B line
/// retires ~90 RMWs/us however many CUs queue for it.  Every lane still gets a value the word held
/// in ONE sequential order of the adds: the old value its run's RMW returned plus the values of the
/// run's lanes before it.  No run longer than one lane (the wave-uniform common case of a row-major
/// expansion): 2 cross-lane moves and a ballot, then the plain atomic.
template <typename type_t>
__device__ __forceinline__ type_t add_runs(type_t* address, type_t value) {
  const unsigned lane = __lane_id();
  const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
  const unsigned long long a = reinterpret_cast<unsigned long long>(address);
  const unsigned lo = (unsigned)__shfl_up((int)(unsigned)a, 1);
  const unsigned hi = (unsigned)__shfl_up((int)(unsigned)(a >> 32), 1);
  const bool follows = lane > 0 && ((active >> (lane - 1)) & 1) &&
                       (((unsigned long long)hi << 32) | lo) == a;
  const unsigned long long heads = __builtin_amdgcn_ballot_w64(!follows);
  if (heads == active)
    return ::atomicAdd(address, value);
  // first lane of this lane's run: the highest head at or below it
  const unsigned first = 63u - (unsigned)__clzll((long long)(heads & (~0ull >> (63u - lane))));
  type_t sum = value;  // inclusive sum over the run's lanes up to this one
#pragma unroll
  for (unsigned d = 1; d < 64; d <<= 1) {
    const type_t below = __shfl_up(sum, d);
    if (lane >= first + d)
      sum += below;
  }
  // last lane of the run: the next lane is a head, inactive, or there is none
  const bool last = lane == 63 || !((active >> (lane + 1)) & 1) || ((heads >> (lane + 1)) & 1);
  const unsigned long long lasts = __builtin_amdgcn_ballot_w64(last);
  type_t old = type_t(0);
  if (last)
    old = ::atomicAdd(address, sum);
  const unsigned mine = lane + (unsigned)__builtin_ctzll(lasts >> lane);
  old = __shfl(old, (int)mine);
  return old + (sum - value);
}
#endif
}  // namespace detail

template <typename type_t>
__host__ __device__ __forceinline__ type_t add(type_t* address, type_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef GRX_ATOMIC_NO_RUNS
  if constexpr (std::is_same<type_t, float>::value || std::is_same<type_t, double>::value)
    return detail::add_runs(address, value);
  else
#endif
    return ::atomicAdd(address, value);
#else
  type_t old = *address;
  *address = old + value;
  return old;
#endif
}

template <typename type_t>
__host__ __device__ __forceinline__ type_t min(type_t* address, type_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef GRX_ATOMIC_NO_PRETEST
#ifdef GRX_ATOMIC_MONOTONE_L1
  // opt-in: valid only for words that never increase while a kernel runs (BFS/SSSP labels):
  // an L1-cached copy can only be too LARGE, so "no improvement" from it is always right.
  {
    type_t cached = *address;
    if (!(value < cached))
      return cached;
  }
#endif
  type_t seen = detail::peek(address);
  if (!(value < seen))
    return seen;
#endif
  if constexpr (std::is_same<type_t, float>::value)
    return detail::fmin_rmw(address, value);
  else if constexpr (std::is_same<type_t, double>::value)
    return detail::dmin_rmw(address, value);
  else
    return ::atomicMin(address, value);
#else
  type_t old = *address;
  if (value < old)
    *address = value;
  return old;
#endif
}

template <typename type_t>
__host__ __device__ __forceinline__ type_t max(type_t* address, type_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef GRX_ATOMIC_NO_PRETEST
  type_t seen = detail::peek(address);
  if (!(seen < value))
    return seen;
#endif
  if constexpr (std::is_same<type_t, float>::value)
    return detail::fmax_rmw(address, value);
  else if constexpr (std::is_same<type_t, double>::value)
    return detail::dmax_rmw(address, value);
  else
    return ::atomicMax(address, value);
#else
  type_t old = *address;
  if (old < value)
    *address = value;
  return old;
#endif
}

template <typename type_t>
__host__ __device__ __forceinline__ type_t cas(type_t* address, type_t compare, type_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
  return ::atomicCAS(address, compare, value);
#else
  type_t old = *address;
  if (old == compare)
    *address = value;
  return old;
#endif
}

template <typename type_t>
__host__ __device__ __forceinline__ type_t exch(type_t* address, type_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef GRX_ATOMIC_NO_PRETEST
  // an exchange that would store what the word already holds is a read: linearise it at this
  // (agent-scope, L2-served) load and skip the memory-side RMW.  "Stamp" idioms -- the first of
  // many improvers of a vertex in one round wins `exch(&stamp[v], round) != round`, the others
  // find the stamp set -- issue one RMW per vertex and round instead of one per improvement
  // (SSSP iteration 1 on RMAT-22: ~4 improvements per vertex, and the kernel runs at the
  // chip's scattered-atomic rate).
  if (detail::peek(address) == value)
    return value;
#endif
  return ::atomicExch(address, value);
#else
  type_t old = *address;
  *address = value;
  return old;
#endif
}

/// Atomic OR, returns the previous word (not in the reference; used for dense "seen" bitmaps).
template <typename type_t>
__host__ __device__ __forceinline__ type_t bit_or(type_t* address, type_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
  return ::atomicOr(address, value);
#else
  type_t old = *address;
  *address = old | value;
  return old;
#endif
}

}  // namespace atomic
}  // namespace math

/// Plain (default cache policy) element load / store used by the graph views
/// and by client lambdas (reference util/load_store.hxx:23-42).
namespace thread {

template <typename type_t>
__host__ __device__ __forceinline__ type_t load(type_t* ptr) {
  return *ptr;
}

template <typename type_t>
__host__ __device__ __forceinline__ void store(type_t* ptr, const type_t& val) {
  *ptr = val;
}

}  // namespace thread
}  // namespace gunrock
