// What one scattered 4-byte operation costs on one MI355X, by kind: plain load, agent-scope load,
// plain store, atomic min with / without a returned value at agent scope (executes at the memory
// side) and at workgroup scope (executes in the XCD's L2), over tables from 16 KB (L1-resident) to
// 64 MB.  Indices are hashed from the thread id (no index stream: the operation itself is timed).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/scatter_probe tools/scatter_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

enum { LOAD, LOAD_AGENT, STORE, MIN_AGENT_RET, MIN_AGENT_NORET, MIN_WG_RET, MIN_WG_NORET, KINDS };
static const char* names[KINDS] = {"plain load", "agent-scope load", "plain store",
                                   "atomic min agent scope, returned", "atomic min agent scope, no return",
                                   "atomic min workgroup scope, returned", "atomic min workgroup scope, no return"};

template <int KIND>
__global__ void __launch_bounds__(256) probe(unsigned* table, uint32_t mask, int per_thread, unsigned* sink) {
  unsigned acc = 0;
  uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll 4
  for (int k = 0; k < per_thread; ++k) {
    x = mix32(x + k);
    unsigned* p = table + (x & mask);
    const unsigned v = x >> 8;  // mostly larger than what is there after a while: mins mostly fail
    if (KIND == LOAD) acc += *p;
    if (KIND == LOAD_AGENT) acc += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (KIND == STORE) *p = v;
    if (KIND == MIN_AGENT_RET) acc += __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (KIND == MIN_AGENT_NORET) __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (KIND == MIN_WG_RET) acc += __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (KIND == MIN_WG_NORET) __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (acc == 0x12345678u) *sink = acc;
}

template <int KIND>
static int run(unsigned* table, size_t entries, unsigned* sink) {
  const int per_thread = 64, grid = 256 * 8 * 4;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(table, 0xff, entries * 4));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    probe<KIND><<<grid, 256>>>(table, (uint32_t)(entries - 1), per_thread, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
  }
  const double ops = (double)grid * 256 * per_thread;
  printf("  %-40s %8.1f G/s  (%.3f ms, %.0f M ops)\n", names[KIND], ops / best / 1e6, best, ops / 1e6);
  return 0;
}

int main() {
  unsigned *table, *sink;
  CK(hipMalloc(&table, 64u << 20));
  CK(hipMalloc(&sink, 4));
  for (size_t bytes : {16u << 10, 256u << 10, 2u << 20, 16u << 20, 64u << 20}) {
    printf("table %zu KB:\n", bytes >> 10);
    const size_t e = bytes / 4;
    if (run<LOAD>(table, e, sink) || run<LOAD_AGENT>(table, e, sink) || run<STORE>(table, e, sink) ||
        run<MIN_AGENT_RET>(table, e, sink) || run<MIN_AGENT_NORET>(table, e, sink) ||
        run<MIN_WG_RET>(table, e, sink) || run<MIN_WG_NORET>(table, e, sink))
      return 1;
  }
  return 0;
}
