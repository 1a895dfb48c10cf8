// What an LDS-resident "settled" bitmap in front of a random 4-byte label lookup can buy on one
// MI355X: the gather probe of gather_roof.hip (R-MAT-skewed indices from a streamed int32 array,
// agent-scope label loads) with the lookups of ids below a limit answered from a bitmap in LDS.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_filter_probe tools/lds_filter_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int BLOCK, bool FILTER>
__global__ void __launch_bounds__(BLOCK) probe_kernel(const int32_t* idx, size_t n, const uint32_t* table,
                                                       const uint32_t* bits, int32_t limit,
                                                       unsigned long long* sink) {
  extern __shared__ uint32_t s_bits[];
  if (FILTER) {
    for (int w = threadIdx.x; w < limit / 32; w += BLOCK)
      s_bits[w] = bits[w];
    __syncthreads();
  }
  unsigned long long acc = 0;
  const size_t stride = (size_t)gridDim.x * BLOCK * 4;
  for (size_t i0 = blockIdx.x * (size_t)BLOCK * 4 + threadIdx.x; i0 < n; i0 += stride) {
    int32_t j[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      j[k] = (i0 + k * BLOCK < n) ? __builtin_nontemporal_load(idx + i0 + k * BLOCK) : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (j[k] < 0)
        continue;
      if (FILTER && j[k] < limit && ((s_bits[j[k] >> 5] >> (j[k] & 31)) & 1u)) {
        acc += 1;
        continue;
      }
      acc += __hip_atomic_load(table + j[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (acc == 0x1234567812345678ull)
    *sink = acc;
}

static uint64_t mix(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

template <int BLOCK, bool FILTER>
static int run(const char* what, int blocks_per_cu, const int32_t* d_idx, size_t n, const uint32_t* table,
               const uint32_t* bits, int32_t limit, unsigned long long* sink, hipEvent_t a, hipEvent_t b) {
  const size_t lds = FILTER ? (size_t)limit / 8 : 0;
  CK(hipFuncSetAttribute((const void*)probe_kernel<BLOCK, FILTER>, hipFuncAttributeMaxDynamicSharedMemorySize,
                         (int)lds));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    probe_kernel<BLOCK, FILTER><<<256 * blocks_per_cu, BLOCK, lds>>>(d_idx, n, table, bits, limit, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("%-46s %7.3f ms = %6.1f G edges/s\n", what, best, n / best / 1e6);
  fflush(stdout);
  return 0;
}

int main() {
  const size_t n = 128u << 20;
  const int lg = 22;
  std::vector<int32_t> h(n);
  for (size_t i = 0; i < n; ++i) {
    uint64_t r[3];
    r[0] = mix(i * 2 + 1);
    r[1] = mix(r[0]);
    r[2] = mix(r[1]);
    uint32_t v = 0;
    for (int bit = 0; bit < lg; ++bit) {  // every id bit is 1 with probability 61/256 = 0.24
      const unsigned byte = (unsigned)((r[bit / 8] >> (8 * (bit % 8))) & 255);
      v |= (byte < 61 ? 1u : 0u) << bit;
    }
    h[i] = (int32_t)v;
  }
  int32_t* d_idx;
  uint32_t *table, *bits_all, *bits_half;
  unsigned long long* sink;
  CK(hipMalloc(&d_idx, n * 4));
  CK(hipMalloc(&sink, 8));
  CK(hipMalloc(&table, 4u << lg));
  CK(hipMalloc(&bits_all, 1u << 17));
  CK(hipMalloc(&bits_half, 1u << 17));
  CK(hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemset(table, 1, 4u << lg));
  CK(hipMemset(bits_all, 0xff, 1u << 17));
  CK(hipMemset(bits_half, 0x55, 1u << 17));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  size_t below[3] = {0, 0, 0};
  for (size_t i = 0; i < n; ++i) {
    below[0] += h[i] < (512 << 10);
    below[1] += h[i] < (768 << 10);
    below[2] += h[i] < (1024 << 10);
  }
  printf("ids below 512K / 768K / 1M: %.3f %.3f %.3f\n", below[0] / (double)n, below[1] / (double)n,
         below[2] / (double)n);
  if (run<256, false>("256 thr x 8/CU, no filter", 8, d_idx, n, table, bits_all, 0, sink, a, b)) return 1;
  if (run<512, false>("512 thr x 2/CU, no filter", 2, d_idx, n, table, bits_all, 0, sink, a, b)) return 1;
  if (run<1024, false>("1024 thr x 1/CU, no filter", 1, d_idx, n, table, bits_all, 0, sink, a, b)) return 1;
  if (run<1024, false>("1024 thr x 2/CU, no filter", 2, d_idx, n, table, bits_all, 0, sink, a, b)) return 1;
  if (run<512, true>("512 thr x 2/CU, 64 KB bitmap (512K ids), all set", 2, d_idx, n, table, bits_all, 512 << 10, sink, a, b)) return 1;
  if (run<1024, true>("1024 thr x 1/CU, 96 KB bitmap (768K ids), all set", 1, d_idx, n, table, bits_all, 768 << 10, sink, a, b)) return 1;
  if (run<1024, true>("1024 thr x 1/CU, 128 KB bitmap (1M ids), all set", 1, d_idx, n, table, bits_all, 1024 << 10, sink, a, b)) return 1;
  if (run<1024, true>("1024 thr x 1/CU, 128 KB bitmap, half set", 1, d_idx, n, table, bits_half, 1024 << 10, sink, a, b)) return 1;
  if (run<256, true>("256 thr x 4/CU, 32 KB bitmap (256K ids), all set", 4, d_idx, n, table, bits_all, 256 << 10, sink, a, b)) return 1;
  return 0;
}
