// Random 4-byte gather ceiling of one MI355X: G lookups/s for tables of several sizes and the two
// load flavours the advance kernels use (plain = L1-cached, agent-scope relaxed atomic load = sc1).
// Indices come from a streamed int32 array (like the CSR column array), uniformly random or with an
// R-MAT-like skew.  Build: hipcc --offload-arch=gfx950 -O3 -o gather_roof tools/gather_roof.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, typename T>
__global__ void __launch_bounds__(256) gather_kernel(const int32_t* idx, size_t n, const T* table,
                                                      unsigned long long* sink) {
  unsigned long long acc = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += stride) {
    const int32_t j = __builtin_nontemporal_load(idx + i);
    T v;
    if (MODE == 0)
      v = table[j];
    else
      v = __hip_atomic_load(table + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc += (unsigned long long)v;
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

int main() {
  const size_t n = 64u << 20;  // 64 M lookups, 256 MB of indices
  std::vector<int32_t> h(n);
  int32_t* d_idx;
  unsigned long long* sink;
  CK(hipMalloc(&d_idx, n * 4));
  CK(hipMalloc(&sink, 8));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int skew = 0; skew < 2; ++skew)
    for (int lg = 17; lg <= 24; lg += (lg < 22 ? 2 : 1)) {  // table entries 2^lg
      const size_t entries = (size_t)1 << lg;
      for (size_t i = 0; i < n; ++i) {
        uint64_t r = mix(i * 2 + skew);
        if (!skew) {
          h[i] = (int32_t)(r & (entries - 1));
        } else {  // every id bit is 1 with probability 0.24 (R-MAT a+c = 0.76)
          uint32_t v = 0;
          uint64_t r2 = mix(r);
          for (int bit = 0; bit < lg; ++bit) {
            const unsigned byte = (unsigned)((bit < 8 ? r >> (8 * bit) : r2 >> (8 * (bit - 8))) & 255);
            v |= (byte < 61 ? 1u : 0u) << bit;
          }
          h[i] = (int32_t)v;
        }
      }
      CK(hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice));
      for (int width = 4; width <= 8; width += 4) {
        void* table;
        CK(hipMalloc(&table, entries * width));
        CK(hipMemset(table, 1, entries * width));
        for (int mode = 0; mode < 2; ++mode) {
          float best = 1e9f;
          for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(a));
            if (width == 4) {
              if (mode == 0) gather_kernel<0, uint32_t><<<256 * 8, 256>>>(d_idx, n, (uint32_t*)table, sink);
              else gather_kernel<1, uint32_t><<<256 * 8, 256>>>(d_idx, n, (uint32_t*)table, sink);
            } else {
              if (mode == 0) gather_kernel<0, unsigned long long><<<256 * 8, 256>>>(d_idx, n, (unsigned long long*)table, sink);
              else gather_kernel<1, unsigned long long><<<256 * 8, 256>>>(d_idx, n, (unsigned long long*)table, sink);
            }
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
          }
          printf("%s idx, table %7.2f MB (%d-B entries), %s load: %7.3f ms = %6.1f G lookups/s\n",
                 skew ? "skewed " : "uniform", entries * width / 1048576.0, width,
                 mode ? "sc1  " : "plain", best, n / best / 1e6);
          fflush(stdout);
        }
        CK(hipFree(table));
      }
    }
  return 0;
}
