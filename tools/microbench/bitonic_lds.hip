// Why does the per-tile LDS bitonic sort of binning.hip take ~2.3 us per barrier phase?  One workgroup per CU sorts
// 8192 random u64 keys in LDS; variants isolate the barrier, the padding, the fused register phases and the thread count.
// build: hipcc -O3 --offload-arch=gfx950 bitonic_lds.hip -o bitonic_lds ; run: ./bitonic_lds
#include <hip/hip_runtime.h>
#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/block/block_load.hpp>
#include <rocprim/block/block_store.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

template <bool kPad> __device__ __forceinline__ unsigned slot(unsigned i) { return kPad ? i + (i >> 4) : i; }

// fused: 3 substages per phase on 8 keys in registers (binning.hip::bitonic_sort_lds)
template <int kThreads, bool kPad, bool kBarrier>
__device__ __forceinline__ void sort_fused(uint64_t* s, unsigned npad_log2) {
  const unsigned npad = 1u << npad_log2;
  for (unsigned kb = 1; kb <= npad_log2; kb++) {
    int jb = (int)kb - 1;
    while (jb >= 0) {
      const int g = min(3, jb + 1);
      const int lowpos = jb - g + 1;
      const unsigned m = 1u << lowpos, cnt = 1u << g;
      const unsigned items = npad >> g;
      for (unsigned t = threadIdx.x; t < items; t += kThreads) {
        const unsigned base = ((t >> lowpos) << (lowpos + g)) | (t & (m - 1));
        const bool asc = ((base >> kb) & 1u) == 0;
        uint64_t v[8];
#pragma unroll
        for (unsigned e = 0; e < 8; e++) if (e < cnt) v[e] = s[slot<kPad>(base + e * m)];
#pragma unroll
        for (int sft = 2; sft >= 0; sft--) if (sft < g) {
#pragma unroll
          for (unsigned e = 0; e < 8; e++) if (!(e & (1u << sft)) && (e | (1u << sft)) < cnt) {
            const uint64_t a = v[e], b = v[e | (1u << sft)];
            const bool sw = asc ? (a > b) : (a < b);
            v[e] = sw ? b : a; v[e | (1u << sft)] = sw ? a : b;
          }
        }
#pragma unroll
        for (unsigned e = 0; e < 8; e++) if (e < cnt) s[slot<kPad>(base + e * m)] = v[e];
      }
      if (kBarrier) __syncthreads();
      jb -= g;
    }
  }
}

// plain: one substage per barrier, one compare-exchange per thread iteration
template <int kThreads, bool kBarrier>
__device__ __forceinline__ void sort_plain(uint64_t* s, unsigned npad_log2) {
  const unsigned half = (1u << npad_log2) >> 1;
  for (unsigned kb = 1; kb <= npad_log2; kb++)
    for (int jb = (int)kb - 1; jb >= 0; jb--) {
      const unsigned j = 1u << jb;
      for (unsigned i = threadIdx.x; i < half; i += kThreads) {
        const unsigned a = ((i >> jb) << (jb + 1)) + (i & (j - 1)), b = a + j;
        const bool asc = ((a >> kb) & 1u) == 0;
        const uint64_t x = s[a], y = s[b];
        if (asc ? (x > y) : (x < y)) { s[a] = y; s[b] = x; }
      }
      if (kBarrier) __syncthreads();
    }
}

template <int kThreads, int kVariant>
__global__ void __launch_bounds__(kThreads) k_sort(const uint64_t* in, uint64_t* out, unsigned n_log2, int reps) {
  extern __shared__ uint64_t s[];
  const unsigned n = 1u << n_log2;
  for (int r = 0; r < reps; r++) {
    for (unsigned i = threadIdx.x; i < n; i += kThreads) s[slot<kVariant == 0>(i)] = in[(size_t)blockIdx.x * n + i] + r;
    __syncthreads();
    if (kVariant == 0) sort_fused<kThreads, true, true>(s, n_log2);
    if (kVariant == 1) sort_fused<kThreads, false, true>(s, n_log2);
    if (kVariant == 2) sort_fused<kThreads, false, false>(s, n_log2);  // wrong result: no barriers (cost of the barriers)
    if (kVariant == 3) sort_plain<kThreads, true>(s, n_log2);
    if (kVariant == 4) sort_plain<kThreads, false>(s, n_log2);
    __syncthreads();
    for (unsigned i = threadIdx.x; i < n; i += kThreads) out[(size_t)blockIdx.x * n + i] = s[slot<kVariant == 0>(i)];
  }
}

// rocPRIM block radix sort on bits [0, end_bit): kThreads x kItems keys in registers
template <int kThreads, int kItems, int kBits>
__global__ void __launch_bounds__(kThreads) k_radix(const uint64_t* in, uint64_t* out, unsigned end_bit, int reps) {
  using sorter = rocprim::block_radix_sort<uint64_t, kThreads, kItems, rocprim::empty_type, 1, 1, kBits>;
  __shared__ typename sorter::storage_type storage;
  const size_t base = (size_t)blockIdx.x * kThreads * kItems;
  for (int r = 0; r < reps; r++) {
    uint64_t k[kItems];
#pragma unroll
    for (int i = 0; i < kItems; i++) k[i] = (in[base + threadIdx.x * kItems + i] + r) & ((1ull << end_bit) - 1);
    sorter().sort(k, storage, 0, end_bit);
#pragma unroll
    for (int i = 0; i < kItems; i++) out[base + threadIdx.x * kItems + i] = k[i];
    __syncthreads();
  }
}
template <int kThreads, int kItems, int kBits>
static void run_radix(const char* name, const uint64_t* in, uint64_t* out, unsigned end_bit, int blocks);

template <typename F> static float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

template <int kThreads, int kVariant>
static void run(const char* name, const uint64_t* in, uint64_t* out, unsigned n_log2, int blocks) {
  const size_t lds = ((size_t)(1u << n_log2) * 17 / 16 + 16) * 8;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort<kThreads, kVariant>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int reps = 20;
  float ms = timeit([&] { hipLaunchKernelGGL((k_sort<kThreads, kVariant>), dim3(blocks), dim3(kThreads), lds, 0, in, out, n_log2, reps); });
  printf("%-34s n=%5u threads=%4d blocks=%3d: %.1f us per sort\n", name, 1u << n_log2, kThreads, blocks, 1e3 * ms / reps);
}

template <int kThreads, int kItems, int kBits>
static void run_radix(const char* name, const uint64_t* in, uint64_t* out, unsigned end_bit, int blocks) {
  const int reps = 20;
  float ms = timeit([&] { hipLaunchKernelGGL((k_radix<kThreads, kItems, kBits>), dim3(blocks), dim3(kThreads), 0, 0, in, out, end_bit, reps); });
  printf("%-34s n=%5d threads=%4d blocks=%3d: %.1f us per sort (bits %u, %d per pass)\n", name, kThreads * kItems, kThreads, blocks,
         1e3 * ms / reps, end_bit, kBits);
}

int main() {
  const int blocks = 256;
  const unsigned max_n = 8192;
  std::vector<uint64_t> h((size_t)blocks * max_n);
  uint64_t x = 88172645463325252ull;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x; }
  uint64_t *in, *out;
  hipMalloc(&in, h.size() * 8); hipMalloc(&out, h.size() * 8);
  hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  for (int b : {1, 256}) {
    run<1024, 0>("fused padded", in, out, 13, b);
    run<1024, 1>("fused unpadded", in, out, 13, b);
    run<1024, 2>("fused unpadded NO barriers", in, out, 13, b);
    run<1024, 3>("plain", in, out, 13, b);
    run<1024, 4>("plain NO barriers", in, out, 13, b);
    run<512, 0>("fused padded", in, out, 12, b);
    run<256, 0>("fused padded", in, out, 12, b);
    run<256, 0>("fused padded", in, out, 10, b);
  }
  for (int b : {1, 256}) {
    run_radix<1024, 8, 4>("rocprim block_radix_sort", in, out, 51, b);
    run_radix<1024, 8, 8>("rocprim block_radix_sort", in, out, 51, b);
    run_radix<512, 16, 8>("rocprim block_radix_sort", in, out, 51, b);
    run_radix<256, 16, 8>("rocprim block_radix_sort", in, out, 51, b);
    run_radix<256, 16, 4>("rocprim block_radix_sort", in, out, 51, b);
    run_radix<256, 4, 8>("rocprim block_radix_sort", in, out, 51, b);
    run_radix<256, 4, 4>("rocprim block_radix_sort", in, out, 51, b);
    run_radix<64, 4, 8>("rocprim block_radix_sort", in, out, 51, b);
  }
  // correctness of variant 0
  std::vector<uint64_t> o(max_n);
  hipLaunchKernelGGL((k_sort<1024, 0>), dim3(1), dim3(1024), (size_t)(max_n * 17 / 16 + 16) * 8, 0, in, out, 13u, 1);
  hipMemcpy(o.data(), out, max_n * 8, hipMemcpyDeviceToHost);
  printf("sorted: %s\n", std::is_sorted(o.begin(), o.end()) ? "yes" : "NO");
  return 0;
}
