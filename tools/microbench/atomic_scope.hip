// Scattered fp32 atomic adds (3 consecutive floats per lane, random rows) at agent scope vs workgroup scope,
// into ONE buffer and into per-XCD private buffers (XCC id from the hardware register).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kRows = 6 * 256 * 256;  // rows of 3 floats (the 256^2 level of the light)
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
template <int kScope, bool kPerXcd>
__global__ void __launch_bounds__(256) k(float* buf, int iters, unsigned seed) {
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + seed;
  float* base = kPerXcd ? buf + (size_t)xcc_id() * kRows * 3 : buf;
  for (int i = 0; i < iters; i++) {
    s = s * 1664525u + 1013904223u;
    float* p = base + (size_t)(s % kRows) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) __hip_atomic_fetch_add(p + c, 1.0f, __ATOMIC_RELAXED, kScope);
  }
}
template <typename K> static float run(K kern, float* buf, size_t bytes, double* total) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipMemset(buf, 0, bytes);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, buf, 16, 12345u);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  std::vector<float> h(bytes / 4);
  (void)hipMemcpy(h.data(), buf, bytes, hipMemcpyDeviceToHost);
  double t = 0; for (float v : h) t += v;
  *total = t;
  return ms;
}
int main() {
  const size_t one = (size_t)kRows * 3 * 4;
  float* buf; (void)hipMalloc(&buf, one * 8);
  const double expect = 2048.0 * 256 * 16 * 3;
  double t; float ms;
  ms = run(k<__HIP_MEMORY_SCOPE_AGENT, false>, buf, one, &t);      printf("agent scope, one buffer      %.3f ms  sum %.0f / %.0f\n", ms, t, expect);
  ms = run(k<__HIP_MEMORY_SCOPE_WORKGROUP, false>, buf, one, &t);  printf("workgroup scope, one buffer  %.3f ms  sum %.0f / %.0f\n", ms, t, expect);
  ms = run(k<__HIP_MEMORY_SCOPE_AGENT, true>, buf, one * 8, &t);   printf("agent scope, per-XCD buffers %.3f ms  sum %.0f / %.0f\n", ms, t, expect);
  ms = run(k<__HIP_MEMORY_SCOPE_WORKGROUP, true>, buf, one * 8, &t); printf("workgroup scope, per-XCD     %.3f ms  sum %.0f / %.0f\n", ms, t, expect);
  return 0;
}
