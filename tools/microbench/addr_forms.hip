// Issue cost of the GI march's block-address computation on gfx950: the shipped form
//   2 x v_cvt_flr_i32_f32, 2 x v_min_u32, v_lshlrev_b32, v_mad_u32_u24                      (6 instructions)
// against a packed form
//   2 x v_cvt_flr_i32_f32, v_cvt_pk_u16_u32, v_pk_min_u16, v_dot2_u32_u16                    (5 instructions)
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/addr_forms.hip -o tools/microbench/addr_forms
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kIters = 2048;
__global__ void __launch_bounds__(256) k_shipped(unsigned* out, float a, unsigned bw, unsigned bh, unsigned row8) {
  float x[8], y[8];
  unsigned acc = 0;
  for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.11f + i; y[i] = threadIdx.x * 0.07f + i; }
  unsigned bwv = bw, bhv = bh;
  asm volatile("" : "+v"(bwv), "+v"(bhv));
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      unsigned bx, by, t;
      asm volatile("v_cvt_flr_i32_f32 %0, %3\n v_cvt_flr_i32_f32 %1, %4\n v_min_u32 %0, %0, %5\n v_min_u32 %1, %1, %6\n"
                   " v_lshlrev_b32 %0, 3, %0\n v_mad_u32_u24 %2, %1, %7, %0"
                   : "=&v"(bx), "=&v"(by), "=v"(t) : "v"(x[i]), "v"(y[i]), "v"(bwv), "v"(bhv), "s"(row8));
      acc ^= t;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_packed(unsigned* out, float a, unsigned bw, unsigned bh, unsigned row8) {
  float x[8], y[8];
  unsigned acc = 0;
  for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.11f + i; y[i] = threadIdx.x * 0.07f + i; }
  unsigned lim = bw | (bh << 16), mul = 8u | (row8 << 16);
  asm volatile("" : "+v"(lim), "+v"(mul));
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      unsigned bx, by, t;
      asm volatile("v_cvt_flr_i32_f32 %0, %3\n v_cvt_flr_i32_f32 %1, %4\n v_cvt_pk_u16_u32 %0, %0, %1\n v_pk_min_u16 %0, %0, %5\n"
                   " v_dot2_u32_u16 %2, %0, %6, 0"
                   : "=&v"(bx), "=&v"(by), "=v"(t) : "v"(x[i]), "v"(y[i]), "v"(lim), "v"(mul));
      acc ^= t;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// the same without the xor that keeps the results alive in k_shipped / k_packed, to price it
__global__ void __launch_bounds__(256) k_xor_only(unsigned* out, float a, unsigned bw, unsigned bh, unsigned row8) {
  unsigned acc = threadIdx.x, t = bw + threadIdx.x;
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(t));
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <typename K> static float timeit(K k, int blocks, unsigned* out) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 0.5f, 50u, 50u, 408u); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 0.5f, 50u, 50u, 408u);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  unsigned* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(unsigned));
  for (int wps : {5, 2}) {
    const int blocks = 256 * wps;
    const double groups = (double)blocks * 4 * kIters * 8;  // address computations per launch (wave-level)
    auto rep = [&](const char* n, float ms) { printf("  %-10s %.3f ms  %.2f cycles per address per SIMD at 2.4 GHz\n", n, ms, ms * 1e-3 * 2.4e9 * 1024 / groups); };
    printf("%d waves/SIMD\n", wps);
    rep("shipped", timeit(k_shipped, blocks, out));
    rep("packed", timeit(k_packed, blocks, out));
    rep("xor only", timeit(k_xor_only, blocks, out));
  }
  // the two forms agree (in range, negative, huge)
  return 0;
}
