// Issue rate of v_pk_*_f32 operand forms (op_sel broadcast, neg modifiers, SGPR source) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 2048;
#define KERNEL(NAME, ASM)                                                                     \
  __global__ void __launch_bounds__(256) NAME(float* out, float a) {                          \
    f32x2 x[8];                                                                               \
    for (int i = 0; i < 8; i++) x[i] = f32x2{threadIdx.x * 1e-3f + i + 1.0f, threadIdx.x * 2e-3f + i + 1.0f}; \
    f32x2 y = {a, a * 1.0001f};                                                               \
    for (int it = 0; it < kIters; it++) {                                                     \
      _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(x[i]) : "v"(y)); \
    }                                                                                         \
    float s = 0;                                                                              \
    for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;                                         \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                  \
  }
KERNEL(k_mul_plain, "v_pk_mul_f32 %0, %0, %1")
KERNEL(k_mul_bcast, "v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]")
KERNEL(k_mul_swap, "v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]")
KERNEL(k_fma_plain, "v_pk_fma_f32 %0, %0, %1, %1")
KERNEL(k_fma_neg, "v_pk_fma_f32 %0, %0, %1, %1 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")
KERNEL(k_add_plain, "v_pk_add_f32 %0, %0, %1")
// dependent chain of 8 packed multiplies on ONE accumulator (latency-bound unless other waves fill in)
__global__ void __launch_bounds__(256) k_mul_chain(float* out, float a) {
  f32x2 x = {threadIdx.x * 1e-3f + 1.0f, threadIdx.x * 2e-3f + 1.0f}, y = {a, a * 1.0001f};
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  }
  out[blockIdx.x * 256 + threadIdx.x] = x.x + x.y;
}
__global__ void __launch_bounds__(256) k_smul_chain(float* out, float a) {
  float x = threadIdx.x * 1e-3f + 1.0f;
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
  }
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
// plain multiplies, 8 independent accumulators: VGPR vs SGPR second source
__global__ void __launch_bounds__(256) k_smul_v(float* out, float a) {
  float x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + i + 1.0f;
  float av = a + threadIdx.x * 0.0f;
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "v"(av));
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_smul_s(float* out, float a) {
  float x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + i + 1.0f;
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "s"(a));
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_pkmul_s(float* out, float a) {
  f32x2 x[8];
  for (int i = 0; i < 8; i++) x[i] = f32x2{threadIdx.x * 1e-3f + i + 1.0f, threadIdx.x * 2e-3f + i + 1.0f};
  f32x2 y = {a, a};
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(x[i]) : "s"(y));
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_fma3(float* out, float a) {  // VOP3-encoded plain fma, VGPR sources
  float x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + i + 1.0f;
  float av = a + threadIdx.x * 0.0f;
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(av));
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_cmp(float* out, float a) {  // v_cmp to SGPR pair + v_cndmask
  float x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + i + 1.0f;
  float av = a + threadIdx.x * 0.0f;
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(av) : "vcc");
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_cvt(float* out, float a) {
  float x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + i + 1.0f;
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_f32_i32 %0, %0" : "+v"(x[i]));
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K> static float timeit(K k, int blocks, float* out) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 0.99999f); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 0.99999f);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  for (int wps : {6, 2}) {  // waves per SIMD
    const int blocks = 256 * wps;
    const double instrs = (double)blocks * 4 * kIters * 8;  // wave-instructions per launch
    auto rep = [&](const char* n, float ms) { printf("  %-12s %.3f ms  %.2f cycles/instr/SIMD at 2.4 GHz\n", n, ms, ms * 1e-3 * 2.4e9 * 1024 / instrs); };
    printf("%d waves/SIMD\n", wps);
    rep("pk_mul", timeit(k_mul_plain, blocks, out));
    rep("pk_mul bcast", timeit(k_mul_bcast, blocks, out));
    rep("pk_mul swap", timeit(k_mul_swap, blocks, out));
    rep("pk_fma", timeit(k_fma_plain, blocks, out));
    rep("pk_fma neg", timeit(k_fma_neg, blocks, out));
    rep("pk_add", timeit(k_add_plain, blocks, out));
    rep("pk_mul chain", timeit(k_mul_chain, blocks, out));
    rep("mul chain", timeit(k_smul_chain, blocks, out));
    rep("mul vgpr", timeit(k_smul_v, blocks, out));
    rep("mul sgpr", timeit(k_smul_s, blocks, out));
    rep("pk_mul sgpr", timeit(k_pkmul_s, blocks, out));
    rep("fma vop3", timeit(k_fma3, blocks, out));
    rep("cmp+cndmask/2", timeit(k_cmp, blocks, out) / 2);
    rep("cvt x2 /2", timeit(k_cvt, blocks, out) / 2);
  }
  return 0;
}
