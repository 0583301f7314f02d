// Issue-rate microbenchmark: plain v_fma_f32 vs v_pk_fma_f32 vs v_mul_f32, all CUs busy, 8 waves/SIMD.
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096;

__global__ void __launch_bounds__(256) k_fma(float* out, float a, float b) {
  float x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < kIters; it++)
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = __builtin_fmaf(x[i], a, b);
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_pkfma(float* out, float a, float b) {
  f32x2 x[4];
  for (int i = 0; i < 4; i++) x[i] = f32x2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
  const f32x2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < kIters; it++)
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = __builtin_elementwise_fma(x[i], av, bv);
  float s = 0;
  for (int i = 0; i < 4; i++) s += x[i].x + x[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_mul(float* out, float a) {
  float x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + i + 1.0f;
  for (int it = 0; it < kIters; it++)
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = x[i] * a;
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_pkmul(float* out, float a) {
  f32x2 x[4];
  for (int i = 0; i < 4; i++) x[i] = f32x2{threadIdx.x * 1e-3f + i + 1.0f, threadIdx.x * 2e-3f + i + 1.0f};
  const f32x2 av = {a, a};
  for (int it = 0; it < kIters; it++)
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = x[i] * av;
  float s = 0;
  for (int i = 0; i < 4; i++) s += x[i].x + x[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> static float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 5; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves/SIMD
  float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
  const double lane_ops = (double)blocks * 256 * kIters * 8;  // scalar element-ops per launch
  float t1 = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f); });
  float t2 = timeit([&] { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f); });
  float t3 = timeit([&] { hipLaunchKernelGGL(k_mul, dim3(blocks), dim3(256), 0, 0, out, 0.9999f); });
  float t4 = timeit([&] { hipLaunchKernelGGL(k_pkmul, dim3(blocks), dim3(256), 0, 0, out, 0.9999f); });
  printf("v_fma_f32    %.3f ms  %.1f Gelem-op/s\n", t1, lane_ops / t1 / 1e6);
  printf("v_pk_fma_f32 %.3f ms  %.1f Gelem-op/s\n", t2, lane_ops / t2 / 1e6);
  printf("v_mul_f32    %.3f ms  %.1f Gelem-op/s\n", t3, lane_ops / t3 / 1e6);
  printf("v_pk_mul_f32 %.3f ms  %.1f Gelem-op/s\n", t4, lane_ops / t4 / 1e6);
  return 0;
}
