// Gather-rate microbenchmark: how many cycles does one wave64 `buffer_load_dword ... idxen` cost a CU, as a function
// of the address pattern?  The GI march issues one such gather per ray-step (2.6 G lane-gathers per pass at 800x800);
// once its VALU work was cut to ~12 instructions per sample the kernel stopped scaling, and this tells which property
// of the address pattern the texture-address / L1 path charges for.
// Patterns over an L2/L1-resident 800x800 fp32 plane (2.56 MB), 8 loads in flight per wave, 8 waves/SIMD:
//   row64      lanes read 64 consecutive pixels of one row                   (1-2 lines)
//   tile8x8    lanes read an 8x8 pixel block                                  (8 rows)
//   tile16x4   16x4 block                                                     (4 rows)
//   scat8x8_N  8x8 block + a per-lane pseudo-random offset within +-N pixels  (what neighbouring pixels' samples do)
//   quad2x2    lanes of a quad read a 2x2 block, quads scattered over 64x64
//   same       all lanes read one address
//   random     lanes read uniformly random pixels of the plane
// build: hipcc -O3 --offload-arch=gfx950 gather_rate.hip -o gather_rate ; run: ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kW = 800, kH = 800, kIters = 512, kInFlight = 8;

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// mode: 0 row64, 1 tile8x8, 2 tile16x4, 3 scat (param = N), 4 quad2x2, 5 same, 6 random, 7 tile32x2, 8 scat on a 16x4 block
// Per-lane offsets are fixed per k (registers); only a wave-uniform base moves per iteration (scalar ALU), so one
// v_add per gather is all the vector work besides the load.
__device__ __forceinline__ int lane_offset(int mode, int param, int lane, unsigned salt) {
  const unsigned hl = hash(salt + (unsigned)lane * 2654435761u);
  int x = 0, y = 0;
  switch (mode) {
    case 0: x = lane; break;
    case 1: x = (lane & 7); y = (lane >> 3); break;
    case 2: x = (lane & 15); y = (lane >> 4); break;
    case 3: x = (lane & 7) + (int)(hl % (2 * param + 1)) - param; y = (lane >> 3) + (int)((hl >> 10) % (2 * param + 1)) - param; break;
    case 4: { const unsigned hq = hash(salt + (unsigned)(lane >> 2) * 40503u);
              x = (int)(hq % 62) + (lane & 1); y = (int)((hq >> 8) % 62) + ((lane >> 1) & 1); } break;
    case 5: break;
    case 6: x = (int)(hl % 600); y = (int)((hl >> 11) % 600); break;
    case 7: x = (lane & 31); y = (lane >> 5); break;
    case 8: x = (lane & 15) + (int)(hl % (2 * param + 1)) - param; y = (lane >> 4) + (int)((hl >> 10) % (2 * param + 1)) - param; break;
  }
  return y * kW + x;
}

template <bool kLoad>
__global__ void __launch_bounds__(256) k_gather_t(const float* plane, float* out, int mode, int param) {
  const int lane = threadIdx.x & 63;
  const unsigned wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(plane), 4, kW * kH, 0x00020000);
  int off[kInFlight];
#pragma unroll
  for (int k = 0; k < kInFlight; k++) off[k] = lane_offset(mode, param, lane, wave_id * 977u + k * 131u);
  float acc = 0.0f;
  unsigned base = (wave_id * 7919u) % (unsigned)(kW * 100);
  for (int it = 0; it < kIters; it++) {
    unsigned idx[kInFlight];
#pragma unroll
    for (int k = 0; k < kInFlight; k++) {
      base += 40 * kW + 37;                                  // scalar: wave-uniform walk over the plane
      if (base >= (unsigned)(kW * (kH - 200))) base -= (unsigned)(kW * (kH - 200));
      idx[k] = base + (unsigned)(64 * kW + 64) + (unsigned)off[k];
    }
    if constexpr (kLoad) {
      float z[kInFlight];
#pragma unroll
      for (int k = 0; k < kInFlight; k++) asm volatile("buffer_load_dword %0, %1, %2, 0 idxen" : "=&v"(z[k]) : "v"(idx[k]), "s"(rsrc));
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]));
#pragma unroll
      for (int k = 0; k < kInFlight; k++) acc += z[k];
    } else {
#pragma unroll
      for (int k = 0; k < kInFlight; k++) acc += __uint_as_float(idx[k] & 0x3fffffffu);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
#define k_gather k_gather_t<true>
#define k_nogather k_gather_t<false>

template <typename F> static float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 3; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 3;
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double clk_ghz = prop.clockRate / 1e6;
  const int blocks = cus * 8;
  float *plane, *out;
  hipMalloc(&plane, kW * kH * sizeof(float)); hipMalloc(&out, blocks * 256 * sizeof(float));
  std::vector<float> h(kW * kH);
  for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 97) * 0.01f;
  hipMemcpy(plane, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  struct { const char* name; int mode, param; } pats[] = {
      {"row64", 0, 0}, {"tile32x2", 7, 0}, {"tile16x4", 2, 0}, {"tile8x8", 1, 0}, {"scat8x8_2", 3, 2}, {"scat8x8_6", 3, 6},
      {"scat8x8_12", 3, 12}, {"scat16x4_2", 8, 2}, {"scat16x4_6", 8, 6}, {"scat16x4_12", 8, 12}, {"quad2x2", 4, 0},
      {"same", 5, 0}, {"random", 6, 0}};
  printf("%d CUs, %.2f GHz (reported), %d waves/SIMD\n", cus, clk_ghz, 8);
  for (auto& p : pats) {
    float tg = timeit([&] { hipLaunchKernelGGL((k_gather_t<true>), dim3(blocks), dim3(256), 0, 0, plane, out, p.mode, p.param); });
    float tn = timeit([&] { hipLaunchKernelGGL((k_gather_t<false>), dim3(blocks), dim3(256), 0, 0, plane, out, p.mode, p.param); });
    const double wave_gathers_per_cu = (double)blocks * 4 * kIters * kInFlight / cus;
    const double clk_total = tg * 1e-3 * clk_ghz * 1e9 / wave_gathers_per_cu;
    const double clk_net = (tg - tn) * 1e-3 * clk_ghz * 1e9 / wave_gathers_per_cu;
    printf("%-12s gather %.3f ms, index-only %.3f ms -> %.1f clk per wave-gather per CU (%.1f net of the index math), %.2f Tlane-gathers/s\n",
           p.name, tg, tn, clk_total, clk_net, (double)blocks * 256 * kIters * kInFlight / (tg * 1e-3) / 1e12);
  }
  return 0;
}
