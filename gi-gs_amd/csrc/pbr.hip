// pbr.hip -- deferred split-sum shade of the G-buffer and the cubemap light filters.
//
// Reference behaviour restated:
//   pbr_shading                          pbr/shade.py:108-241
//   CubemapLight.get_mip / cubemap_mip   pbr/light.py:142-152, 54-79
//   DiffuseCubemapFwd/BwdKernel          pbr/renderutils/c_src/cubemap.cu:110-169  ("RU/")
//   SpecularBoundsKernel                 RU/cubemap.cu:181-244
//   SpecularCubemapFwd/BwdKernel         RU/cubemap.cu:246-350
// The three texture lookups of the shade are nvdiffrast `dr.texture` calls in the reference
// (third party, unpinned -> parity unpinned); the sampling rule implemented here is written
// down in include/gigs_hip.h and oracle/pbr_oracle.cpp.
//
// MI355X design
//   * shade forward/backward are ONE fused kernel each, one lane per pixel: the reference runs
//     ~40 small torch kernels + 3 texture ops over the same 640k pixels; fused, the pass reads
//     each G-buffer plane once (~80 B/pixel) and is bound by the gathers into the light
//     textures, which are small (diffuse 18 KB, specular mips 6.3 MB) and stay in L2;
//   * gradients of the light textures: the 16x16x6 diffuse map is accumulated per workgroup in
//     LDS (ds_add_f32, 18 KB) and flushed once, the specular mips take global float atomics;
//   * the cubemap-filter backward passes are GATHERS, not the reference's atomic scatters: the
//     GGX window test dot(L, V) >= cutoff is symmetric in (L, V), so the set of outputs that
//     touch a texel is that texel's own window; same terms, no atomics, reproducible sums;
//   * pixel-invariant tables (solid angle per texel) are built once on the host with libm and
//     cached per device.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "gigs_common.h"
#include "pixel_ops.h"

namespace gigs {

// ------------------------------------------------------------------------------------------
// helpers shared by all kernels
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ v3 safe_normalize(v3 v) {  // RU/vec3f.h:90-94
  const float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
  return l > 0.0f ? v3{v.x / l, v.y / l, v.z / l} : v3{0, 0, 0};
}
__device__ __forceinline__ v3 cube_dir_raw(float fx, float fy, int side) {
  switch (side) {
    case 0: return {1, -fy, -fx};
    case 1: return {-1, -fy, fx};
    case 2: return {fx, 1, fy};
    case 3: return {fx, -1, -fy};
    case 4: return {fx, -fy, 1};
    default: return {-fx, -fy, -1};
  }
}
__device__ __forceinline__ v3 cube_to_dir(int x, int y, int side, int N) {  // RU/cubemap.cu:33-47
  const float fx = 2.0f * (((float)x + 0.5f) / (float)N) - 1.0f;
  const float fy = 2.0f * (((float)y + 0.5f) / (float)N) - 1.0f;
  return safe_normalize(cube_dir_raw(fx, fy, side));
}
__device__ __forceinline__ float ndf_ggx(float alphaSqr, float cosTheta) {  // RU/cubemap.cu:174-179
  const float c = fminf(fmaxf(cosTheta, 0.0f), 1.0f);
  const float d = (c * alphaSqr - c) * c + 1.0f;
  return (float)((double)alphaSqr / ((double)(d * d) * 3.14159265358979323846));
}

// ---- per-resolution texel table: float4 (unit direction of the texel centre, solid angle) ----
// direction = cube_to_dir (RU/cubemap.cu:33-47, computed on the device once per resolution with
// the same IEEE sequence as the oracle); solid angle = pixel_area (RU/cubemap.cu:17-31) from a
// 1-D table built on the HOST with libm atanf: area[i] = atan((i+1)/H) - atan(i/H), i = |x - H|.
// Both are pixel- and iteration-invariant, so the filters read one 16-byte entry per texel
// instead of redoing 5 divisions, a sqrt and 4 atanf per (output, input) pair.
__global__ void __launch_bounds__(256)
texel_table_kernel(int N, const float* __restrict__ area, float4* __restrict__ table) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= 6 * N * N) return;
  const int pz = o / (N * N), py = (o / N) % N, px = o % N;
  const v3 d = cube_to_dir(px, py, pz, N);
  float a = 1.0f;
  if (N > 1) {
    const int Hh = N / 2;
    a = area[abs(px - Hh)] * area[abs(py - Hh)];
  }
  table[o] = make_float4(d.x, d.y, d.z, a);
}

struct TexelTable { float4* dev = nullptr; };
static std::mutex g_table_mu;
static std::map<long long, TexelTable> g_table;

static const float4* texel_table(int N, hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_table_mu);
  const long long key = ((long long)dev << 32) | (unsigned)N;
  auto it = g_table.find(key);
  if (it != g_table.end()) return it->second.dev;
  const int Hh = N / 2;
  std::vector<float> h(Hh + 2, 1.0f);
  if (N > 1)
    for (int i = 0; i <= Hh; i++) h[i] = atanf((float)(i + 1) / (float)Hh) - atanf((float)i / (float)Hh);
  float* area = nullptr;
  TexelTable t;
  // one-time per (device, resolution); kept for the lifetime of the process
  if (hipMalloc(&area, h.size() * sizeof(float)) != hipSuccess) return nullptr;
  if (hipMalloc(&t.dev, (size_t)6 * N * N * sizeof(float4)) != hipSuccess) return nullptr;
  if (hipMemcpyAsync(area, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) return nullptr;
  hipLaunchKernelGGL(texel_table_kernel, dim3((6 * N * N + 255) / 256), dim3(256), 0, s, N, area, t.dev);
  if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
  hipFree(area);
  g_table[key] = t;
  return t.dev;
}

// Sum over the 64 lanes of a wave (DPP, no LDS); total valid in lane 63.
template <int kCtrl, int kRowMask = 0xf>
__device__ __forceinline__ float pbr_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, kRowMask, 0xf, false));
}
__device__ __forceinline__ float wave_sum63(float v) {
  v += pbr_dpp<0xb1>(v);
  v += pbr_dpp<0x4e>(v);
  v += pbr_dpp<0x124>(v);
  v += pbr_dpp<0x128>(v);
  v += pbr_dpp<0x142, 0xa>(v);
  v += pbr_dpp<0x143, 0xc>(v);
  return v;
}

// ------------------------------------------------------------------------------------------
// cubemap filters: ONE WAVE PER OUTPUT TEXEL (4 per 256-lane workgroup).  The 64 lanes sweep
// the texel's window as an 8x8 patch, so neighbouring lanes read neighbouring 16-byte table /
// texture entries, and the four partial sums are combined with DPP.  Even the 16x16 level
// (1536 outputs) then fills the chip with 1536 waves instead of 24.
// ------------------------------------------------------------------------------------------
// forward: out[o] = sum_in tex[in] * w(o, in);  backward (gather): g_in[i] = sum_o g[o] * w(o, i)
template <bool kBackward>
__global__ void __launch_bounds__(256)
diffuse_cubemap_kernel(int N, const float4* __restrict__ table, const float* __restrict__ src,
                       float* __restrict__ dst) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= 6 * N * N) return;  // wave-uniform
  const float4 me = table[o];
  float c0 = 0, c1 = 0, c2 = 0;
  for (int i = lane; i < 6 * N * N; i += 64) {
    const float4 ot = table[i];
    // forward: N = me, L = other, area of L; backward: N = other, L = me, area of me
    const float d = kBackward ? (ot.x * me.x + ot.y * me.y + ot.z * me.z) : (me.x * ot.x + me.y * ot.y + me.z * ot.z);
    const float costheta = fminf(fmaxf(d, 0.0f), 0.999f);
    const float w = costheta * (kBackward ? me.w : ot.w) / 3.141592f;
    const float* t = src + 3 * (size_t)i;
    c0 += t[0] * w; c1 += t[1] * w; c2 += t[2] * w;
  }
  c0 = wave_sum63(c0); c1 = wave_sum63(c1); c2 = wave_sum63(c2);
  if (lane == 63) { dst[3 * (size_t)o] = c0; dst[3 * (size_t)o + 1] = c1; dst[3 * (size_t)o + 2] = c2; }
}


__global__ void __launch_bounds__(64)
specular_bounds_kernel(int N, float cutoff, float* __restrict__ bounds) {
  const int o = blockIdx.x * 64 + threadIdx.x;
  if (o >= 6 * N * N) return;
  const int pz = o / (N * N), py = (o / N) % N, px = o % N;
  const v3 VNR = cube_to_dir(px, py, pz, N);
  const int TILE = 16;
  for (int s = 0; s < 6; ++s) {
    int minx = N - 1, maxx = 0, miny = N - 1, maxy = 0;
    for (int tx = 0; tx < (N + TILE - 1) / TILE; tx++)
      for (int ty = 0; ty < (N + TILE - 1) / TILE; ty++) {
        const int tsx = tx * TILE, tsy = ty * TILE;
        const int tex = min((tx + 1) * TILE, N), tey = min((ty + 1) * TILE, N);
        const v3 L0 = cube_to_dir(tsx, tsy, s, N), L1 = cube_to_dir(tex, tsy, s, N);
        const v3 L2 = cube_to_dir(tsx, tey, s, N), L3 = cube_to_dir(tex, tey, s, N);
        const float mnx = fminf(fminf(L0.x, L1.x), fminf(L2.x, L3.x)), mxx = fmaxf(fmaxf(L0.x, L1.x), fmaxf(L2.x, L3.x));
        const float mny = fminf(fminf(L0.y, L1.y), fminf(L2.y, L3.y)), mxy = fmaxf(fmaxf(L0.y, L1.y), fmaxf(L2.y, L3.y));
        const float mnz = fminf(fminf(L0.z, L1.z), fminf(L2.z, L3.z)), mxz = fmaxf(fmaxf(L0.z, L1.z), fmaxf(L2.z, L3.z));
        const float maxdp = fmaxf(mnx * VNR.x, mxx * VNR.x) + fmaxf(mny * VNR.y, mxy * VNR.y) + fmaxf(mnz * VNR.z, mxz * VNR.z);
        if (maxdp >= cutoff) {
          for (int y = tsy; y < tey; ++y)
            for (int x = tsx; x < tex; ++x) {
              const v3 L = cube_to_dir(x, y, s, N);
              if (dot3(L, VNR) >= cutoff) {
                minx = min(minx, x); maxx = max(maxx, x);
                miny = min(miny, y); maxy = max(maxy, y);
              }
            }
        }
      }
    float* b = bounds + 24 * (size_t)o + s * 4;
    b[0] = (float)minx; b[1] = (float)maxx; b[2] = (float)miny; b[3] = (float)maxy;
  }
}

// safeNormalize (RU/vec3f.h:90-94) for |v| in (2^-60, 2^60): the three IEEE quotients v / l share one
// reciprocal refinement (same FMA chain as the compiler's division expansion, bit-identical results).
__device__ __forceinline__ v3 normalize_exact(v3 v) {
  const float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
  if (!(l > 0x1p-60f && l < 0x1p60f)) return safe_normalize(v);
  const float r0 = __builtin_amdgcn_rcpf(l);
  const float e0 = __builtin_fmaf(-l, r0, 1.0f);
  const float r1 = __builtin_fmaf(e0, r0, r0);
  v3 q;
  {
    const float q0 = v.x * r1, e1 = __builtin_fmaf(-l, q0, v.x), q1 = __builtin_fmaf(e1, r1, q0);
    q.x = __builtin_fmaf(__builtin_fmaf(-l, q1, v.x), r1, q1);
  }
  {
    const float q0 = v.y * r1, e1 = __builtin_fmaf(-l, q0, v.y), q1 = __builtin_fmaf(e1, r1, q0);
    q.y = __builtin_fmaf(__builtin_fmaf(-l, q1, v.y), r1, q1);
  }
  {
    const float q0 = v.z * r1, e1 = __builtin_fmaf(-l, q0, v.z), q1 = __builtin_fmaf(e1, r1, q0);
    q.z = __builtin_fmaf(__builtin_fmaf(-l, q1, v.z), r1, q1);
  }
  return q;
}

// forward: out[o] = (sum_in tex[in] w, sum w) over in in window(o)
// backward (gather over the same, symmetric, window): g_in[i] = sum_o g[o].rgb * w(o, i)
// forward: out[o] = (sum_in tex[in] w, sum w) over in in window(o)
// backward (gather over the same window, which is symmetric because the test dot(L, V) >= cutoff
// is): g_in[i] = sum_o g[o].rgb * w(o, i).
// Per accepted pair H = safeNormalize(L + V), c = V.H and d = (c a^2 - c) c + 1 are the reference's
// own IEEE sequence (RU/cubemap.cu:174-179, 270-277) on the cached unit directions.  This matters: d = 1 - c^2 (1 - alpha^2) cancels to ~alpha^2 at the lobe centre, so a 1-ulp change of
// c = V.H (e.g. the algebraic shortcut sqrt((1 + L.V) / 2)) moves a weight by ~1e-7 / alpha^2 --
// 2e-3 at the roughness-0.08 level.
template <bool kBackward>
__global__ void __launch_bounds__(256)
specular_cubemap_kernel(int N, const float4* __restrict__ table, const float* __restrict__ src,
                        const float* __restrict__ bounds, float roughness, float cutoff,
                        float* __restrict__ dst) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= 6 * N * N) return;  // wave-uniform
  const int lx = lane & 7, ly = lane >> 3;
  const float4 me = table[o];
  const float alpha = roughness * roughness, alphaSqr = alpha * alpha;
  float wsum = 0.0f, c0 = 0, c1 = 0, c2 = 0;
  const int stride = kBackward ? 4 : 3;
  const float4* b4 = reinterpret_cast<const float4*>(bounds + 24 * (size_t)o);
  for (int s = 0; s < 6; ++s) {
    const float4 b = b4[s];
    const int xmin = (int)b.x, xmax = (int)b.y, ymin = (int)b.z, ymax = (int)b.w;
    if (xmin > xmax) continue;
    for (int y = ymin + ly; y <= ymax; y += 8)
      for (int x = xmin + lx; x <= xmax; x += 8) {
        const int i = (s * N + y) * N + x;
        const float4 ot = table[i];
        const float d = kBackward ? (me.x * ot.x + me.y * ot.y + me.z * ot.z) : (ot.x * me.x + ot.y * me.y + ot.z * me.z);
        if (d >= cutoff) {
          const float wiDotN = fmaxf(d, 0.0f);
          // forward: VNR = me, L = other; backward: VNR = other, L = me
          const v3 Hh = kBackward ? normalize_exact(v3{me.x + ot.x, me.y + ot.y, me.z + ot.z})
                                  : normalize_exact(v3{ot.x + me.x, ot.y + me.y, ot.z + me.z});
          const float VNRDotH = fmaxf(kBackward ? (ot.x * Hh.x + ot.y * Hh.y + ot.z * Hh.z)
                                                : (me.x * Hh.x + me.y * Hh.y + me.z * Hh.z), 0.0f);
          // c and dd exactly as the reference; only the last, well-conditioned division
          // alphaSqr / (dd^2 * pi) is done in fp32 instead of fp64 (<= 2e-7 relative)
          const float c = fminf(fmaxf(VNRDotH, 0.0f), 1.0f);
          const float dd = (c * alphaSqr - c) * c + 1.0f;
          const float ndf = alphaSqr / ((dd * dd) * 3.14159265358979323846f);
          const float w = wiDotN * ndf * (kBackward ? me.w : ot.w) / 4.0f;
          const float* t = src + (size_t)stride * i;
          c0 += t[0] * w; c1 += t[1] * w; c2 += t[2] * w;
          wsum += w;
        }
      }
  }
  c0 = wave_sum63(c0); c1 = wave_sum63(c1); c2 = wave_sum63(c2);
  if (!kBackward) wsum = wave_sum63(wsum);
  if (lane == 63) {
    if (kBackward) {
      float* q = dst + 3 * (size_t)o;
      q[0] = c0; q[1] = c1; q[2] = c2;
    } else {
      float* q = dst + 4 * (size_t)o;
      q[0] = c0; q[1] = c1; q[2] = c2; q[3] = wsum;
    }
  }
}


// ------------------------------------------------------------------------------------------
// GGX pre-filter through a cached weight table.
// The weight of a (texel, texel) pair, f = max(L.V, 0) * D_ggx(V.H), depends only on the resolution,
// the roughness and the cutoff -- not on the cubemap values that change every training step.  With
// 288 GB of HBM it is cheaper to keep f for every candidate of every window (186 M floats = 0.74 GB
// for the 256..16 chain, twice that with the role-swapped table the backward needs) than to redo the
// normalisation, the NDF and its divisions each step: the filter becomes one streaming pass
// (4 B of weight + an L2-resident texel per candidate) instead of a VALU-bound one.
// Table layout: for (texel o, face s) the entries [offsets[6o+s], +w*h) hold f row-major over the
// face's AABB, or -1 for candidates outside the cone.  kSwap = false: roles (VNR = o, L = in) as the
// forward uses them; kSwap = true: (VNR = other, L = o) as the gather-form backward needs them, so
// that every product is bit-identical to the table-free kernels above.
// ------------------------------------------------------------------------------------------
template <bool kSwap>
__global__ void __launch_bounds__(256)
specular_weights_kernel(int N, const float4* __restrict__ table, const float* __restrict__ bounds,
                        const uint32_t* __restrict__ offsets, float roughness, float cutoff,
                        float* __restrict__ W) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= 6 * N * N) return;
  const float4 me = table[o];
  const float alpha = roughness * roughness, alphaSqr = alpha * alpha;
  const float4* b4 = reinterpret_cast<const float4*>(bounds + 24 * (size_t)o);
  for (int s = 0; s < 6; ++s) {
    const float4 b = b4[s];
    const int xmin = (int)b.x, xmax = (int)b.y, ymin = (int)b.z, ymax = (int)b.w;
    if (xmin > xmax || ymin > ymax) continue;
    const int wd = xmax - xmin + 1, n = wd * (ymax - ymin + 1);
    const float inv = 1.0f / (float)wd;
    const uint32_t base = offsets[6 * (size_t)o + s];
    for (int i = lane; i < n; i += 64) {
      const int yy = (int)(((float)i + 0.5f) * inv), xx = i - yy * wd;
      const float4 ot = table[(s * N + ymin + yy) * N + xmin + xx];
      const float d = kSwap ? (me.x * ot.x + me.y * ot.y + me.z * ot.z) : (ot.x * me.x + ot.y * me.y + ot.z * me.z);
      float f = -1.0f;
      if (d >= cutoff) {
        const float wiDotN = fmaxf(d, 0.0f);
        const v3 Hh = kSwap ? normalize_exact(v3{me.x + ot.x, me.y + ot.y, me.z + ot.z})
                            : normalize_exact(v3{ot.x + me.x, ot.y + me.y, ot.z + me.z});
        const float VNRDotH = fmaxf(kSwap ? (ot.x * Hh.x + ot.y * Hh.y + ot.z * Hh.z)
                                          : (me.x * Hh.x + me.y * Hh.y + me.z * Hh.z), 0.0f);
        const float c = fminf(fmaxf(VNRDotH, 0.0f), 1.0f);
        const float dd = (c * alphaSqr - c) * c + 1.0f;
        // the full pair weight, ((wiDotN * ndf) * area) / 4 as in RU/cubemap.cu:276: area of the input
        // texel in the forward table, of the gathering texel in the role-swapped (backward) table
        f = wiDotN * (alphaSqr / ((dd * dd) * 3.14159265358979323846f)) * (kSwap ? me.w : ot.w) / 4.0f;
      }
      W[(size_t)base + i] = f;
    }
  }
}

// W2[o][i] = W[o][i] / divisor[texel i of o's window] (markers < 0 kept).  With divisor = the forward's weight sums this
// folds the d(rgb / wsum) division of the normalised filter's backward into its (cached) table: the backward then gathers
// the incoming gradient directly, one launch instead of a division pass + a gather.
__global__ void __launch_bounds__(256)
specular_divide_weights_kernel(int N, const float* __restrict__ bounds, const uint32_t* __restrict__ offsets,
                               const float* __restrict__ W, const float* __restrict__ divisor, float* __restrict__ W2) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= 6 * N * N) return;
  const float4* b4 = reinterpret_cast<const float4*>(bounds + 24 * (size_t)o);
  for (int s = 0; s < 6; ++s) {
    const float4 b = b4[s];
    const int xmin = (int)b.x, xmax = (int)b.y, ymin = (int)b.z, ymax = (int)b.w;
    if (xmin > xmax || ymin > ymax) continue;
    const int wd = xmax - xmin + 1, n = wd * (ymax - ymin + 1);
    const float inv = 1.0f / (float)wd;
    const uint32_t base = offsets[6 * (size_t)o + s];
    for (int i = lane; i < n; i += 64) {
      const int yy = (int)(((float)i + 0.5f) * inv), xx = i - yy * wd;
      const float w = W[(size_t)base + i];
      W2[(size_t)base + i] = w >= 0.0f ? w / divisor[(s * N + ymin + yy) * N + xmin + xx] : w;
    }
  }
}

// forward: out[o] = (sum tex[in] * f * area(in) / 4, sum of weights); backward: g_in[o] = sum g[other].rgb * f * area(o) / 4
// kNorm: the forward writes rgb / wsum to dst [.,3] and wsum to `wsum_out` (the division
// ops.py:458 does in torch); the backward then reads a 3-channel gradient that the caller has
// already divided by wsum.
// Sum over groups of kLanes consecutive lanes (8 = half a DPP row, 16 = a row, 64 = the wave); the total is
// valid in the last lane of every group.
template <int kLanes>
__device__ __forceinline__ float group_sum_last(float v) {
  v += pbr_dpp<0xb1>(v);
  v += pbr_dpp<0x4e>(v);
  if (kLanes == 8) return v + pbr_dpp<0x141>(v);  // row_half_mirror: the other quad of the 8-lane group
  v += pbr_dpp<0x124>(v);
  v += pbr_dpp<0x128>(v);
  if (kLanes == 64) {
    v += pbr_dpp<0x142, 0xa>(v);
    v += pbr_dpp<0x143, 0xc>(v);
  }
  return v;
}

// Applies the cached GGX weights: out[o] = sum_i W[o][i] * src[window_o[i]].  A texel's window is up to six
// face rectangles (`bounds`), and its weights are ONE contiguous run of the table (the rectangles in face
// order, `offsets[6 o]` onward), so the kernel treats the window as a flat candidate list:
//   * all six rectangles are loaded up front (six independent 16-byte loads: one memory round trip instead
//     of one per face) and every candidate index is mapped to its face with selects;
//   * the weight stream and the texel gathers of kU candidates per lane are issued together, the gathers
//     unconditionally (a rejected candidate, weight -1, still lies inside its face);
//   * kLanes = 8 / 16 pack eight / four texels into a wave for the levels whose windows hold ~10^2..10^3
//     candidates (the fixed per-texel work is shared by the wave), kLanes = 64 gives a whole wave to a texel.
// The pass streams the table once (0.74 GB per direction for the reference's 256..16 chain): it is HBM-bound.
template <bool kBackward, bool kNorm, int kLanes>
__device__ __forceinline__ void specular_apply_body(int block, int N, const float* __restrict__ src,
                                                    const float* __restrict__ bounds, const uint32_t* __restrict__ offsets,
                                                    const float* __restrict__ W, float* __restrict__ dst,
                                                    float* __restrict__ wsum_out) {
  constexpr int kPerWave = 64 / kLanes;
  constexpr int kU = 4;
  const int lane = threadIdx.x & 63;
  const int total = 6 * N * N;
  const int o_raw = (block * 4 + (threadIdx.x >> 6)) * kPerWave + lane / kLanes;
  const bool valid = o_raw < total;
  const int o = valid ? o_raw : total - 1;  // surplus groups stay in the wave for the DPP sums
  const int g = lane % kLanes;
  const int stride = (kBackward && !kNorm) ? 4 : 3;

  int wd[6], pre[7], base[6];
  float inv[6];
  const float4* b4 = reinterpret_cast<const float4*>(bounds + 24 * (size_t)o);
  const uint32_t first = offsets[6 * (size_t)o];
  pre[0] = 0;
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const float4 b = b4[s];
    const int xmin = (int)b.x, xmax = (int)b.y, ymin = (int)b.z, ymax = (int)b.w;
    const bool empty = xmin > xmax || ymin > ymax;
    wd[s] = empty ? 1 : xmax - xmin + 1;
    pre[s + 1] = pre[s] + (empty ? 0 : wd[s] * (ymax - ymin + 1));
    base[s] = empty ? 0 : (s * N + ymin) * N + xmin;
    // row = floor((loc + 0.5) / wd): the half-texel slack dwarfs the 1-ulp error of the hardware reciprocal
    inv[s] = __builtin_amdgcn_rcpf((float)wd[s]);
  }
  const int n = valid ? pre[6] : 0;
  const float* wrow = W + first;

  float wsum = 0.0f, c0 = 0, c1 = 0, c2 = 0;
  // face of candidate i: the last face with pre[k] <= i (empty faces are overridden by the next one)
  int f_wd, f_base, f_pre, f_end;
  float f_inv;
  auto find_face = [&](int i) {
    f_wd = wd[0]; f_base = base[0]; f_pre = 0; f_end = pre[1]; f_inv = inv[0];
#pragma unroll
    for (int k = 1; k < 6; k++) {
      const bool ge = i >= pre[k];
      f_wd = ge ? wd[k] : f_wd; f_base = ge ? base[k] : f_base; f_pre = ge ? pre[k] : f_pre;
      f_end = ge ? pre[k + 1] : f_end; f_inv = ge ? inv[k] : f_inv;
    }
  };
  find_face(g);
  for (int i0 = g; i0 < n; i0 += kLanes * kU) {
    float w[kU];
    int idx[kU];
    // Lanes walk their candidates in increasing order, so the face changes at most five times per texel:
    // the select chain runs only in iterations where some lane of the wave crosses a face (or the list) end.
    const bool cross = i0 + kLanes * (kU - 1) >= f_end && f_end < n;  // running past the END OF THE LIST is handled below
    if (__any(cross)) {
#pragma unroll
      for (int u = 0; u < kU; u++) {
        const int i = i0 + kLanes * u;
        const bool in = i < n;
        const int ic = in ? i : 0;
        w[u] = in ? wrow[i] : -1.0f;
        find_face(ic);
        const int loc = ic - f_pre;
        const int yy = (int)(((float)loc + 0.5f) * f_inv), xx = loc - yy * f_wd;
        idx[u] = f_base + yy * N + xx;
      }
      find_face(min(i0 + kLanes * kU, n - 1));
    } else {
#pragma unroll
      for (int u = 0; u < kU; u++) {
        const int i = i0 + kLanes * u;
        const bool in = i < n;
        w[u] = in ? wrow[i] : -1.0f;
        const int loc = in ? i - f_pre : 0;
        const int yy = (int)(((float)loc + 0.5f) * f_inv), xx = loc - yy * f_wd;
        idx[u] = f_base + yy * N + xx;
      }
    }
    float t0[kU], t1[kU], t2[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const float* t = src + (size_t)stride * idx[u];
      t0[u] = t[0]; t1[u] = t[1]; t2[u] = t[2];
    }
#pragma unroll
    for (int u = 0; u < kU; u++) {
      if (w[u] >= 0.0f) {  // -1 marks candidates outside the cone
        c0 += t0[u] * w[u]; c1 += t1[u] * w[u]; c2 += t2[u] * w[u];
        wsum += w[u];
      }
    }
  }
  c0 = group_sum_last<kLanes>(c0); c1 = group_sum_last<kLanes>(c1); c2 = group_sum_last<kLanes>(c2);
  if (!kBackward) wsum = group_sum_last<kLanes>(wsum);
  if (g == kLanes - 1 && valid) {
    if (kBackward) {
      float* q = dst + 3 * (size_t)o;
      q[0] = c0; q[1] = c1; q[2] = c2;
    } else if (kNorm) {
      float* q = dst + 3 * (size_t)o;
      q[0] = c0 / wsum; q[1] = c1 / wsum; q[2] = c2 / wsum;
      wsum_out[o] = wsum;
    } else {
      float* q = dst + 4 * (size_t)o;
      q[0] = c0; q[1] = c1; q[2] = c2; q[3] = wsum;
    }
  }
}

template <bool kBackward, bool kNorm, int kLanes>
__global__ void __launch_bounds__(256)
specular_apply_kernel(int N, const float* __restrict__ src, const float* __restrict__ bounds,
                      const uint32_t* __restrict__ offsets, const float* __restrict__ W,
                      float* __restrict__ dst, float* __restrict__ wsum_out) {
  specular_apply_body<kBackward, kNorm, kLanes>(blockIdx.x, N, src, bounds, offsets, W, dst, wsum_out);
}

// All levels of the chain in ONE launch (gigs_specular_cubemap_multi_*): the levels are independent, so their
// workgroups share one grid -- the largest level first -- instead of five dependent launches (five kernel nodes with
// 10-17 us of dispatch latency between them on the light's side stream, each with its own under-filled tail).
struct SpecLevel {
  int N, lanes, block_begin;
  const float* src;
  const float* bounds;
  const uint32_t* offsets;
  const float* W;
  float* dst;
  float* wsum_out;
};
struct SpecLevels { int n; SpecLevel lv[8]; };

template <bool kBackward>
__global__ void __launch_bounds__(256) specular_apply_multi_kernel(SpecLevels L) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < 8; i++)
    if (i < L.n && (int)blockIdx.x >= L.lv[i].block_begin) k = i;
  const SpecLevel& v = L.lv[k];
  const int block = (int)blockIdx.x - v.block_begin;
  if (v.lanes == 8) specular_apply_body<kBackward, true, 8>(block, v.N, v.src, v.bounds, v.offsets, v.W, v.dst, v.wsum_out);
  else if (v.lanes == 16) specular_apply_body<kBackward, true, 16>(block, v.N, v.src, v.bounds, v.offsets, v.W, v.dst, v.wsum_out);
  else specular_apply_body<kBackward, true, 64>(block, v.N, v.src, v.bounds, v.offsets, v.W, v.dst, v.wsum_out);
}

// tuning knobs (gigs_options.spec_max8 / spec_max16): largest mean window served by 8- and by 16-lane groups
static int spec_lanes_for(const Options& o, int avg_window) {
  return (avg_window > 0 && avg_window <= o.spec_max8) ? 8 : (avg_window > 0 && avg_window <= o.spec_max16) ? 16 : 64;
}

template <bool kBackward, bool kNorm>
static void launch_specular_apply(const Options& o, int res, int avg_window, const float* src, const float* bounds,
                                  const uint32_t* offsets, const float* W, float* dst, float* wsum_out, hipStream_t s) {
  const int total = 6 * res * res;
  const int max8 = o.spec_max8, max16 = o.spec_max16;
  if (avg_window > 0 && avg_window <= max8) {
    const int waves = (total + 7) / 8;
    hipLaunchKernelGGL((specular_apply_kernel<kBackward, kNorm, 8>), dim3((waves + 3) / 4), dim3(256), 0, s, res, src,
                       bounds, offsets, W, dst, wsum_out);
  } else if (avg_window > 0 && avg_window <= max16) {
    const int waves = (total + 3) / 4;
    hipLaunchKernelGGL((specular_apply_kernel<kBackward, kNorm, 16>), dim3((waves + 3) / 4), dim3(256), 0, s, res, src,
                       bounds, offsets, W, dst, wsum_out);
  } else {
    hipLaunchKernelGGL((specular_apply_kernel<kBackward, kNorm, 64>), dim3((total + 3) / 4), dim3(256), 0, s, res, src,
                       bounds, offsets, W, dst, wsum_out);
  }
}

// ------------------------------------------------------------------------------------------
// cube / 2-D texture sampling
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int cube_face_uv(float x, float y, float z, float& u, float& v) {
  const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
  int idx;
  float c;
  if (az > fmaxf(ax, ay)) { idx = 4; c = z; }
  else if (ay > ax) { idx = 2; c = y; y = z; }
  else { idx = 0; c = x; x = z; }
  if (c < 0.f) idx += 1;
  const float m = (1.0f / fabsf(c)) * 0.5f;
  const float m0 = (idx == 0 || idx == 5) ? -m : m;
  const float m1 = (idx != 2) ? -m : m;
  u = x * m0 + 0.5f;
  v = y * m1 + 0.5f;
  if (!isfinite(u) || !isfinite(v)) return -1;
  u = fminf(fmaxf(u, 0.f), 1.f);
  v = fminf(fmaxf(v, 0.f), 1.f);
  return idx;
}

struct Taps { int idx[4]; float w[4]; };

__device__ __forceinline__ bool cube_taps(int res, float dx, float dy, float dz, Taps& t) {
  float u, v;
  const int face = cube_face_uv(dx, dy, dz, u, v);
  if (face < 0) return false;
  const float fu = u * (float)res - 0.5f, fv = v * (float)res - 0.5f;
  const float flu = floorf(fu), flv = floorf(fv);
  const int iu0 = (int)flu, iv0 = (int)flv;
  const float tu = fu - flu, tv = fv - flv;
  float wsum = 0.0f;
  bool dropped = false;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int ox = k & 1, oy = k >> 1;
    const int ix = iu0 + ox, iy = iv0 + oy;
    const float w = (ox ? tu : 1.0f - tu) * (oy ? tv : 1.0f - tv);
    const bool out_x = ix < 0 || ix >= res, out_y = iy < 0 || iy >= res;
    int idx;
    if (!out_x && !out_y) {
      idx = (face * res + iy) * res + ix;
    } else if (out_x && out_y) {
      idx = -1;
      dropped = true;
    } else {
      const float a = 2.0f * (((float)ix + 0.5f) / (float)res) - 1.0f;
      const float b = 2.0f * (((float)iy + 0.5f) / (float)res) - 1.0f;
      const v3 d = cube_dir_raw(a, b, face);
      float u2, v2;
      const int f2 = cube_face_uv(d.x, d.y, d.z, u2, v2);
      const int x2 = min(res - 1, max(0, (int)floorf(u2 * (float)res)));
      const int y2 = min(res - 1, max(0, (int)floorf(v2 * (float)res)));
      idx = (f2 * res + y2) * res + x2;
    }
    t.idx[k] = idx;
    t.w[k] = w;
    if (idx >= 0) wsum += w;
  }
  if (dropped) {
#pragma unroll
    for (int k = 0; k < 4; k++) t.w[k] = t.idx[k] >= 0 ? t.w[k] / wsum : 0.0f;
  }
  return true;
}

__device__ __forceinline__ v3 cube_sample(const float* __restrict__ tex, const Taps& t) {
  v3 r = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (t.idx[k] >= 0) {
      const float* p = tex + 3 * (size_t)t.idx[k];
      r.x += p[0] * t.w[k];
      r.y += p[1] * t.w[k];
      r.z += p[2] * t.w[k];
    }
  return r;
}

// cubemap_mip: forward 2x2 average, backward = bilinear cube lookup of 0.25 * dout
__global__ void __launch_bounds__(256)
cubemap_mip_fwd_kernel(int r, int C, const float* __restrict__ in, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 6 * r * r * C) return;
  const int c = i % C, x = (i / C) % r, y = (i / (C * r)) % r, f = i / (C * r * r);
  const int R2 = 2 * r;
  const float* p = in + ((size_t)(f * R2 + 2 * y) * R2 + 2 * x) * C + c;
  out[i] = (p[0] + p[C] + p[(size_t)R2 * C] + p[(size_t)R2 * C + C]) * 0.25f;
}

__global__ void __launch_bounds__(256)
cubemap_mip_bwd_kernel(int r, const float* __restrict__ dout, float* __restrict__ din, const float* __restrict__ add,
                       const float* __restrict__ dout2) {
  const int res = 2 * r;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 6 * res * res) return;
  const int x = i % res, y = (i / res) % res, s = i / (res * res);
  const float start = -1.0f + 1.0f / (float)res, end = 1.0f - 1.0f / (float)res;
  const float stepv = (end - start) / (float)(res - 1);
  const float gx = x < res / 2 ? start + stepv * (float)x : end - stepv * (float)(res - 1 - x);
  const float gy = y < res / 2 ? start + stepv * (float)y : end - stepv * (float)(res - 1 - y);
  v3 d = cube_dir_raw(gx, gy, s);
  const float n = fmaxf(sqrtf(d.x * d.x + d.y * d.y + d.z * d.z), 1e-12f);
  d = {d.x / n, d.y / n, d.z / n};
  Taps t;
  v3 v = {0, 0, 0};
  if (cube_taps(r, d.x, d.y, d.z, t)) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (t.idx[k] >= 0) {
        const float* p = dout + 3 * (size_t)t.idx[k];
        float p0 = p[0], p1 = p[1], p2 = p[2];
        if (dout2) {  // a second gradient of the coarse level (it fed two filters): the sum autograd would have formed first
          const float* q = dout2 + 3 * (size_t)t.idx[k];
          p0 += q[0]; p1 += q[1]; p2 += q[2];
        }
        v.x += (p0 * 0.25f) * t.w[k];
        v.y += (p1 * 0.25f) * t.w[k];
        v.z += (p2 * 0.25f) * t.w[k];
      }
  }
  if (add) {  // the level's own gradient (it also feeds a filter): summed here instead of by a separate pass
    v.x += add[3 * (size_t)i]; v.y += add[3 * (size_t)i + 1]; v.z += add[3 * (size_t)i + 2];
  }
  din[3 * (size_t)i] = v.x; din[3 * (size_t)i + 1] = v.y; din[3 * (size_t)i + 2] = v.z;
}

// dr.texture(cubemap[None], dirs[None], filter_mode="linear", boundary_mode="cube") for a list of directions
// (train.py:409-417 envmap TV, render.py:80 / relight.py:108 envmap export): out is [n,3], or [3,n] planes if planar.
__global__ void __launch_bounds__(256)
cube_texture_fwd_kernel(int res, const float* __restrict__ tex, int n, const float* __restrict__ dirs,
                        float* __restrict__ out, int planar) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Taps t;
  v3 v = {0, 0, 0};
  if (cube_taps(res, dirs[3 * (size_t)i], dirs[3 * (size_t)i + 1], dirs[3 * (size_t)i + 2], t)) v = cube_sample(tex, t);
  if (planar) {
    out[i] = v.x; out[(size_t)n + i] = v.y; out[2 * (size_t)n + i] = v.z;
  } else {
    out[3 * (size_t)i] = v.x; out[3 * (size_t)i + 1] = v.y; out[3 * (size_t)i + 2] = v.z;
  }
}

// latlong_to_cubemap (relight.py:92-111): every cube texel looks its direction up in an equirectangular map.
//   gy, gx = linspace(-1 + 1/res, 1 - 1/res, res)   (torch: start + i*step below the middle, end - (n-1-i)*step above)
//   v = normalize(cube_to_dir(face, gx, gy));  tu = atan2(v.x, -v.z) / (2 pi) + 0.5;  tv = acos(clamp(v.y, -1, 1)) / pi
//   out = dr.texture(latlong[None], (tu, tv), filter_mode="linear")   -- nvdiffrast, third party and absent: restated
//   from its documented behaviour (bilinear, texel centres at (i + 0.5)/size, boundary_mode "wrap"): PARITY UNPINNED.
__device__ __forceinline__ float torch_linspace(float start, float end, int n, int i) {
  if (n <= 1) return start;
  const float step = (end - start) / (float)(n - 1);
  return (i < n / 2) ? start + step * (float)i : end - step * (float)(n - 1 - i);
}
__global__ void __launch_bounds__(256)
latlong_to_cubemap_kernel(int res_y, int res_x, int Hl, int Wl, int C, const float* __restrict__ latlong,
                          float* __restrict__ cube) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = 6 * res_y * res_x;
  if (i >= n) return;
  const int face = i / (res_y * res_x), rem = i - face * res_y * res_x;
  const int ty = rem / res_x, tx = rem - ty * res_x;
  const float gy = torch_linspace(-1.0f + 1.0f / (float)res_y, 1.0f - 1.0f / (float)res_y, res_y, ty);
  const float gx = torch_linspace(-1.0f + 1.0f / (float)res_x, 1.0f - 1.0f / (float)res_x, res_x, tx);
  float rx, ry, rz;  // cube_to_dir (relight.py:75-89)
  switch (face) {
    case 0: rx = 1.0f; ry = -gy; rz = -gx; break;
    case 1: rx = -1.0f; ry = -gy; rz = gx; break;
    case 2: rx = gx; ry = 1.0f; rz = gy; break;
    case 3: rx = gx; ry = -1.0f; rz = -gy; break;
    case 4: rx = gx; ry = -gy; rz = 1.0f; break;
    default: rx = -gx; ry = -gy; rz = -1.0f; break;
  }
  const float inv = 1.0f / fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), 1e-12f);  // F.normalize
  rx *= inv; ry *= inv; rz *= inv;
  const float kPi = 3.14159265358979323846f;
  const float tu = atan2f(rx, -rz) / (2.0f * kPi) + 0.5f;
  const float tv = acosf(fminf(fmaxf(ry, -1.0f), 1.0f)) / kPi;
  const float u = tu * (float)Wl - 0.5f, v = tv * (float)Hl - 0.5f;
  const float fu0 = floorf(u), fv0 = floorf(v);
  const float fu = u - fu0, fv = v - fv0;
  auto wrap = [](int a, int m) { a %= m; return a < 0 ? a + m : a; };
  const int iu0 = wrap((int)fu0, Wl), iu1 = wrap((int)fu0 + 1, Wl);
  const int iv0 = wrap((int)fv0, Hl), iv1 = wrap((int)fv0 + 1, Hl);
  for (int c = 0; c < C; c++) {
    const float a = latlong[((size_t)iv0 * Wl + iu0) * C + c] * (1.0f - fu) + latlong[((size_t)iv0 * Wl + iu1) * C + c] * fu;
    const float b = latlong[((size_t)iv1 * Wl + iu0) * C + c] * (1.0f - fu) + latlong[((size_t)iv1 * Wl + iu1) * C + c] * fu;
    cube[(size_t)i * C + c] = a * (1.0f - fv) + b * fv;
  }
}

// ------------------------------------------------------------------------------------------
// shade
// ------------------------------------------------------------------------------------------
struct ShadeArgs {
  int H, W;
  const float *normals, *view_dirs, *albedo, *roughness;
  const uint8_t* mask;
  const float *occlusion, *metallic, *background;
  const float* diffuse; int diffuse_res;
  int L; const float* spec[8]; int spec_res[8];
  const float* lut; int lut_w, lut_h;
  int tone, gamma;
#ifdef GIGS_DIAG
  int ablate;  // -DGIGS_DIAG builds only (tools/gpu_ablate_shade.sh, env GIGS_ABLATE): bit 0 skips the diffuse-map atomics, bit 1 the specular ones
#endif
  int part;    // backward: 0 = everything; 1 = the material gradients only; 2 = the light-texture gradients only (gigs_shade_ext)
  // backward: gradient textures small enough to be accumulated per workgroup in LDS
  int lds_total;        // floats of dynamic LDS
  int lds_diffuse_off;  // offset (floats) of the diffuse-map accumulator, or -1
  int lds_spec_off[8];  // per specular level, or -1
  // forward outputs
  float *render_rgb, *diffuse_rgb, *specular_rgb, *diffuse_light;
  // backward inputs (may be null) and outputs
  const float *g_render, *g_diffuse_rgb, *g_specular_rgb, *g_diffuse_light;
  float *d_albedo, *d_roughness, *d_metallic, *d_diffuse;
  float* d_spec[8];
  // layout of normals / albedo and of every [H,W,3] output and gradient: element (p, c) at p * ps + c * cs
  // ([H,W,3]: ps = 3, cs = 1;  [3,H,W] planes: ps = 1, cs = H*W).  view_dirs is always [H,W,3].
  int ps, cs;
  // gigs_shade_ext (stage-2 fusion): roughness = raw * rough_scale + rough_bias, extra outputs / gradients
  float rough_scale, rough_bias;
  float *out_F0, *out_linear, *out_roughness;
  const float *g_albedo_mul_a, *g_albedo_mul_b, *g_roughness_add, *g_metallic_add;
  const float *g_scale, *lamb_mask, *lamb_acc4;
};

__device__ __forceinline__ float get_mip(float r, int L, float& dmip_dr) {  // pbr/light.py:142-152
  const float MINR = 0.08f, MAXR = 0.5f;
  if (r < MAXR) {
    const float c = fminf(fmaxf(r, MINR), MAXR);
    dmip_dr = (r >= MINR && r <= MAXR) ? (1.0f / (MAXR - MINR)) * (float)(L - 2) : 0.0f;
    return (c - MINR) / (MAXR - MINR) * (float)(L - 2);
  }
  const float c = fminf(fmaxf(r, MAXR), 1.0f);
  dmip_dr = (r >= MAXR && r <= 1.0f) ? 1.0f / (1.0f - MAXR) : 0.0f;
  return (c - MAXR) / (1.0f - MAXR) + (float)L - 2.0f;
}
__device__ __forceinline__ float aces(float x, float& d) {  // pbr/shade.py:33-47
  const float a = 2.51f, b = 0.03f, c = 2.43f, dd = 0.59f, e = 0.14f;
  const float num = x * (a * x + b), den = x * (c * x + dd) + e;
  d = ((2 * a * x + b) * den - num * (2 * c * x + dd)) / (den * den);
  return num / den;
}

// Everything both passes need, computed identically in forward and backward.
struct ShadePix {
  v3 a, dl_raw, dl, drgb, sp, s0, s1, F0, refl, srgb;
  float r, occ, m, fgx, fgy, dfgx_dv, dfgy_dv, lf, dmdr;
  int l0, l1;
  bool lvl_inside;
  Taps td, t0, t1;
  bool has_d, has0, has1;
};

__device__ __forceinline__ void shade_pixel(const ShadeArgs& A, int p, ShadePix& q) {
  const size_t e = (size_t)p * A.ps, cs = A.cs;
  const v3 n = {A.normals[e], A.normals[e + cs], A.normals[e + 2 * cs]};
  const v3 v = {A.view_dirs[3 * p], A.view_dirs[3 * p + 1], A.view_dirs[3 * p + 2]};
  q.a = {A.albedo[e], A.albedo[e + cs], A.albedo[e + 2 * cs]};
  q.r = A.roughness[p] * A.rough_scale + A.rough_bias;  // scale 1, bias 0 (exact) unless the caller fuses the remap
  const float ndv = n.x * v.x + n.y * v.y + n.z * v.z;
  const float c2 = 2.0f * fmaxf(ndv, 0.0f);
  const v3 ref = {c2 * n.x - v.x, c2 * n.y - v.y, c2 * n.z - v.z};
  const v3 nt = {-n.y, n.z, -n.x}, vt = {-v.y, v.z, -v.x}, rt = {-ref.y, ref.z, -ref.x};
  q.has_d = cube_taps(A.diffuse_res, nt.x, nt.y, nt.z, q.td);
  q.dl_raw = q.has_d ? cube_sample(A.diffuse, q.td) : v3{0, 0, 0};
  q.occ = A.occlusion ? A.occlusion[p] : 1.0f;
  q.dl = A.occlusion ? q.dl_raw * q.occ : q.dl_raw;
  q.drgb = {q.dl.x * q.a.x, q.dl.y * q.a.y, q.dl.z * q.a.z};
  const float nov = fminf(fmaxf(nt.x * vt.x + nt.y * vt.y + nt.z * vt.z, 1e-4f), 1.0f);
  {
    const float fu = nov * (float)A.lut_w - 0.5f, fv = q.r * (float)A.lut_h - 0.5f;
    const float flu = floorf(fu), flv = floorf(fv);
    const float tu = fu - flu, tv = fv - flv;
    const int x0 = min(A.lut_w - 1, max(0, (int)flu)), x1 = min(A.lut_w - 1, max(0, (int)flu + 1));
    const int y0 = min(A.lut_h - 1, max(0, (int)flv)), y1 = min(A.lut_h - 1, max(0, (int)flv + 1));
    const float2 t00 = reinterpret_cast<const float2*>(A.lut)[(size_t)y0 * A.lut_w + x0];
    const float2 t10 = reinterpret_cast<const float2*>(A.lut)[(size_t)y0 * A.lut_w + x1];
    const float2 t01 = reinterpret_cast<const float2*>(A.lut)[(size_t)y1 * A.lut_w + x0];
    const float2 t11 = reinterpret_cast<const float2*>(A.lut)[(size_t)y1 * A.lut_w + x1];
    const float w00 = (1 - tu) * (1 - tv), w10 = tu * (1 - tv), w01 = (1 - tu) * tv, w11 = tu * tv;
    q.fgx = t00.x * w00 + t10.x * w10 + t01.x * w01 + t11.x * w11;
    q.fgy = t00.y * w00 + t10.y * w10 + t01.y * w01 + t11.y * w11;
    // d/dv of the bilinear blend (taps fixed), times dv/d(roughness) = lut_h
    q.dfgx_dv = ((t01.x - t00.x) * (1 - tu) + (t11.x - t10.x) * tu) * (float)A.lut_h;
    q.dfgy_dv = ((t01.y - t00.y) * (1 - tu) + (t11.y - t10.y) * tu) * (float)A.lut_h;
  }
  const float lvl_raw = get_mip(q.r, A.L, q.dmdr);
  const float lvl = fminf(fmaxf(lvl_raw, 0.0f), (float)(A.L - 1));
  q.lvl_inside = lvl_raw >= 0.0f && lvl_raw <= (float)(A.L - 1);
  q.l0 = min((int)floorf(lvl), A.L - 1);
  q.l1 = min(q.l0 + 1, A.L - 1);
  q.lf = lvl - (float)q.l0;
  q.has0 = cube_taps(A.spec_res[q.l0], rt.x, rt.y, rt.z, q.t0);
  q.s0 = q.has0 ? cube_sample(A.spec[q.l0], q.t0) : v3{0, 0, 0};
  q.has1 = false;
  q.s1 = {0, 0, 0};
  if (q.l1 != q.l0) {
    q.has1 = cube_taps(A.spec_res[q.l1], rt.x, rt.y, rt.z, q.t1);
    if (q.has1) q.s1 = cube_sample(A.spec[q.l1], q.t1);
    q.sp = {q.s0.x * (1 - q.lf) + q.s1.x * q.lf, q.s0.y * (1 - q.lf) + q.s1.y * q.lf, q.s0.z * (1 - q.lf) + q.s1.z * q.lf};
  } else {
    q.sp = q.s0;
  }
  if (A.metallic) {
    q.m = A.metallic[p];
    q.F0 = {(1.0f - q.m) * 0.04f + q.a.x * q.m, (1.0f - q.m) * 0.04f + q.a.y * q.m, (1.0f - q.m) * 0.04f + q.a.z * q.m};
  } else {
    q.m = 0.0f;
    q.F0 = {0.04f, 0.04f, 0.04f};
  }
  q.refl = {q.F0.x * q.fgx + q.fgy, q.F0.y * q.fgx + q.fgy, q.F0.z * q.fgx + q.fgy};
  q.srgb = {q.sp.x * q.refl.x, q.sp.y * q.refl.y, q.sp.z * q.refl.z};
}

__global__ void __launch_bounds__(256)
shade_fwd_kernel(ShadeArgs A) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= A.H * A.W) return;
  ShadePix q;
  shade_pixel(A, p, q);
  float rr[3] = {q.drgb.x + q.srgb.x, q.drgb.y + q.srgb.y, q.drgb.z + q.srgb.z};
  float dd;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float x = rr[c];
    if (A.tone) x = aces(x, dd);
    x = fminf(fmaxf(x, 0.0f), 1.0f);
    if (A.gamma) x = lin2srgb(x, dd);
    rr[c] = x;
  }
  v3 drgb = q.drgb, srgb = q.srgb;
  if (A.gamma) {
    drgb = {lin2srgb(drgb.x, dd), lin2srgb(drgb.y, dd), lin2srgb(drgb.z, dd)};
    srgb = {lin2srgb(srgb.x, dd), lin2srgb(srgb.y, dd), lin2srgb(srgb.z, dd)};
  }
  const bool mk = A.mask[p] != 0;
  const size_t e = (size_t)p * A.ps, cs = A.cs;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    rr[c] = mk ? rr[c] : (A.background ? A.background[e + c * cs] : 0.0f);
    A.render_rgb[e + c * cs] = rr[c];
  }
  if (A.diffuse_rgb) { A.diffuse_rgb[e] = drgb.x; A.diffuse_rgb[e + cs] = drgb.y; A.diffuse_rgb[e + 2 * cs] = drgb.z; }
  if (A.specular_rgb) { A.specular_rgb[e] = srgb.x; A.specular_rgb[e + cs] = srgb.y; A.specular_rgb[e + 2 * cs] = srgb.z; }
  if (A.diffuse_light) { A.diffuse_light[e] = q.dl.x; A.diffuse_light[e + cs] = q.dl.y; A.diffuse_light[e + 2 * cs] = q.dl.z; }
  if (A.out_F0) { A.out_F0[e] = q.F0.x; A.out_F0[e + cs] = q.F0.y; A.out_F0[e + 2 * cs] = q.F0.z; }
  if (A.out_linear) {
#pragma unroll
    for (int c = 0; c < 3; c++) A.out_linear[e + c * cs] = srgb2lin(rr[c]);
  }
  if (A.out_roughness) A.out_roughness[p] = q.r;
}

constexpr int kShadeBwdBlock = 1024;     // 16 waves share one set of LDS accumulators
constexpr int kShadeLdsBudget = 30 * 1024;  // floats (120 KB of the CU's 160 KB)

// Scatter-add of one RGB triple per lane into a texture gradient, with neighbouring lanes that
// hit the SAME texel summed first: neighbouring pixels usually share bilinear taps, and 64
// same-address float atomics serialise.  Runs of equal keys inside each 16-lane DPP row are
// reduced with a segmented scan (row_shr 1/2/4/8, in registers); only the last lane of a run
// issues the three atomics.  Must be called by all 64 lanes (key < 0 = nothing to add).
template <int kCtrl>
__device__ __forceinline__ float row_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, 0xf, 0xf, false));
}
template <int kCtrl>
__device__ __forceinline__ int row_i(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, kCtrl, 0xf, 0xf, false);
}
template <bool kLds>
__device__ __forceinline__ void run_add3(float* target, int key, float v0, float v1, float v2) {
  const int head = (row_i<0x111>(~key, key) != key) ? 1 : 0;  // row_shr:1; lane 0 of a row is a head
  int f = head;
#define GIGS_SEG_STEP(CTRL)                                            \
  {                                                                    \
    const float u0 = row_f<CTRL>(v0), u1 = row_f<CTRL>(v1), u2 = row_f<CTRL>(v2); \
    const int fu = row_i<CTRL>(1, f);                                  \
    if (!f) { v0 += u0; v1 += u1; v2 += u2; }                          \
    f |= fu;                                                           \
  }
  GIGS_SEG_STEP(0x111) GIGS_SEG_STEP(0x112) GIGS_SEG_STEP(0x114) GIGS_SEG_STEP(0x118)
#undef GIGS_SEG_STEP
  const int tail = row_i<0x101>(1, head);  // row_shl:1 -> head flag of the next lane; 1 at the row end
  if (key >= 0 && tail && target) {
    if (v0 != 0.0f) atomicAdd(target, v0);
    if (v1 != 0.0f) atomicAdd(target + 1, v1);
    if (v2 != 0.0f) atomicAdd(target + 2, v2);
  }
}

// scatters g_out through the same taps; accumulates into d_tex (caller zeroes).  Neighbouring samples that hit the same
// texel are pre-summed per 16-lane row (run_add3): the envmap TV's latlong grid (losses.get_envmap_dirs) sends whole rows
// of 512 samples to the four texels around a pole, which serialised 0.32 ms of memory-side atomics.
__global__ void __launch_bounds__(256)
cube_texture_bwd_kernel(int res, int n, const float* __restrict__ dirs, const float* __restrict__ g_out,
                        float* __restrict__ d_tex, int planar) {
  const int gi = blockIdx.x * 256 + threadIdx.x;
  const bool live = gi < n;
  const int i = live ? gi : 0;  // dead lanes add nothing: the DPP scans need every lane
  Taps t;
  const bool ok = cube_taps(res, dirs[3 * (size_t)i], dirs[3 * (size_t)i + 1], dirs[3 * (size_t)i + 2], t) && live;
  const float g0 = planar ? g_out[i] : g_out[3 * (size_t)i];
  const float g1 = planar ? g_out[(size_t)n + i] : g_out[3 * (size_t)i + 1];
  const float g2 = planar ? g_out[2 * (size_t)n + i] : g_out[3 * (size_t)i + 2];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int idx = (ok && t.idx[k] >= 0 && t.w[k] != 0.0f) ? t.idx[k] : -1;
    const float w = idx >= 0 ? t.w[k] : 0.0f;
    run_add3<false>(d_tex + 3 * (size_t)max(idx, 0), idx, g0 * w, g1 * w, g2 * w);
  }
}

// The same backward as a gather, for direction sets that do not change (the envmap TV's latlong grid): the taps of every
// sample are exported once (cube_taps_export_kernel), the host sorts them by texel into a CSR list (pbr/texture.py), and
// each texel then sums its own entries in sample order -- no atomics (6.3 M of them bound the scatter at 0.11 ms for the
// 512 x 1024 grid), no zero-fill, reproducible.  Texels with more than `heavy` entries (the poles) get a wave each.
__global__ void __launch_bounds__(256)
cube_taps_export_kernel(int res, int n, const float* __restrict__ dirs, int* __restrict__ idx, float* __restrict__ w) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Taps t;
  const bool ok = cube_taps(res, dirs[3 * (size_t)i], dirs[3 * (size_t)i + 1], dirs[3 * (size_t)i + 2], t);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const bool use = ok && t.idx[k] >= 0 && t.w[k] != 0.0f;
    idx[4 * (size_t)i + k] = use ? t.idx[k] : -1;
    w[4 * (size_t)i + k] = use ? t.w[k] : 0.0f;
  }
}

__global__ void __launch_bounds__(256)
cube_texture_bwd_gather_kernel(int n_tex, const int* __restrict__ offsets, const int* __restrict__ ent_sample,
                               const float* __restrict__ ent_w, const float* __restrict__ g_out, int n, int planar,
                               float* __restrict__ d_tex, int heavy, int light_blocks, const int* __restrict__ heavy_ids,
                               int n_heavy) {
  auto grad = [&](int s, float& a, float& b, float& c) {
    if (planar) { a = g_out[s]; b = g_out[(size_t)n + s]; c = g_out[2 * (size_t)n + s]; }
    else { a = g_out[3 * (size_t)s]; b = g_out[3 * (size_t)s + 1]; c = g_out[3 * (size_t)s + 2]; }
  };
  if ((int)blockIdx.x < light_blocks) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tex) return;
    const int b = offsets[t], e = offsets[t + 1];
    if (e - b > heavy) return;  // summed by a wave below
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    for (int k = b; k < e; k++) {
      float g0, g1, g2;
      grad(ent_sample[k], g0, g1, g2);
      const float w = ent_w[k];
      c0 += g0 * w; c1 += g1 * w; c2 += g2 * w;
    }
    d_tex[3 * (size_t)t] = c0; d_tex[3 * (size_t)t + 1] = c1; d_tex[3 * (size_t)t + 2] = c2;
    return;
  }
  const int h = ((int)blockIdx.x - light_blocks) * 4 + (threadIdx.x >> 6);
  if (h >= n_heavy) return;  // wave-uniform
  const int lane = threadIdx.x & 63;
  const int t = heavy_ids[h];
  const int b = offsets[t], e = offsets[t + 1];
  float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
  for (int k = b + lane; k < e; k += 64) {
    float g0, g1, g2;
    grad(ent_sample[k], g0, g1, g2);
    const float w = ent_w[k];
    c0 += g0 * w; c1 += g1 * w; c2 += g2 * w;
  }
  c0 = wave_sum63(c0); c1 = wave_sum63(c1); c2 = wave_sum63(c2);
  if (lane == 63) { d_tex[3 * (size_t)t] = c0; d_tex[3 * (size_t)t + 1] = c1; d_tex[3 * (size_t)t + 2] = c2; }
}

// Gradients of the light textures: hundreds of thousands of pixels add into a few thousand texels of
// the coarse levels (16^2 diffuse, 16^2 / 32^2 specular), which serialises memory-side float atomics on
// the same addresses (measured: 0.72 of 0.77 ms).  Every level that fits is therefore accumulated per
// 1024-lane workgroup in LDS (up to 120 KB) and flushed once, non-zero entries only; larger levels take
// global atomics.  In both cases neighbouring lanes hitting the same texel are pre-summed (run_add3).
__global__ void __launch_bounds__(kShadeBwdBlock)
shade_bwd_kernel(ShadeArgs A) {
  extern __shared__ __align__(16) float s_lds[];
  for (int i = threadIdx.x; i < A.lds_total; i += kShadeBwdBlock) s_lds[i] = 0.0f;
  __syncthreads();
  // persistent workgroups: each keeps its LDS accumulators across several 1024-pixel chunks, so the zero-fill
  // and the flush (and the flush's global atomics) are paid once per workgroup, not once per chunk
  const int n_chunks = (A.H * A.W + kShadeBwdBlock - 1) / kShadeBwdBlock;
  for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
  const int pg = chunk * kShadeBwdBlock + threadIdx.x;
  const bool live = pg < A.H * A.W;
  const int p = live ? pg : 0;  // dead lanes shade pixel 0 and add nothing: the DPP scans need every lane
  ShadePix q;
  shade_pixel(A, p, q);
  float g_dl[3] = {0, 0, 0}, g_sp[3] = {0, 0, 0};
  if (live) {
    const bool mk = A.mask[p] != 0;
    const size_t e = (size_t)p * A.ps, cs = A.cs;
    const float gscale = A.g_scale ? A.g_scale[0] : 1.0f;  // x * 1.0f is exact: the plain operator is unchanged
    float g_d[3] = {0, 0, 0}, g_s[3] = {0, 0, 0};  // grads w.r.t. linear diffuse_rgb / specular_rgb
    const float pre_d[3] = {q.drgb.x, q.drgb.y, q.drgb.z}, pre_s[3] = {q.srgb.x, q.srgb.y, q.srgb.z};
#pragma unroll
    for (int c = 0; c < 3; c++) {
      // render = where(mask, gamma(clamp(tone(d + s))), bg)
      float g = (A.g_render && mk) ? A.g_render[e + c * cs] * gscale : 0.0f;
      if (g != 0.0f) {
        float x = pre_d[c] + pre_s[c], d_tone = 1.0f, d_gam = 1.0f;
        if (A.tone) x = aces(x, d_tone);
        const float xc = fminf(fmaxf(x, 0.0f), 1.0f);
        const float d_clamp = (x >= 0.0f && x <= 1.0f) ? 1.0f : 0.0f;
        if (A.gamma) lin2srgb(xc, d_gam);
        g = g * d_gam * d_clamp * d_tone;
      }
      float gd = A.g_diffuse_rgb ? A.g_diffuse_rgb[e + c * cs] : 0.0f;
      float gs = A.g_specular_rgb ? A.g_specular_rgb[e + c * cs] : 0.0f;
      if (A.gamma) {
        float d1, d2;
        lin2srgb(pre_d[c], d1);
        lin2srgb(pre_s[c], d2);
        gd *= d1;
        gs *= d2;
      }
      g_d[c] = g + gd;
      g_s[c] = g + gs;
    }
    const float av[3] = {q.a.x, q.a.y, q.a.z}, dlv[3] = {q.dl.x, q.dl.y, q.dl.z};
    const float spv[3] = {q.sp.x, q.sp.y, q.sp.z}, F0v[3] = {q.F0.x, q.F0.y, q.F0.z};
    const float reflv[3] = {q.refl.x, q.refl.y, q.refl.z};
    const float s0v[3] = {q.s0.x, q.s0.y, q.s0.z}, s1v[3] = {q.s1.x, q.s1.y, q.s1.z};
    float d_alb[3], d_m = 0.0f, d_fgx = 0.0f, d_fgy = 0.0f, d_lvl = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      d_alb[c] = g_d[c] * dlv[c];
      g_dl[c] = g_d[c] * av[c] + (A.g_diffuse_light ? A.g_diffuse_light[e + c * cs] : 0.0f);
      g_sp[c] = g_s[c] * reflv[c];
      const float g_refl = g_s[c] * spv[c];
      const float dF0 = g_refl * q.fgx;
      d_fgx += g_refl * F0v[c];
      d_fgy += g_refl;
      if (A.metallic) {
        d_m += dF0 * (av[c] - 0.04f);
        d_alb[c] += dF0 * q.m;
      }
      if (q.l1 != q.l0) d_lvl += g_sp[c] * (s1v[c] - s0v[c]);
    }
    if (A.part != 2) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        if (A.g_albedo_mul_a) d_alb[c] += (A.g_albedo_mul_a[e + c * cs] * gscale) * A.g_albedo_mul_b[e + c * cs];
        A.d_albedo[e + c * cs] = d_alb[c];
      }
      float add_r = A.g_roughness_add ? A.g_roughness_add[p] : 0.0f, add_m = A.g_metallic_add ? A.g_metallic_add[p] : 0.0f;
      if (A.lamb_mask) {  // stage2_loss_bwd_kernel's two lines
        const float m = A.lamb_mask[p], cnt = A.lamb_acc4[3];
        add_r += -m / cnt * (0.001f * gscale);
        add_m += m / cnt * (0.001f * gscale);
      }
      if (A.d_metallic) A.d_metallic[p] = d_m + add_m;
      const float d_r = d_fgx * q.dfgx_dv + d_fgy * q.dfgy_dv + (q.lvl_inside ? d_lvl * q.dmdr : 0.0f);
      A.d_roughness[p] = (d_r + add_r) * A.rough_scale;
    }
  }
  if (A.part == 1) continue;  // the material gradients only (uniform over the grid)
  // ---- light textures (wave-uniform control flow from here on) ----
#ifdef GIGS_DIAG
#define GIGS_ABLATED(bit) (A.ablate & (bit))
#else
#define GIGS_ABLATED(bit) 0  // the shipped kernel has no result-changing switch
#endif
  if (A.d_diffuse && !GIGS_ABLATED(1)) {
    float* base = A.lds_diffuse_off >= 0 ? s_lds + A.lds_diffuse_off : A.d_diffuse;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int idx = (live && q.has_d) ? q.td.idx[k] : -1;
      const float w = idx >= 0 ? q.td.w[k] * q.occ : 0.0f;
      run_add3<false>(base + 3 * (size_t)max(idx, 0), idx, g_dl[0] * w, g_dl[1] * w, g_dl[2] * w);
    }
  }
  if (!GIGS_ABLATED(2)) {
    const float wl = (q.l1 != q.l0) ? (1 - q.lf) : 1.0f;
    float* t0 = A.lds_spec_off[q.l0] >= 0 ? s_lds + A.lds_spec_off[q.l0] : A.d_spec[q.l0];
    float* t1 = A.lds_spec_off[q.l1] >= 0 ? s_lds + A.lds_spec_off[q.l1] : A.d_spec[q.l1];
    if (!A.d_spec[q.l0]) t0 = nullptr;
    if (!A.d_spec[q.l1]) t1 = nullptr;
    // diagnostic (GIGS_ABLATE bit 2 / 3): drop the adds that go to LDS-resident / to global levels
#ifdef GIGS_DIAG
    if (((A.ablate & 4) && A.lds_spec_off[q.l0] >= 0) || ((A.ablate & 8) && A.lds_spec_off[q.l0] < 0)) t0 = nullptr;
    if (((A.ablate & 4) && A.lds_spec_off[q.l1] >= 0) || ((A.ablate & 8) && A.lds_spec_off[q.l1] < 0)) t1 = nullptr;
#endif
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int idx = (live && q.has0 && t0) ? q.t0.idx[k] : -1;
      const float w = idx >= 0 ? q.t0.w[k] * wl : 0.0f;
      run_add3<false>(t0 ? t0 + 3 * (size_t)max(idx, 0) : nullptr, idx >= 0 ? (q.l0 << 24) | idx : -1, g_sp[0] * w, g_sp[1] * w, g_sp[2] * w);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int idx = (live && q.l1 != q.l0 && q.has1 && t1) ? q.t1.idx[k] : -1;
      const float w = idx >= 0 ? q.t1.w[k] * q.lf : 0.0f;
      run_add3<false>(t1 ? t1 + 3 * (size_t)max(idx, 0) : nullptr, idx >= 0 ? (q.l1 << 24) | idx : -1, g_sp[0] * w, g_sp[1] * w, g_sp[2] * w);
    }
  }
  }  // chunk loop
  if (A.lds_total > 0) {
    __syncthreads();
    if (A.lds_diffuse_off >= 0 && A.d_diffuse) {
      const int n = 6 * A.diffuse_res * A.diffuse_res * 3;
      for (int i = threadIdx.x; i < n; i += kShadeBwdBlock) {
        const float val = s_lds[A.lds_diffuse_off + i];
        if (val != 0.0f) atomicAdd(A.d_diffuse + i, val);
      }
    }
    for (int l = 0; l < A.L; l++) {
      if (A.lds_spec_off[l] < 0 || !A.d_spec[l]) continue;
      const int n = 6 * A.spec_res[l] * A.spec_res[l] * 3;
      for (int i = threadIdx.x; i < n; i += kShadeBwdBlock) {
        const float val = s_lds[A.lds_spec_off[l] + i];
        if (val != 0.0f) atomicAdd(A.d_spec[l] + i, val);
      }
    }
  }
}

}  // namespace gigs

// ------------------------------------------------------------------------------------------
// C ABI (declared in include/gigs_hip.h)
// ------------------------------------------------------------------------------------------
#include "../../include/gigs_hip.h"

extern "C" {

int gigs_internal_fail(int code, const char* msg);  // api.hip
void gigs_internal_stage_begin(int stage, void* stream, void** token);
void gigs_internal_stage_end(void* token);
const gigs::Options* gigs_internal_options(const gigs_ctx* ctx);  // the context's options (NULL = the defaults)

#define PBR_CHECK_LAUNCH()                                        \
  do {                                                            \
    if (hipGetLastError() != hipSuccess) return gigs_internal_fail(GIGS_ERR_HIP, "pbr kernel launch failed"); \
  } while (0)

int gigs_diffuse_cubemap_fwd(int res, const float* cubemap, float* out, void* stream) {
  if (res <= 0 || !cubemap || !out) return gigs_internal_fail(GIGS_ERR_INVALID, "diffuse_cubemap_fwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const float4* table = gigs::texel_table(res, s);
  if (!table) return gigs_internal_fail(GIGS_ERR_HIP, "texel table");
  void* tok; gigs_internal_stage_begin(16, stream, &tok);
  hipLaunchKernelGGL(gigs::diffuse_cubemap_kernel<false>, dim3((6 * res * res + 3) / 4), dim3(256), 0, s, res, table, cubemap, out);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_diffuse_cubemap_bwd(int res, const float* grad_out, float* grad_cubemap, void* stream) {
  if (res <= 0 || !grad_out || !grad_cubemap) return gigs_internal_fail(GIGS_ERR_INVALID, "diffuse_cubemap_bwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const float4* table = gigs::texel_table(res, s);
  if (!table) return gigs_internal_fail(GIGS_ERR_HIP, "texel table");
  void* tok; gigs_internal_stage_begin(17, stream, &tok);
  hipLaunchKernelGGL(gigs::diffuse_cubemap_kernel<true>, dim3((6 * res * res + 3) / 4), dim3(256), 0, s, res, table, grad_out, grad_cubemap);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_specular_bounds(int res, float costheta_cutoff, float* bounds, void* stream) {
  if (res <= 0 || !bounds) return gigs_internal_fail(GIGS_ERR_INVALID, "specular_bounds: bad argument");
  hipLaunchKernelGGL(gigs::specular_bounds_kernel, dim3((6 * res * res + 63) / 64), dim3(64), 0, (hipStream_t)stream, res, costheta_cutoff, bounds);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_specular_cubemap_fwd(int res, const float* cubemap, const float* bounds, float roughness,
                              float costheta_cutoff, float* out, void* stream) {
  if (res <= 0 || !cubemap || !bounds || !out) return gigs_internal_fail(GIGS_ERR_INVALID, "specular_cubemap_fwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const float4* table = gigs::texel_table(res, s);
  if (!table) return gigs_internal_fail(GIGS_ERR_HIP, "texel table");
  void* tok; gigs_internal_stage_begin(16, stream, &tok);
  hipLaunchKernelGGL(gigs::specular_cubemap_kernel<false>, dim3((6 * res * res + 3) / 4), dim3(256), 0, s, res, table, cubemap, bounds, roughness, costheta_cutoff, out);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_specular_cubemap_bwd(int res, const float* bounds, const float* grad_out, float roughness,
                              float costheta_cutoff, float* grad_cubemap, void* stream) {
  if (res <= 0 || !bounds || !grad_out || !grad_cubemap) return gigs_internal_fail(GIGS_ERR_INVALID, "specular_cubemap_bwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const float4* table = gigs::texel_table(res, s);
  if (!table) return gigs_internal_fail(GIGS_ERR_HIP, "texel table");
  void* tok; gigs_internal_stage_begin(17, stream, &tok);
  hipLaunchKernelGGL(gigs::specular_cubemap_kernel<true>, dim3((6 * res * res + 3) / 4), dim3(256), 0, s, res, table, grad_out, bounds, roughness, costheta_cutoff, grad_cubemap);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_specular_weights(int res, const float* bounds, const uint32_t* offsets, float roughness,
                          float costheta_cutoff, int swap_roles, float* weights, void* stream) {
  if (res <= 0 || !bounds || !offsets || !weights) return gigs_internal_fail(GIGS_ERR_INVALID, "specular_weights: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const float4* table = gigs::texel_table(res, s);
  if (!table) return gigs_internal_fail(GIGS_ERR_HIP, "texel table");
  const dim3 grid((6 * res * res + 3) / 4);
  if (swap_roles)
    hipLaunchKernelGGL(gigs::specular_weights_kernel<true>, grid, dim3(256), 0, s, res, table, bounds, offsets, roughness, costheta_cutoff, weights);
  else
    hipLaunchKernelGGL(gigs::specular_weights_kernel<false>, grid, dim3(256), 0, s, res, table, bounds, offsets, roughness, costheta_cutoff, weights);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_specular_weights_divide(int res, const float* bounds, const uint32_t* offsets, const float* weights,
                                 const float* texel_divisor, float* out_weights, void* stream) {
  if (res <= 0 || !bounds || !offsets || !weights || !texel_divisor || !out_weights)
    return gigs_internal_fail(GIGS_ERR_INVALID, "specular_weights_divide: bad argument");
  hipLaunchKernelGGL(gigs::specular_divide_weights_kernel, dim3((6 * res * res + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     res, bounds, offsets, weights, texel_divisor, out_weights);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_specular_cubemap_fwd_w(gigs_ctx* ctx, int res, const float* cubemap, const float* bounds, const uint32_t* offsets,
                                const float* weights, int avg_window, float* out, float* wsum_out, void* stream) {
  const gigs::Options& o = *gigs_internal_options(ctx);
  if (res <= 0 || !cubemap || !bounds || !offsets || !weights || !out) return gigs_internal_fail(GIGS_ERR_INVALID, "specular_cubemap_fwd_w: bad argument");
  hipStream_t s = (hipStream_t)stream;
  void* tok; gigs_internal_stage_begin(16, stream, &tok);
  if (wsum_out)
    gigs::launch_specular_apply<false, true>(o, res, avg_window, cubemap, bounds, offsets, weights, out, wsum_out, s);
  else
    gigs::launch_specular_apply<false, false>(o, res, avg_window, cubemap, bounds, offsets, weights, out, nullptr, s);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_specular_cubemap_bwd_w(gigs_ctx* ctx, int res, const float* bounds, const uint32_t* offsets, const float* weights_swapped,
                                int avg_window, const float* grad_out, int grad_is_rgb, float* grad_cubemap, void* stream) {
  const gigs::Options& o = *gigs_internal_options(ctx);
  if (res <= 0 || !bounds || !offsets || !weights_swapped || !grad_out || !grad_cubemap) return gigs_internal_fail(GIGS_ERR_INVALID, "specular_cubemap_bwd_w: bad argument");
  hipStream_t s = (hipStream_t)stream;
  void* tok; gigs_internal_stage_begin(17, stream, &tok);
  if (grad_is_rgb)
    gigs::launch_specular_apply<true, true>(o, res, avg_window, grad_out, bounds, offsets, weights_swapped, grad_cubemap, nullptr, s);
  else
    gigs::launch_specular_apply<true, false>(o, res, avg_window, grad_out, bounds, offsets, weights_swapped, grad_cubemap, nullptr, s);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_specular_cubemap_multi_w(gigs_ctx* ctx, int n_levels, const gigs_spec_level* levels, int backward, void* stream) {
  const gigs::Options& o = *gigs_internal_options(ctx);
  if (n_levels <= 0 || n_levels > 8 || !levels) return gigs_internal_fail(GIGS_ERR_INVALID, "specular_cubemap_multi_w: bad level count");
  gigs::SpecLevels L;
  L.n = n_levels;
  int blocks = 0;
  for (int i = 0; i < n_levels; i++) {
    const gigs_spec_level& a = levels[i];
    if (a.res <= 0 || !a.src || !a.bounds || !a.offsets || !a.weights || !a.dst || (!backward && !a.wsum))
      return gigs_internal_fail(GIGS_ERR_INVALID, "specular_cubemap_multi_w: bad level");
    gigs::SpecLevel& v = L.lv[i];
    v.N = a.res; v.lanes = gigs::spec_lanes_for(o, a.avg_window); v.block_begin = blocks;
    v.src = a.src; v.bounds = a.bounds; v.offsets = a.offsets; v.W = a.weights; v.dst = a.dst; v.wsum_out = a.wsum;
    const int total = 6 * a.res * a.res;
    const int waves = v.lanes == 64 ? total : (total + (64 / v.lanes) - 1) / (64 / v.lanes);
    blocks += (waves + 3) / 4;
  }
  void* tok; gigs_internal_stage_begin(backward ? 17 : 16, stream, &tok);
  if (backward) hipLaunchKernelGGL(gigs::specular_apply_multi_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
  else hipLaunchKernelGGL(gigs::specular_apply_multi_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_cubemap_mip_fwd(int res_out, int channels, const float* in, float* out, void* stream) {
  if (res_out <= 0 || channels <= 0 || !in || !out) return gigs_internal_fail(GIGS_ERR_INVALID, "cubemap_mip_fwd: bad argument");
  void* tok; gigs_internal_stage_begin(16, stream, &tok);
  hipLaunchKernelGGL(gigs::cubemap_mip_fwd_kernel, dim3((6 * res_out * res_out * channels + 255) / 256), dim3(256), 0, (hipStream_t)stream, res_out, channels, in, out);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_cubemap_mip_bwd(int res_out, const float* dout, float* din, void* stream) {
  if (res_out <= 0 || !dout || !din) return gigs_internal_fail(GIGS_ERR_INVALID, "cubemap_mip_bwd: bad argument");
  const int res = 2 * res_out;
  void* tok; gigs_internal_stage_begin(17, stream, &tok);
  hipLaunchKernelGGL(gigs::cubemap_mip_bwd_kernel, dim3((6 * res * res + 255) / 256), dim3(256), 0, (hipStream_t)stream, res_out, dout, din, (const float*)nullptr,
                     (const float*)nullptr);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_cubemap_mip_bwd_add(int res_out, const float* dout, const float* add, float* din, void* stream) {
  if (res_out <= 0 || !dout || !add || !din) return gigs_internal_fail(GIGS_ERR_INVALID, "cubemap_mip_bwd_add: bad argument");
  const int res = 2 * res_out;
  void* tok; gigs_internal_stage_begin(17, stream, &tok);
  hipLaunchKernelGGL(gigs::cubemap_mip_bwd_kernel, dim3((6 * res * res + 255) / 256), dim3(256), 0, (hipStream_t)stream, res_out, dout, din, add,
                     (const float*)nullptr);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_cubemap_mip_bwd_add2(int res_out, const float* dout, const float* dout2, const float* add, float* din, void* stream) {
  if (res_out <= 0 || !dout || !din) return gigs_internal_fail(GIGS_ERR_INVALID, "cubemap_mip_bwd_add2: bad argument");
  const int res = 2 * res_out;
  void* tok; gigs_internal_stage_begin(17, stream, &tok);
  hipLaunchKernelGGL(gigs::cubemap_mip_bwd_kernel, dim3((6 * res * res + 255) / 256), dim3(256), 0, (hipStream_t)stream, res_out, dout, din, add,
                     dout2);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_cube_texture_fwd(int res, const float* cubemap, int n, const float* dirs, float* out, int planar,
                          void* stream) {
  if (res <= 0 || n < 0 || !cubemap || (n > 0 && (!dirs || !out)))
    return gigs_internal_fail(GIGS_ERR_INVALID, "cube_texture_fwd: bad argument");
  if (n == 0) return 0;
  void* tok; gigs_internal_stage_begin(16, stream, &tok);
  hipLaunchKernelGGL(gigs::cube_texture_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, res,
                     cubemap, n, dirs, out, planar);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_latlong_to_cubemap(int res_y, int res_x, int lat_h, int lat_w, int channels, const float* latlong, float* cubemap,
                            void* stream) {
  if (res_y <= 0 || res_x <= 0 || lat_h <= 0 || lat_w <= 0 || channels <= 0 || !latlong || !cubemap)
    return gigs_internal_fail(GIGS_ERR_INVALID, "latlong_to_cubemap: bad argument");
  const int n = 6 * res_y * res_x;
  void* tok; gigs_internal_stage_begin(16, stream, &tok);
  hipLaunchKernelGGL(gigs::latlong_to_cubemap_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, res_y, res_x,
                     lat_h, lat_w, channels, latlong, cubemap);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_cube_texture_bwd(int res, int n, const float* dirs, const float* g_out, float* d_cubemap, int planar,
                          void* stream) {
  if (res <= 0 || n < 0 || !d_cubemap || (n > 0 && (!dirs || !g_out)))
    return gigs_internal_fail(GIGS_ERR_INVALID, "cube_texture_bwd: bad argument");
  if (n == 0) return 0;
  void* tok; gigs_internal_stage_begin(17, stream, &tok);
  hipLaunchKernelGGL(gigs::cube_texture_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, res, n,
                     dirs, g_out, d_cubemap, planar);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_cube_taps(int res, int n, const float* dirs, int* idx, float* w, void* stream) {
  if (res <= 0 || n < 0 || (n > 0 && (!dirs || !idx || !w))) return gigs_internal_fail(GIGS_ERR_INVALID, "cube_taps: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(gigs::cube_taps_export_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, res, n, dirs, idx, w);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_cube_texture_bwd_gather(int res, int n, int planar, const int* offsets, const int* ent_sample, const float* ent_w,
                                 int heavy, int n_heavy, const int* heavy_ids, const float* g_out, float* d_cubemap,
                                 void* stream) {
  if (res <= 0 || n <= 0 || !offsets || !ent_sample || !ent_w || !g_out || !d_cubemap || heavy < 0 || n_heavy < 0 ||
      (n_heavy > 0 && !heavy_ids))
    return gigs_internal_fail(GIGS_ERR_INVALID, "cube_texture_bwd_gather: bad argument");
  const int n_tex = 6 * res * res;
  const int light_blocks = (n_tex + 255) / 256, heavy_blocks = (n_heavy + 3) / 4;
  void* tok; gigs_internal_stage_begin(17, stream, &tok);
  hipLaunchKernelGGL(gigs::cube_texture_bwd_gather_kernel, dim3(light_blocks + heavy_blocks), dim3(256), 0, (hipStream_t)stream,
                     n_tex, offsets, ent_sample, ent_w, g_out, n, planar, d_cubemap, heavy, light_blocks, heavy_ids, n_heavy);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

static int fill_shade(gigs::ShadeArgs& A, int H, int W, const float* normals, const float* view_dirs,
                      const float* albedo, const float* roughness, const uint8_t* mask,
                      const float* occlusion, const float* metallic, const float* background,
                      const float* diffuse, int diffuse_res, int n_levels, const float* const* spec,
                      const int* spec_res, const float* lut, int lut_w, int lut_h, int tone, int gamma) {
  if (H <= 0 || W <= 0 || !normals || !view_dirs || !albedo || !roughness || !mask || !diffuse || !spec || !spec_res || !lut)
    return gigs_internal_fail(GIGS_ERR_INVALID, "shade: null required input");
  if (n_levels < 2 || n_levels > 8 || diffuse_res <= 0 || lut_w <= 0 || lut_h <= 0)
    return gigs_internal_fail(GIGS_ERR_INVALID, "shade: needs 2..8 specular levels and positive texture sizes");
  memset(&A, 0, sizeof(A));
  A.H = H; A.W = W; A.normals = normals; A.view_dirs = view_dirs; A.albedo = albedo; A.roughness = roughness;
  A.mask = mask; A.occlusion = occlusion; A.metallic = metallic; A.background = background;
  A.diffuse = diffuse; A.diffuse_res = diffuse_res; A.L = n_levels;
  for (int i = 0; i < n_levels; i++) {
    if (!spec[i] || spec_res[i] <= 0) return gigs_internal_fail(GIGS_ERR_INVALID, "shade: bad specular level");
    A.spec[i] = spec[i];
    A.spec_res[i] = spec_res[i];
  }
  A.lut = lut; A.lut_w = lut_w; A.lut_h = lut_h; A.tone = tone; A.gamma = gamma;
#ifdef GIGS_DIAG
  const char* ab = getenv("GIGS_ABLATE");
  A.ablate = ab ? atoi(ab) : 0;
#endif
  A.ps = 3; A.cs = 1; A.rough_scale = 1.0f; A.rough_bias = 0.0f;
  return 0;
}

static int apply_shade_ext(gigs::ShadeArgs& A, const gigs_shade_ext* ext, bool backward) {
  if (!ext) return 0;
  if (ext->planar) { A.ps = 1; A.cs = A.H * A.W; }
  A.rough_scale = ext->rough_scale; A.rough_bias = ext->rough_bias;
  if (!backward) {
    A.out_F0 = ext->out_F0; A.out_linear = ext->out_linear; A.out_roughness = ext->out_roughness;
  } else {
    if ((ext->g_albedo_mul_a == nullptr) != (ext->g_albedo_mul_b == nullptr))
      return gigs_internal_fail(GIGS_ERR_INVALID, "shade_bwd: g_albedo_mul_a/b must be given together");
    A.g_albedo_mul_a = ext->g_albedo_mul_a; A.g_albedo_mul_b = ext->g_albedo_mul_b;
    A.g_roughness_add = ext->g_roughness_add; A.g_metallic_add = ext->g_metallic_add;
    if ((ext->lamb_mask == nullptr) != (ext->lamb_acc4 == nullptr))
      return gigs_internal_fail(GIGS_ERR_INVALID, "shade_bwd: lamb_mask / lamb_acc4 must be given together");
    A.g_scale = ext->g_scale; A.lamb_mask = ext->lamb_mask; A.lamb_acc4 = ext->lamb_acc4;
  }
  return 0;
}

int gigs_shade_fwd(int H, int W, const float* normals, const float* view_dirs, const float* albedo,
                   const float* roughness, const uint8_t* mask, const float* occlusion,
                   const float* metallic, const float* background, const float* diffuse, int diffuse_res,
                   int n_levels, const float* const* spec, const int* spec_res, const float* lut,
                   int lut_w, int lut_h, int tone, int gamma, float* render_rgb, float* diffuse_rgb,
                   float* specular_rgb, float* diffuse_light, void* stream) {
  return gigs_shade_fwd_ex(nullptr, H, W, normals, view_dirs, albedo, roughness, mask, occlusion, metallic, background, diffuse,
                           diffuse_res, n_levels, spec, spec_res, lut, lut_w, lut_h, tone, gamma, render_rgb,
                           diffuse_rgb, specular_rgb, diffuse_light, nullptr, stream);
}

int gigs_shade_fwd_ex(gigs_ctx* ctx, int H, int W, const float* normals, const float* view_dirs, const float* albedo,
                      const float* roughness, const uint8_t* mask, const float* occlusion,
                      const float* metallic, const float* background, const float* diffuse, int diffuse_res,
                      int n_levels, const float* const* spec, const int* spec_res, const float* lut,
                      int lut_w, int lut_h, int tone, int gamma, float* render_rgb, float* diffuse_rgb,
                      float* specular_rgb, float* diffuse_light, const gigs_shade_ext* ext, void* stream) {
  (void)ctx;  // the forward reads no option; the parameter keeps the two _ex entries alike
  gigs::ShadeArgs A;
  const int rc = fill_shade(A, H, W, normals, view_dirs, albedo, roughness, mask, occlusion, metallic, background,
                            diffuse, diffuse_res, n_levels, spec, spec_res, lut, lut_w, lut_h, tone, gamma);
  if (rc) return rc;
  if (!render_rgb || (!ext && (!diffuse_rgb || !specular_rgb || !diffuse_light)))
    return gigs_internal_fail(GIGS_ERR_INVALID, "shade_fwd: null output");
  if (apply_shade_ext(A, ext, false)) return GIGS_ERR_INVALID;
  A.render_rgb = render_rgb; A.diffuse_rgb = diffuse_rgb; A.specular_rgb = specular_rgb; A.diffuse_light = diffuse_light;
  void* tok; gigs_internal_stage_begin(14, stream, &tok);
  hipLaunchKernelGGL(gigs::shade_fwd_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, A);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

int gigs_shade_bwd(int H, int W, const float* normals, const float* view_dirs, const float* albedo,
                   const float* roughness, const uint8_t* mask, const float* occlusion,
                   const float* metallic, const float* diffuse, int diffuse_res, int n_levels,
                   const float* const* spec, const int* spec_res, const float* lut, int lut_w, int lut_h,
                   int tone, int gamma, const float* g_render, const float* g_diffuse_rgb,
                   const float* g_specular_rgb, const float* g_diffuse_light, float* d_albedo,
                   float* d_roughness, float* d_metallic, float* d_diffuse, float* const* d_spec,
                   void* stream) {
  return gigs_shade_bwd_ex(nullptr, H, W, normals, view_dirs, albedo, roughness, mask, occlusion, metallic, diffuse, diffuse_res,
                           n_levels, spec, spec_res, lut, lut_w, lut_h, tone, gamma, g_render, g_diffuse_rgb,
                           g_specular_rgb, g_diffuse_light, d_albedo, d_roughness, d_metallic, d_diffuse, d_spec, nullptr,
                           stream);
}

int gigs_shade_bwd_ex(gigs_ctx* ctx, int H, int W, const float* normals, const float* view_dirs, const float* albedo,
                      const float* roughness, const uint8_t* mask, const float* occlusion,
                      const float* metallic, const float* diffuse, int diffuse_res, int n_levels,
                      const float* const* spec, const int* spec_res, const float* lut, int lut_w, int lut_h,
                      int tone, int gamma, const float* g_render, const float* g_diffuse_rgb,
                      const float* g_specular_rgb, const float* g_diffuse_light, float* d_albedo,
                      float* d_roughness, float* d_metallic, float* d_diffuse, float* const* d_spec,
                      const gigs_shade_ext* ext, void* stream) {
  gigs::ShadeArgs A;
  const int rc = fill_shade(A, H, W, normals, view_dirs, albedo, roughness, mask, occlusion, metallic, nullptr,
                            diffuse, diffuse_res, n_levels, spec, spec_res, lut, lut_w, lut_h, tone, gamma);
  if (rc) return rc;
  if (apply_shade_ext(A, ext, true)) return GIGS_ERR_INVALID;
  A.part = (ext && ext->part >= 0 && ext->part <= 2) ? ext->part : 0;
  if (A.part != 2 && (!d_albedo || !d_roughness)) return gigs_internal_fail(GIGS_ERR_INVALID, "shade_bwd: null output");
  if (A.part == 1) { d_diffuse = nullptr; d_spec = nullptr; }
  A.g_render = g_render; A.g_diffuse_rgb = g_diffuse_rgb; A.g_specular_rgb = g_specular_rgb; A.g_diffuse_light = g_diffuse_light;
  A.d_albedo = d_albedo; A.d_roughness = d_roughness; A.d_metallic = d_metallic; A.d_diffuse = d_diffuse;
  for (int i = 0; i < n_levels; i++) A.d_spec[i] = d_spec ? d_spec[i] : nullptr;
  // LDS plan: the diffuse map first, then specular levels from the coarsest up while they fit
  int used = 0;
  A.lds_diffuse_off = -1;
  for (int i = 0; i < 8; i++) A.lds_spec_off[i] = -1;
  const int n_dd = 6 * diffuse_res * diffuse_res * 3;
  const gigs::Options& o = *gigs_internal_options(ctx);
  // gigs_options.shade_lds_floats: tuning knob (smaller budget -> more workgroups per CU)
  const int lds_budget = o.shade_lds_floats < 0 ? 0 : (o.shade_lds_floats > gigs::kShadeLdsBudget ? gigs::kShadeLdsBudget : o.shade_lds_floats);
  if (d_diffuse && n_dd <= lds_budget) { A.lds_diffuse_off = 0; used = n_dd; }
  for (int i = n_levels - 1; i >= 0; i--) {
    const int n = 6 * spec_res[i] * spec_res[i] * 3;
    if (A.d_spec[i] && used + n <= lds_budget) { A.lds_spec_off[i] = used; used += n; }
  }
  A.lds_total = used;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gigs::shade_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            gigs::kShadeLdsBudget * (int)sizeof(float)) != hipSuccess)
      return gigs_internal_fail(GIGS_ERR_HIP, "shade_bwd: cannot raise the dynamic LDS limit");
    attr_set = true;
  }
  void* tok; gigs_internal_stage_begin(15, stream, &tok);
  const int n_chunks = (H * W + gigs::kShadeBwdBlock - 1) / gigs::kShadeBwdBlock;
  static const int n_cus = [] {  // one workgroup per CU (120 KB of LDS each); gigs_options.shade_bwd_blocks overrides
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
  }();
  const int max_blocks = o.shade_bwd_blocks > 0 ? o.shade_bwd_blocks : n_cus;
  // part 1 (materials only) keeps no LDS accumulators: one workgroup per chunk instead of the persistent grid
  const int blocks = (A.part == 1 || n_chunks < max_blocks) ? n_chunks : max_blocks;
  hipLaunchKernelGGL(gigs::shade_bwd_kernel, dim3(blocks), dim3(gigs::kShadeBwdBlock), (size_t)used * sizeof(float),
                     (hipStream_t)stream, A);
  gigs_internal_stage_end(tok);
  PBR_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
