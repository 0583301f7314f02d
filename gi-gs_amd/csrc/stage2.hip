// stage2.hip -- the tensor glue of one stage-2 ("PBR + indirect") iteration as three kernels.
//
// The reference expresses these steps as a few dozen torch elementwise / reduction ops in
// gaussian_renderer/__init__.py:157-199 (G-buffer post-processing) and train.py:382-402 (loss); on
// an MI355X each of those ops is a 3-10 us launch over 0.6 M pixels, ~1.8 ms per iteration in total.
// Here they are one pass each.  Results agree with the torch formulation to fp32 rounding (the same
// formulas; tests/test_gpu_pbr.py compares them).
//
//   gbuffer_post_kernel   masks, F.normalize, 3x3 median and the rotation of the shading normal into
//                         view space, for normal_map and out_normal_view (no gradient: train.py detaches
//                         both before they are used in stage 2)
//   stage2_loss_fwd       IRR -> sRGB -> 3x3 median -> + direct -> L1 vs ground truth, plus the masked
//                         sums of the "lamb" regulariser; block-reduced, 4 atomics per workgroup
//   stage2_loss_bwd       the gradient of that loss w.r.t. render_direct, IRR (through the median's tap
//                         selection and the sRGB curve) and the roughness / metallic maps
// All planes are [C,H,W] fp32, the rasterizer's layout.  HBM-bound streaming kernels.
#include "../../include/gigs_hip.h"
#include "gigs_common.h"
#include "pixel_ops.h"

namespace gigs {

// F.normalize(v, dim=0) where |v| > 0, v itself otherwise (gaussian_renderer/__init__.py:160-163, 191-194);
// a NaN vector stays NaN (|v| > 0 is false).
__device__ __forceinline__ v3 normalize_where(v3 v) {
  const float n = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
  if (!(n > 0.0f)) return v;
  const float d = fmaxf(n, 1e-12f);
  return {v.x / d, v.y / d, v.z / d};
}

// per-channel 3x3 median of normalize_where(src) with zero padding; NaN window -> NaN (as median3x3_kernel)
__device__ __forceinline__ v3 median_of_normalized(const float* __restrict__ src, int H, int W, int y, int x) {
  const size_t HW = (size_t)H * W;
  float t0[9], t1[9], t2[9];
  bool n0 = false, n1 = false, n2 = false;
  int k = 0;
#pragma unroll
  for (int dy = -1; dy <= 1; dy++)
#pragma unroll
    for (int dx = -1; dx <= 1; dx++, k++) {
      const int yy = y + dy, xx = x + dx;
      v3 v = {0.0f, 0.0f, 0.0f};
      if (!(yy < 0 || yy >= H || xx < 0 || xx >= W)) {
        const size_t q = (size_t)yy * W + xx;
        v = normalize_where({src[q], src[HW + q], src[2 * HW + q]});
      }
      t0[k] = v.x; t1[k] = v.y; t2[k] = v.z;
      n0 |= v.x != v.x; n1 |= v.y != v.y; n2 |= v.z != v.z;
    }
  const float nan = __builtin_nanf("");
  return {n0 ? nan : median9(t0), n1 ? nan : median9(t1), n2 ? nan : median9(t2)};
}

__global__ void __launch_bounds__(256)
gbuffer_post_kernel(int H, int W, const float* __restrict__ normal_map, const float* __restrict__ out_normal_view,
                    const float* __restrict__ vm, float* __restrict__ normals_view, uint8_t* __restrict__ mask_u8,
                    float* __restrict__ mask_f, float* __restrict__ onv) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const size_t HW = (size_t)H * W, p = (size_t)y * W + x;
  const bool mk = normal_map[p] != 0.0f && normal_map[HW + p] != 0.0f && normal_map[2 * HW + p] != 0.0f;
  if (mask_u8) mask_u8[p] = mk ? 1 : 0;
  if (mask_f) mask_f[p] = mk ? 1.0f : 0.0f;
  const v3 n = median_of_normalized(normal_map, H, W, y, x);
  // -(n @ R), R = viewmatrix[:3, :3] of the row-major 4x4 tensor
  normals_view[p] = -(n.x * vm[0] + n.y * vm[4] + n.z * vm[8]);
  normals_view[HW + p] = -(n.x * vm[1] + n.y * vm[5] + n.z * vm[9]);
  normals_view[2 * HW + p] = -(n.x * vm[2] + n.y * vm[6] + n.z * vm[10]);
  const v3 o = median_of_normalized(out_normal_view, H, W, y, x);
  onv[p] = o.x; onv[HW + p] = o.y; onv[2 * HW + p] = o.z;
}

// ---- backward of the normal post-processing (stage 1: train.py:327-328 differentiates through
// gaussian_renderer/__init__.py:160-186) ---------------------------------------------------------------------------
// normals_view = -(median3x3(normalize_where(normal_map)) @ R).  Pass 1, one lane per output pixel: rotate the
// incoming gradient back, g_med[c] = -sum_j R[c][j] g[j], and hand it to the tap the median selected (first tap in
// row-major order equal to the median, as median3x3_bwd_kernel; padding taps and NaN windows drop it).  Pass 2, one
// lane per source pixel: the derivative of normalize_where.
__global__ void __launch_bounds__(256)
gbuffer_post_bwd_scatter_kernel(int H, int W, const float* __restrict__ normal_map, const float* __restrict__ vm,
                                const float* __restrict__ g_nv, float* __restrict__ g_hat) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const size_t HW = (size_t)H * W, p = (size_t)y * W + x;
  const float g0 = g_nv[p], g1 = g_nv[HW + p], g2 = g_nv[2 * HW + p];
  float gm[3];
#pragma unroll
  for (int c = 0; c < 3; c++) gm[c] = -(vm[4 * c] * g0 + vm[4 * c + 1] * g1 + vm[4 * c + 2] * g2);
  if (gm[0] == 0.0f && gm[1] == 0.0f && gm[2] == 0.0f) return;
  float t[3][9];
  bool nan[3] = {false, false, false};
  int k = 0;
#pragma unroll
  for (int dy = -1; dy <= 1; dy++)
#pragma unroll
    for (int dx = -1; dx <= 1; dx++, k++) {
      const int yy = y + dy, xx = x + dx;
      v3 v = {0.0f, 0.0f, 0.0f};
      if (!(yy < 0 || yy >= H || xx < 0 || xx >= W)) {
        const size_t q = (size_t)yy * W + xx;
        v = normalize_where({normal_map[q], normal_map[HW + q], normal_map[2 * HW + q]});
      }
      t[0][k] = v.x; t[1][k] = v.y; t[2][k] = v.z;
      nan[0] |= v.x != v.x; nan[1] |= v.y != v.y; nan[2] |= v.z != v.z;
    }
#pragma unroll
  for (int c = 0; c < 3; c++) {
    if (nan[c] || gm[c] == 0.0f) continue;
    float srt[9];
#pragma unroll
    for (int j = 0; j < 9; j++) srt[j] = t[c][j];
    const float med = median9(srt);
    int sel = -1;
#pragma unroll
    for (int j = 8; j >= 0; j--)
      if (t[c][j] == med) sel = j;  // ends on the first match
    if (sel < 0) continue;
    const int yy = y + sel / 3 - 1, xx = x + sel % 3 - 1;
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;  // a zero-padding tap was the median
    atomicAdd(g_hat + c * HW + (size_t)yy * W + xx, gm[c]);
  }
}

__global__ void __launch_bounds__(256)
gbuffer_post_bwd_finish_kernel(size_t HW, const float* __restrict__ normal_map, const float* __restrict__ g_hat,
                               float* __restrict__ g_normal_map) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const float vx = normal_map[p], vy = normal_map[HW + p], vz = normal_map[2 * HW + p];
  const float gx = g_hat[p], gy = g_hat[HW + p], gz = g_hat[2 * HW + p];
  const float n2 = vx * vx + vy * vy + vz * vz;
  const float n = sqrtf(n2);
  float ox = gx, oy = gy, oz = gz;  // |v| == 0 (or NaN): torch.where passes v itself
  if (n > 0.0f) {
    if (n > 1e-12f) {
      const float s = (vx * gx + vy * gy + vz * gz) / (n2 * n);
      ox = gx / n - vx * s; oy = gy / n - vy * s; oz = gz / n - vz * s;
    } else {
      ox = gx / 1e-12f; oy = gy / 1e-12f; oz = gz / 1e-12f;
    }
  }
  g_normal_map[p] = ox; g_normal_map[HW + p] = oy; g_normal_map[2 * HW + p] = oz;
}

// normal_map_from_depth of gaussian_renderer/__init__.py:157-163: mask = (v != 0).all(0), v <- normalize_where(v)
__global__ void __launch_bounds__(256)
normalize_mask_kernel(size_t HW, const float* __restrict__ in, float* __restrict__ out, uint8_t* __restrict__ mask) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const v3 v = {in[p], in[HW + p], in[2 * HW + p]};
  if (mask) mask[p] = (v.x != 0.0f && v.y != 0.0f && v.z != 0.0f) ? 1 : 0;
  const v3 o = normalize_where(v);
  out[p] = o.x; out[HW + p] = o.y; out[2 * HW + p] = o.z;
}

// mask = (v != 0).all(0) of a [3,H,W] plane set as floats (gaussian_renderer/__init__.py:158; the masked TV's weight):
// torch's `!=` + `.all(0)` + `.float()` are a compare, a fill, a reduction and a cast
__global__ void __launch_bounds__(256)
nonzero_mask_kernel(size_t HW, const float* __restrict__ in, float* __restrict__ mask) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  mask[p] = (in[p] != 0.0f && in[HW + p] != 0.0f && in[2 * HW + p] != 0.0f) ? 1.0f : 0.0f;
}

// ---- loss -------------------------------------------------------------------------------------
// acc[0] = sum |render_rgb - gt|, acc[1] = sum (1 - roughness) * mask, acc[2] = sum metallic * mask,
// acc[3] = sum mask;  loss = acc0 / (3 H W) + 0.001 * (acc1 / acc3 + acc2 / acc3)   (train.py:396-402)
// linear_to_srgb is a powf per texel; the 3x3 median needs it for 9 taps per pixel and channel.  A workgroup
// (64 x 4 pixels) therefore converts its (64+2) x (4+2) halo tile once per channel into LDS (1.5 powf per pixel
// and channel instead of 9) and the taps are read from there.  Out-of-image taps are 0 (median_blur pads zeros).
constexpr int kTileW = 64, kTileH = 4, kHaloW = kTileW + 2, kHaloH = kTileH + 2;
constexpr int kAccSlots = 256;  // rows of partial sums (gigs_stage2_loss_fwd's acc buffer holds 4 + 4 * kAccSlots floats)
struct LossTaps {
  float v[9];
  bool has_nan;
};
__device__ __forceinline__ void srgb_tile(const float* __restrict__ irr, int H, int W, float (*s_t)[kHaloH][kHaloW]) {
  const size_t HW = (size_t)H * W;
  const int x0 = blockIdx.x * kTileW - 1, y0 = blockIdx.y * kTileH - 1;
  for (int i = threadIdx.x; i < 3 * kHaloH * kHaloW; i += 256) {
    const int c = i / (kHaloH * kHaloW), r = i - c * (kHaloH * kHaloW);
    const int ty = r / kHaloW, tx = r - ty * kHaloW;
    const int yy = y0 + ty, xx = x0 + tx;
    float v = 0.0f, d;
    if (!(yy < 0 || yy >= H || xx < 0 || xx >= W)) v = lin2srgb(irr[c * HW + (size_t)yy * W + xx], d);
    s_t[c][ty][tx] = v;
  }
  __syncthreads();
}
__device__ __forceinline__ LossTaps srgb_taps(const float (*s_c)[kHaloW], int ly, int lx) {
  LossTaps t;
  t.has_nan = false;
  int k = 0;
#pragma unroll
  for (int dy = 0; dy <= 2; dy++)
#pragma unroll
    for (int dx = 0; dx <= 2; dx++, k++) {
      const float s = s_c[ly + dy][lx + dx];
      t.v[k] = s;
      t.has_nan |= s != s;
    }
  return t;
}

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
  // wave sum by shuffles, then the 4 waves through LDS; result valid in thread 0
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[wave] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

__global__ void __launch_bounds__(256)
stage2_loss_fwd_kernel(int H, int W, const float* __restrict__ direct, const float* __restrict__ irr,
                       const float* __restrict__ gt, const float* __restrict__ mask_f,
                       const float* __restrict__ roughness, const float* __restrict__ metallic,
                       float* __restrict__ render_rgb, float* __restrict__ acc, float* __restrict__ d_direct_unit,
                       float* __restrict__ d_irr_unit) {
  __shared__ float s_red[4];
  __shared__ float s_t[3][kHaloH][kHaloW];
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const int x = blockIdx.x * kTileW + lx;
  const int y = blockIdx.y * kTileH + ly;
  const bool live = x < W && y < H;
  const size_t HW = (size_t)H * W, p = live ? (size_t)y * W + x : 0;
  float l1 = 0.0f, rs = 0.0f, ms = 0.0f, cnt = 0.0f;
  srgb_tile(irr, H, W, s_t);
  if (live) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      LossTaps t = srgb_taps(s_t[c], ly, lx);
      float sorted[9];
#pragma unroll
      for (int k = 0; k < 9; k++) sorted[k] = t.v[k];
      const float med = t.has_nan ? __builtin_nanf("") : median9(sorted);
      const float r = direct[c * HW + p] + med;
      if (render_rgb) render_rgb[c * HW + p] = r;
      const float diff = r - gt[c * HW + p];
      l1 += fabsf(diff);
      if (d_direct_unit) {
        // stage2_loss_bwd_kernel's work for a unit upstream gradient, on the tile and the median this pass already has
        const float gs = 1.0f / (3.0f * (float)H * (float)W);
        const float sgn = diff > 0.0f ? gs : (diff < 0.0f ? -gs : 0.0f);  // d|x| = sign(x), 0 at 0 and for NaN
        d_direct_unit[c * HW + p] = sgn;
        if (!(t.has_nan || sgn == 0.0f)) {
          // the whole gradient goes to the first tap (row-major) equal to the median (median3x3_bwd_kernel)
          int k = 0;
          bool routed = false;
          for (int dy = -1; dy <= 1 && !routed; dy++)
            for (int dx = -1; dx <= 1; dx++, k++) {
              if (t.v[k] == med) {
                const int yy = y + dy, xx = x + dx;
                if (!(yy < 0 || yy >= H || xx < 0 || xx >= W)) {
                  const size_t q = (size_t)yy * W + xx;
                  float d;
                  lin2srgb(irr[c * HW + q], d);
                  if (d != 0.0f) atomicAdd(d_irr_unit + c * HW + q, sgn * d);
                }
                routed = true;  // a padding tap selected: the gradient is dropped
                break;
              }
            }
        }
      }
    }
    const float m = mask_f[p];
    rs = (1.0f - roughness[p]) * m;
    ms = metallic[p] * m;
    cnt = m;
  }
  l1 = block_sum_256(l1, s_red);
  rs = block_sum_256(rs, s_red);
  ms = block_sum_256(ms, s_red);
  cnt = block_sum_256(cnt, s_red);
  // Thousands of workgroups adding to the same four floats serialise in L2 (measured: 0.1 ms of a 0.14 ms
  // kernel), so the partial sums are spread over kAccSlots rows and the finish kernel adds the rows up.
  if (threadIdx.x == 0) {
    float* row = acc + 4 * ((blockIdx.y * gridDim.x + blockIdx.x) % kAccSlots);
    atomicAdd(row + 0, l1);
    atomicAdd(row + 1, rs);
    atomicAdd(row + 2, ms);
    atomicAdd(row + 3, cnt);
  }
}

// one wave: column sums of the kAccSlots x 4 partial rows -> acc4 (kept for the backward) and the loss
__global__ void __launch_bounds__(64)
stage2_loss_finish_kernel(int H, int W, const float* __restrict__ rows, float* __restrict__ acc4, float* __restrict__ loss) {
  float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int r = threadIdx.x; r < kAccSlots; r += 64)
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] += rows[4 * r + k];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
  if (threadIdx.x == 0) {
    const float n = 3.0f * (float)H * (float)W;
    const float l1 = v[0] / n;
    const float lamb = v[1] / v[3] + v[2] / v[3];
    loss[0] = l1 + lamb * 0.001f;
#pragma unroll
    for (int k = 0; k < 4; k++) acc4[k] = v[k];
  }
}

__global__ void __launch_bounds__(256)
stage2_loss_bwd_kernel(int H, int W, const float* __restrict__ direct, const float* __restrict__ irr,
                       const float* __restrict__ gt, const float* __restrict__ mask_f, const float* __restrict__ acc,
                       const float* __restrict__ g_loss, float* __restrict__ d_direct, float* __restrict__ d_irr,
                       float* __restrict__ d_roughness, float* __restrict__ d_metallic) {
  __shared__ float s_t[3][kHaloH][kHaloW];
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const int x = blockIdx.x * kTileW + lx;
  const int y = blockIdx.y * kTileH + ly;
  srgb_tile(irr, H, W, s_t);
  if (x >= W || y >= H) return;
  const size_t HW = (size_t)H * W, p = (size_t)y * W + x;
  const float gl = g_loss ? g_loss[0] : 1.0f;
  const float gs = gl / (3.0f * (float)H * (float)W);
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float* src = irr + c * HW;
    LossTaps t = srgb_taps(s_t[c], ly, lx);
    float sorted[9];
#pragma unroll
    for (int k = 0; k < 9; k++) sorted[k] = t.v[k];
    const float med = t.has_nan ? __builtin_nanf("") : median9(sorted);
    const float diff = direct[c * HW + p] + med - gt[c * HW + p];
    const float s = diff > 0.0f ? gs : (diff < 0.0f ? -gs : 0.0f);  // d|x| = sign(x), 0 at 0 and for NaN
    d_direct[c * HW + p] = s;
    if (t.has_nan || s == 0.0f) continue;
    // the whole gradient goes to the first tap (row-major) equal to the median (median3x3_bwd_kernel)
    int k = 0;
    bool routed = false;
    for (int dy = -1; dy <= 1 && !routed; dy++)
      for (int dx = -1; dx <= 1; dx++, k++) {
        if (t.v[k] == med) {
          const int yy = y + dy, xx = x + dx;
          if (!(yy < 0 || yy >= H || xx < 0 || xx >= W)) {
            const size_t q = (size_t)yy * W + xx;
            float d;
            lin2srgb(src[q], d);
            if (d != 0.0f) atomicAdd(d_irr + c * HW + q, s * d);
          }
          routed = true;  // a padding tap selected: the gradient is dropped
          break;
        }
      }
  }
  const float m = mask_f[p], cnt = acc[3];
  d_roughness[p] = -m / cnt * (0.001f * gl);
  d_metallic[p] = m / cnt * (0.001f * gl);
}

// plain zero-fill (a kernel node rather than a memset node when the caller captures into a hipGraph)
__global__ void __launch_bounds__(256) zero_kernel(float* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.0f;
}
static void launch_zero(float* p, size_t n, hipStream_t s) {
  const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, s, p, n);
}
// two buffers, one node (the small one rides along: a node of its own costs ~5 us on the step's critical path)
__global__ void __launch_bounds__(256) zero2_kernel(float* __restrict__ a, size_t na, float* __restrict__ b, size_t nb) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += (size_t)gridDim.x * 256) {
    if (i < na) a[i] = 0.0f;
    else b[i - na] = 0.0f;
  }
}
static void launch_zero2(float* a, size_t na, float* b, size_t nb, hipStream_t s) {
  const size_t n = na + nb;
  const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(zero2_kernel, dim3(blocks), dim3(256), 0, s, a, na, b, nb);
}

}  // namespace gigs

extern "C" {
int gigs_internal_fail(int code, const char* fmt, ...);
void gigs_internal_stage_begin(int stage, void* stream, void** token);
void gigs_internal_stage_end(void* token);

int gigs_gbuffer_post(int height, int width, const float* normal_map, const float* out_normal_view,
                      const float* viewmatrix, float* normals_view, uint8_t* normal_mask, float* normal_mask_f,
                      float* out_normal_view_filtered, void* stream) {
  if (height <= 0 || width <= 0 || !normal_map || !out_normal_view || !viewmatrix || !normals_view ||
      !out_normal_view_filtered)
    return gigs_internal_fail(GIGS_ERR_INVALID, "gbuffer_post: bad argument");
  void* tok; gigs_internal_stage_begin(18, stream, &tok);
  hipLaunchKernelGGL(gigs::gbuffer_post_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0,
                     (hipStream_t)stream, height, width, normal_map, out_normal_view, viewmatrix, normals_view,
                     normal_mask, normal_mask_f, out_normal_view_filtered);
  gigs_internal_stage_end(tok);
  if (hipGetLastError() != hipSuccess) return gigs_internal_fail(GIGS_ERR_HIP, "gbuffer_post: launch failed");
  return 0;
}

int gigs_gbuffer_post_bwd(int height, int width, const float* normal_map, const float* viewmatrix,
                          const float* g_normals_view, float* scratch3, float* g_normal_map, void* stream) {
  if (height <= 0 || width <= 0 || !normal_map || !viewmatrix || !g_normals_view || !scratch3 || !g_normal_map)
    return gigs_internal_fail(GIGS_ERR_INVALID, "gbuffer_post_bwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const size_t HW = (size_t)height * width;
  void* tok; gigs_internal_stage_begin(18, stream, &tok);
  gigs::launch_zero(scratch3, 3 * HW, s);
  hipLaunchKernelGGL(gigs::gbuffer_post_bwd_scatter_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, s,
                     height, width, normal_map, viewmatrix, g_normals_view, scratch3);
  hipLaunchKernelGGL(gigs::gbuffer_post_bwd_finish_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, s, HW,
                     normal_map, scratch3, g_normal_map);
  gigs_internal_stage_end(tok);
  if (hipGetLastError() != hipSuccess) return gigs_internal_fail(GIGS_ERR_HIP, "gbuffer_post_bwd: launch failed");
  return 0;
}

int gigs_normalize_mask(int height, int width, const float* in, float* out, uint8_t* mask, void* stream) {
  if (height <= 0 || width <= 0 || !in || !out) return gigs_internal_fail(GIGS_ERR_INVALID, "normalize_mask: bad argument");
  const size_t HW = (size_t)height * width;
  void* tok; gigs_internal_stage_begin(18, stream, &tok);
  hipLaunchKernelGGL(gigs::normalize_mask_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, (hipStream_t)stream, HW,
                     in, out, mask);
  gigs_internal_stage_end(tok);
  if (hipGetLastError() != hipSuccess) return gigs_internal_fail(GIGS_ERR_HIP, "normalize_mask: launch failed");
  return 0;
}

int gigs_nonzero_mask(int height, int width, const float* in, float* mask, void* stream) {
  if (height <= 0 || width <= 0 || !in || !mask) return gigs_internal_fail(GIGS_ERR_INVALID, "nonzero_mask: bad argument");
  const size_t HW = (size_t)height * width;
  hipLaunchKernelGGL(gigs::nonzero_mask_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, (hipStream_t)stream, HW, in, mask);
  if (hipGetLastError() != hipSuccess) return gigs_internal_fail(GIGS_ERR_HIP, "nonzero_mask: launch failed");
  return 0;
}

int gigs_stage2_loss_fwd(int height, int width, const float* render_direct, const float* irr_linear,
                         const float* gt_image, const float* normal_mask_f, const float* roughness,
                         const float* metallic, float* render_rgb, float* acc4, float* loss, void* stream) {
  if (height <= 0 || width <= 0 || !render_direct || !irr_linear || !gt_image || !normal_mask_f || !roughness ||
      !metallic || !acc4 || !loss)
    return gigs_internal_fail(GIGS_ERR_INVALID, "stage2_loss_fwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  void* tok; gigs_internal_stage_begin(19, stream, &tok);
  float* rows = acc4 + 4;  // [kAccSlots][4] partial sums behind the four totals
  gigs::launch_zero(rows, 4 * gigs::kAccSlots, s);
  hipLaunchKernelGGL(gigs::stage2_loss_fwd_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, s, height,
                     width, render_direct, irr_linear, gt_image, normal_mask_f, roughness, metallic, render_rgb, rows,
                     (float*)nullptr, (float*)nullptr);
  hipLaunchKernelGGL(gigs::stage2_loss_finish_kernel, dim3(1), dim3(64), 0, s, height, width, rows, acc4, loss);
  gigs_internal_stage_end(tok);
  if (hipGetLastError() != hipSuccess) return gigs_internal_fail(GIGS_ERR_HIP, "stage2_loss_fwd: launch failed");
  return 0;
}

int gigs_stage2_loss_fwd_grad(int height, int width, const float* render_direct, const float* irr_linear,
                              const float* gt_image, const float* normal_mask_f, const float* roughness,
                              const float* metallic, float* render_rgb, float* acc4, float* loss,
                              float* d_render_direct_unit, float* d_irr_linear_unit, void* stream) {
  if (height <= 0 || width <= 0 || !render_direct || !irr_linear || !gt_image || !normal_mask_f || !roughness ||
      !metallic || !acc4 || !loss || !d_render_direct_unit || !d_irr_linear_unit)
    return gigs_internal_fail(GIGS_ERR_INVALID, "stage2_loss_fwd_grad: bad argument");
  hipStream_t s = (hipStream_t)stream;
  void* tok; gigs_internal_stage_begin(19, stream, &tok);
  float* rows = acc4 + 4;
  gigs::launch_zero2(rows, 4 * gigs::kAccSlots, d_irr_linear_unit, 3 * (size_t)height * width, s);
  hipLaunchKernelGGL(gigs::stage2_loss_fwd_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, s, height,
                     width, render_direct, irr_linear, gt_image, normal_mask_f, roughness, metallic, render_rgb, rows,
                     d_render_direct_unit, d_irr_linear_unit);
  hipLaunchKernelGGL(gigs::stage2_loss_finish_kernel, dim3(1), dim3(64), 0, s, height, width, rows, acc4, loss);
  gigs_internal_stage_end(tok);
  if (hipGetLastError() != hipSuccess) return gigs_internal_fail(GIGS_ERR_HIP, "stage2_loss_fwd_grad: launch failed");
  return 0;
}

int gigs_stage2_loss_bwd(int height, int width, const float* render_direct, const float* irr_linear,
                         const float* gt_image, const float* normal_mask_f, const float* acc4, const float* g_loss,
                         float* d_render_direct, float* d_irr_linear, float* d_roughness, float* d_metallic,
                         void* stream) {
  if (height <= 0 || width <= 0 || !render_direct || !irr_linear || !gt_image || !normal_mask_f || !acc4 ||
      !d_render_direct || !d_irr_linear || !d_roughness || !d_metallic)
    return gigs_internal_fail(GIGS_ERR_INVALID, "stage2_loss_bwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  void* tok; gigs_internal_stage_begin(20, stream, &tok);
  gigs::launch_zero(d_irr_linear, 3 * (size_t)height * width, s);
  hipLaunchKernelGGL(gigs::stage2_loss_bwd_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, s, height,
                     width, render_direct, irr_linear, gt_image, normal_mask_f, acc4, g_loss, d_render_direct,
                     d_irr_linear, d_roughness, d_metallic);
  gigs_internal_stage_end(tok);
  if (hipGetLastError() != hipSuccess) return gigs_internal_fail(GIGS_ERR_HIP, "stage2_loss_bwd: launch failed");
  return 0;
}

}  // extern "C"
