// preprocess.hip -- per-Gaussian kernels: projection / EWA covariance / SH colour (forward),
// the matching chain rule (backward), and the frustum mask.
//
// Reference behaviour restated (R/ = submodules/diff-gaussian-rasterization):
//   forward : preprocessCUDA           R/cuda_rasterizer/forward.cu:164-276
//             computeColorFromSH       forward.cu:22-80
//             computeCov2D / Cov3D     forward.cu:83-161
//   backward: computeCov2DCUDA         R/cuda_rasterizer/backward.cu:145-279
//             preprocessCUDA           backward.cu:351-401
//             computeColorFromSH (bwd) backward.cu:21-140, computeCov3D (bwd) :283-346
//   mask    : checkFrustum             R/cuda_rasterizer/rasterizer_impl.cu:54-66
//
// MI355X design: one lane per Gaussian, 256-lane workgroups.  The kernels are HBM-streaming
// (52 B + 12*M B in, ~170 B out per Gaussian).  The SH block of a workgroup
// (256 * M * 12 B, contiguous) is staged through LDS with 16-byte coalesced loads and read
// back with an odd per-lane stride (3*M words) so that the 64 lanes hit distinct banks;
// the forward also writes the 80-byte packed blend record (gigs_common.h) that the blend
// kernels gather.  No MFMA: 16x3 FMAs per Gaussian do not vectorise into a matrix tile.
#include "gigs_common.h"

namespace gigs {

#define SHC0 0.28209479177387814f
#define SHC1 0.4886025119029199f
#define SHC2_0 1.0925484305920792f
#define SHC2_1 -1.0925484305920792f
#define SHC2_2 0.31539156525252005f
#define SHC2_3 -1.0925484305920792f
#define SHC2_4 0.5462742152960396f
#define SHC3_0 -0.5900435899266435f
#define SHC3_1 2.890611442640554f
#define SHC3_2 -0.4570457994644658f
#define SHC3_3 0.3731763325901154f
#define SHC3_4 -0.4570457994644658f
#define SHC3_5 1.445305721320277f
#define SHC3_6 -0.5900435899266435f

constexpr int kPreBlock = 256;

// Stage the workgroup's contiguous SH block [nG][M][3] into LDS: coalesced 16-byte global
// loads, then scattered into a per-lane row of `sh_stride(M)` words.  The stride is odd so the
// 64 lanes of a wave read their k-th coefficient from distinct LDS banks (3*M itself is
// 48 words at M = 16: a 16-way conflict).  The block base is 16-byte aligned because
// 256 * M * 12 is.
__host__ __device__ __forceinline__ int sh_stride(int M) { return (3 * M) | 1; }

__device__ __forceinline__ void stage_sh(const float* __restrict__ shs, float* sh_lds,
                                         size_t first, int nG, int M) {
  const int n3 = 3 * M, stride = sh_stride(M);
  const int nfloat = nG * n3;
  const float* src = shs + first * (size_t)n3;
  const int nvec = nfloat >> 2;
  const float4* src4 = reinterpret_cast<const float4*>(src);
  for (int i = threadIdx.x; i < nvec; i += kPreBlock) {
    const float4 v = src4[i];
    const int e = i << 2;
    int lane = e / n3, j = e - lane * n3;
    const float c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      sh_lds[lane * stride + j] = c[k];
      if (++j == n3) { j = 0; ++lane; }
    }
  }
  for (int e = (nvec << 2) + threadIdx.x; e < nfloat; e += kPreBlock) {
    const int lane = e / n3;
    sh_lds[lane * stride + (e - lane * n3)] = src[e];
  }
}

// The same LDS rows from the optimizer's two tensors (gigs_ctx_set_split_sh): coefficient 0 from `dc` [nG][3], coefficients
// 1..M-1 from `rest` [nG][3(M-1)] -- no concatenated copy of the SH block exists anywhere.
__device__ __forceinline__ void stage_sh_split(const float* __restrict__ dc, const float* __restrict__ rest, float* sh_lds,
                                               size_t first, int nG, int M) {
  const int n3 = 3 * M, r = n3 - 3, stride = sh_stride(M);
  const float* d = dc + first * 3;
  for (int e = threadIdx.x; e < nG * 3; e += kPreBlock) {
    const int lane = e / 3;
    sh_lds[lane * stride + (e - lane * 3)] = d[e];
  }
  const float* src = rest + first * (size_t)r;
  const int nfloat = nG * r;
  const int nvec = (reinterpret_cast<uintptr_t>(src) & 15) == 0 ? nfloat >> 2 : 0;
  const float4* src4 = reinterpret_cast<const float4*>(src);
  for (int i = threadIdx.x; i < nvec; i += kPreBlock) {
    const float4 v = src4[i];
    const int e = i << 2;
    int lane = e / r, j = e - lane * r;
    const float c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      sh_lds[lane * stride + 3 + j] = c[k];
      if (++j == r) { j = 0; ++lane; }
    }
  }
  for (int e = (nvec << 2) + threadIdx.x; e < nfloat; e += kPreBlock) {
    const int lane = e / r;
    sh_lds[lane * stride + 3 + (e - lane * r)] = src[e];
  }
}

__device__ __forceinline__ v3 ld3(const float* p) { return {p[0], p[1], p[2]}; }

struct Cov2D {
  v3 t;
  float txtz, tytz, limx, limy;
  m3 Wm, T, Vrk, cov;
};

// forward.cu:83-115 and its recomputation in backward.cu:166-196
__device__ __forceinline__ Cov2D cov2d(v3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                                       const float* cov3D, const float* vm) {
  Cov2D s;
  v3 t = xform_point_4x3(mean, vm);
  s.limx = 1.3f * tan_fovx;
  s.limy = 1.3f * tan_fovy;
  s.txtz = t.x / t.z;
  s.tytz = t.y / t.z;
  t.x = fminf(s.limx, fmaxf(-s.limx, s.txtz)) * t.z;
  t.y = fminf(s.limy, fmaxf(-s.limy, s.tytz)) * t.z;
  s.t = t;
  m3 J = make_m3(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z), 0.0f, fy / t.z,
                 -(fy * t.y) / (t.z * t.z), 0.0f, 0.0f, 0.0f);
  s.Wm = make_m3(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
  s.T = mul3(s.Wm, J);
  s.Vrk = make_m3(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4],
                  cov3D[5]);
  s.cov = mul3(mul3(transpose3(s.T), transpose3(s.Vrk)), s.T);
  return s;
}

__device__ __forceinline__ m3 quat_to_R(float r, float x, float y, float z) {
  return make_m3(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                 2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                 2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

__global__ void __launch_bounds__(kPreBlock)
preprocess_fwd_kernel(FwdArgs a, GeomState g, int* __restrict__ radii) {
  extern __shared__ __align__(16) float sh_lds[];
  const int idx = blockIdx.x * kPreBlock + threadIdx.x;
  const int P = a.P;
  const bool use_sh = (a.colors_precomp == nullptr);
  if (use_sh) {
    const size_t first = (size_t)blockIdx.x * kPreBlock;
    const int nG = min(kPreBlock, P - (int)first);
    if (a.shs_rest) stage_sh_split(a.shs, a.shs_rest, sh_lds, first, nG, a.M);
    else stage_sh(a.shs, sh_lds, first, nG, a.M);
    __syncthreads();
  }
  if (idx >= P) return;

  radii[idx] = 0;
  g.tiles_touched[idx] = 0;
  {  // the blend backward accumulates into grec with atomics: it leaves every forward (and every backward) zeroed,
     // so the backward needs no fill launch in front of it
    float4* gz = reinterpret_cast<float4*>(g.grec + (size_t)idx * GIGS_GREC);
    const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    gz[0] = z4; gz[1] = z4; gz[2] = z4; gz[3] = z4; gz[4] = z4;
  }

  // in_frustum (auxiliary.h:150-176): only the near cull survives in the reference
  const v3 p_orig = ld3(a.means3D + 3 * (size_t)idx);
  const v3 p_view = xform_point_4x3(p_orig, a.viewmatrix);
  if (p_view.z <= 0.2f) return;

  const v4 p_hom = xform_point_4x4(p_orig, a.projmatrix);
  const float p_w = 1.0f / (p_hom.w + 0.0000001f);
  const float p_proj_x = p_hom.x * p_w, p_proj_y = p_hom.y * p_w;

  float c3[6];
  if (a.cov3D_precomp != nullptr) {
#pragma unroll
    for (int i = 0; i < 6; i++) c3[i] = a.cov3D_precomp[6 * (size_t)idx + i];
  } else {
    // computeCov3D (forward.cu:127-161); the quaternion is used as given (:136)
    const v3 sc = ld3(a.scales + 3 * (size_t)idx);
    const float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];
    m3 S = make_m3(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = a.scale_modifier * sc.x;
    S.m[1][1] = a.scale_modifier * sc.y;
    S.m[2][2] = a.scale_modifier * sc.z;
    const m3 Rm = quat_to_R(q.x, q.y, q.z, q.w);
    const m3 Mm = mul3(S, Rm);
    const m3 Sigma = mul3(transpose3(Mm), Mm);
    c3[0] = Sigma.m[0][0];
    c3[1] = Sigma.m[0][1];
    c3[2] = Sigma.m[0][2];
    c3[3] = Sigma.m[1][1];
    c3[4] = Sigma.m[1][2];
    c3[5] = Sigma.m[2][2];
#pragma unroll
    for (int i = 0; i < 6; i++) g.cov3D[6 * (size_t)idx + i] = c3[i];
  }

  const Cov2D s = cov2d(p_orig, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, c3, a.viewmatrix);
  const float cx = s.cov.m[0][0] + 0.3f, cy = s.cov.m[0][1], cz = s.cov.m[1][1] + 0.3f;
  const float det = (cx * cz - cy * cy);
  if (det == 0.0f) return;
  const float det_inv = 1.f / det;
  const float conx = cz * det_inv, cony = -cy * det_inv, conz = cx * det_inv;

  const float mid = 0.5f * (cx + cz);
  const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
  const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
  const float pix = ndc2pix(p_proj_x, a.W), piy = ndc2pix(p_proj_y, a.H);
  unsigned minx, miny, maxx, maxy;
  tile_rect(pix, piy, f2i(my_radius), a.gx, a.gy, minx, miny, maxx, maxy);
  if ((maxx - minx) * (maxy - miny) == 0) return;

  v3 col;
  if (use_sh) {
    // computeColorFromSH (forward.cu:22-80); sh[k] read from the staged LDS block
    const float* sh = sh_lds + threadIdx.x * sh_stride(a.M);
#define SH(k) (v3{sh[3 * (k)], sh[3 * (k) + 1], sh[3 * (k) + 2]})
    const float* cp = a.cam_pos;
    v3 dir = {p_orig.x - cp[0], p_orig.y - cp[1], p_orig.z - cp[2]};
    const float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
    dir = {dir.x / len, dir.y / len, dir.z / len};
    v3 result = SH(0) * SHC0;
    if (a.D > 0) {
      const float x = dir.x, y = dir.y, z = dir.z;
      result = result - SH(1) * (SHC1 * y) + SH(2) * (SHC1 * z) - SH(3) * (SHC1 * x);
      if (a.D > 1) {
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xy = x * y, yz = y * z, xz = x * z;
        result = result + SH(4) * (SHC2_0 * xy) + SH(5) * (SHC2_1 * yz) +
                 SH(6) * (SHC2_2 * (2.0f * zz - xx - yy)) + SH(7) * (SHC2_3 * xz) +
                 SH(8) * (SHC2_4 * (xx - yy));
        if (a.D > 2) {
          result = result + SH(9) * (SHC3_0 * y * (3.0f * xx - yy)) + SH(10) * (SHC3_1 * xy * z) +
                   SH(11) * (SHC3_2 * y * (4.0f * zz - xx - yy)) +
                   SH(12) * (SHC3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) +
                   SH(13) * (SHC3_4 * x * (4.0f * zz - xx - yy)) + SH(14) * (SHC3_5 * z * (xx - yy)) +
                   SH(15) * (SHC3_6 * x * (xx - 3.0f * yy));
        }
      }
    }
#undef SH
    result = {result.x + 0.5f, result.y + 0.5f, result.z + 0.5f};
    g.clamped[3 * (size_t)idx + 0] = (result.x < 0);
    g.clamped[3 * (size_t)idx + 1] = (result.y < 0);
    g.clamped[3 * (size_t)idx + 2] = (result.z < 0);
    col = {result.x < 0.0f ? 0.0f : result.x, result.y < 0.0f ? 0.0f : result.y,
           result.z < 0.0f ? 0.0f : result.z};
    g.rgb[3 * (size_t)idx + 0] = col.x;
    g.rgb[3 * (size_t)idx + 1] = col.y;
    g.rgb[3 * (size_t)idx + 2] = col.z;
  } else {
    col = ld3(a.colors_precomp + 3 * (size_t)idx);
  }

  const float opac = a.opacities[idx];
  g.depths[idx] = p_view.z;
  radii[idx] = f2i(my_radius);
  reinterpret_cast<float2*>(g.means2D)[idx] = make_float2(pix, piy);
  reinterpret_cast<float4*>(g.conic_opacity)[idx] = make_float4(conx, cony, conz, opac);
  g.pos_view[3 * (size_t)idx + 0] = p_view.x;
  g.pos_view[3 * (size_t)idx + 1] = p_view.y;
  g.pos_view[3 * (size_t)idx + 2] = p_view.z;
  g.tiles_touched[idx] = (maxy - miny) * (maxx - minx);

  // packed blend record (layout: gigs_common.h)
  const v3 nrm = ld3(a.normal + 3 * (size_t)idx);
  const v3 alb = ld3(a.albedo + 3 * (size_t)idx);
  float4* rec = g.brec + (size_t)idx * GIGS_BREC_F4;
  rec[0] = make_float4(pix, piy, a.roughness[idx], a.metallic[idx]);
  rec[1] = make_float4(conx, cony, conz, opac);
  rec[2] = make_float4(col.x, col.y, col.z, p_view.x);
  rec[3] = make_float4(nrm.x, nrm.y, nrm.z, p_view.y);
  rec[4] = make_float4(alb.x, alb.y, alb.z, p_view.z);
}

void launch_preprocess_fwd(const FwdArgs& a, const GeomState& g, int* radii, hipStream_t s) {
  const int blocks = (a.P + kPreBlock - 1) / kPreBlock;
  const size_t lds = a.colors_precomp ? 0 : (size_t)kPreBlock * sh_stride(a.M) * sizeof(float);
  hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(blocks), dim3(kPreBlock), lds, s, a, g, radii);
}

// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ vm,
                    uint8_t* __restrict__ present) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const v3 p = ld3(means3D + 3 * (size_t)idx);
  const v3 pv = xform_point_4x3(p, vm);
  present[idx] = !(pv.z <= 0.2f);
}

void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present,
                         hipStream_t s) {
  hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D,
                     viewmatrix, present);
}

// ------------------------------------------------------------------------------------------
// Backward: one kernel does computeCov2DCUDA + preprocessCUDA(bwd) (the reference launches
// them back to back over the same index space, backward.cu:839-857) and first unpacks the
// packed gradient record written by the blend backward into the caller's separate tensors.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ v3 dnormvdv3(v3 v, v3 dv) {  // auxiliary.h:118-128
  const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  v3 r;
  r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
  r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
  r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
  return r;
}

__global__ void __launch_bounds__(kPreBlock)
preprocess_bwd_kernel(BwdArgs a, GeomState g, int sh_always, unsigned* __restrict__ materials_only) {
  extern __shared__ __align__(16) float sh_lds[];
  const int idx = blockIdx.x * kPreBlock + threadIdx.x;
  const int P = a.P, M = a.M, D = a.D;
  const bool live = idx < P;

  // ---- unpack the blend-backward record into the reference's separate gradient tensors
  const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float4 g0 = z4, g1 = z4, g2 = z4, g3 = z4, g4 = z4;
  if (live) {
    float4* gr = reinterpret_cast<float4*>(g.grec + (size_t)idx * GIGS_GREC);
    g0 = gr[0]; g1 = gr[1]; g2 = gr[2]; g3 = gr[3]; g4 = gr[4];
    // consumed: zero again, so that another backward on the same forward state starts from zero (see preprocess_fwd)
    gr[0] = z4; gr[1] = z4; gr[2] = z4; gr[3] = z4; gr[4] = z4;
  }

  if (materials_only) {
    // Declared stage-2 gradient set (gigs_ctx_set_materials_only): the loss reaches the material planes only, whose blend
    // gradients do not feed dL/dalpha (backward.cu:580-590) -- every other slot of the record is an exact zero, and with
    // it every other output of this kernel.  Those outputs are not written (the caller keeps no tensor for them); the
    // premise is CHECKED: a live Gaussian with any other non-zero (or NaN) slot counts as a violation.
    bool bad = false;
    if (live) {
      a.dL_dmean2D[3 * (size_t)idx + 0] = g0.x;
      a.dL_dmean2D[3 * (size_t)idx + 1] = g0.y;
      a.dL_dmean2D[3 * (size_t)idx + 2] = g0.z;
      a.dL_dalbedo[3 * (size_t)idx + 0] = g3.y;
      a.dL_dalbedo[3 * (size_t)idx + 1] = g3.z;
      a.dL_dalbedo[3 * (size_t)idx + 2] = g3.w;
      a.dL_droughness[idx] = g4.x;
      a.dL_dmetallic[idx] = g4.y;
      bad = g0.x != 0.0f || g0.y != 0.0f || g0.w != 0.0f || g1.x != 0.0f || g1.y != 0.0f || g1.z != 0.0f || g1.w != 0.0f ||
            g2.x != 0.0f || g2.y != 0.0f || g2.z != 0.0f || g2.w != 0.0f || g3.x != 0.0f || g4.z != 0.0f;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicAdd(materials_only, 1u);
    return;
  }

  // The SH coefficients (12*M bytes per Gaussian) matter only where a colour gradient arrived: with
  // dL_dcolor == 0 the reference's computeColorFromSH backward (backward.cu:21-140) yields dL_dsh = 0
  // and adds 0 to dL_dmean for finite coefficients. A block whose Gaussians all have a zero colour
  // gradient (every block of a stage-2 step, whose loss never reads the SH colour plane) therefore
  // skips the read and writes its dL_dsh zeros as one contiguous run.
  int block_sh = 0;
  if (a.shs) {
    const size_t first = (size_t)blockIdx.x * kPreBlock;
    const int nG = min(kPreBlock, P - (int)first);
    block_sh = sh_always || __syncthreads_or(live && !(g1.w == 0.0f && g2.x == 0.0f && g2.y == 0.0f));
    if (block_sh) {
      stage_sh(a.shs, sh_lds, first, nG, M);
      __syncthreads();
    } else {
      float* z = a.dL_dsh + first * M * 3;
      const int n = nG * M * 3;
      const int n4 = (reinterpret_cast<uintptr_t>(z) & 15) == 0 ? n / 4 : 0;  // a slab view may start off a 16-byte boundary
      for (int i = threadIdx.x; i < n4; i += kPreBlock) reinterpret_cast<float4*>(z)[i] = z4;
      for (int i = 4 * n4 + threadIdx.x; i < n; i += kPreBlock) z[i] = 0.0f;
    }
  }
  if (!live) return;
  // g0 = (m2d.x, m2d.y, m2d.abs, con.xx) g1 = (con.xy, con.yy, dopac, dcol.r)
  // g2 = (dcol.g, dcol.b, dn.x, dn.y) g3 = (dn.z, dalb.r, dalb.g, dalb.b) g4 = (drough, dmetal, ddepth, -)
  a.dL_dmean2D[3 * (size_t)idx + 0] = g0.x;
  a.dL_dmean2D[3 * (size_t)idx + 1] = g0.y;
  a.dL_dmean2D[3 * (size_t)idx + 2] = g0.z;
  if (a.dL_dconic) reinterpret_cast<float4*>(a.dL_dconic)[idx] = make_float4(g0.w, g1.x, 0.0f, g1.y);
  a.dL_dopacity[idx] = g1.z;
  a.dL_dcolor[3 * (size_t)idx + 0] = g1.w;
  a.dL_dcolor[3 * (size_t)idx + 1] = g2.x;
  a.dL_dcolor[3 * (size_t)idx + 2] = g2.y;
  a.dL_dnormal[3 * (size_t)idx + 0] = g2.z;
  a.dL_dnormal[3 * (size_t)idx + 1] = g2.w;
  a.dL_dnormal[3 * (size_t)idx + 2] = g3.x;
  a.dL_dalbedo[3 * (size_t)idx + 0] = g3.y;
  a.dL_dalbedo[3 * (size_t)idx + 1] = g3.z;
  a.dL_dalbedo[3 * (size_t)idx + 2] = g3.w;
  a.dL_droughness[idx] = g4.x;
  a.dL_dmetallic[idx] = g4.y;
  if (a.dL_ddepth) a.dL_ddepth[idx] = g4.z;

  // Every term of the geometry gradients below carries one of these as a factor: with all of them zero (the detached
  // blend weights of a stage-2 step leave only the material planes' gradients) the chain yields zeros for finite
  // parameters, so the Gaussian's geometry (64 bytes) is not read and nothing is evaluated.
  const bool geo_zero = !sh_always && g0.x == 0.0f && g0.y == 0.0f && g0.w == 0.0f && g1.x == 0.0f && g1.y == 0.0f &&
                        g4.z == 0.0f && g1.w == 0.0f && g2.x == 0.0f && g2.y == 0.0f;
  if (geo_zero || !(a.radii[idx] > 0)) {
    // culled Gaussian: the reference leaves the caller's zero-initialised outputs untouched; writing
    // the zeros here lets the caller hand in uninitialised memory (no memset launches per tensor)
    a.dL_dmean3D[3 * (size_t)idx + 0] = 0.0f; a.dL_dmean3D[3 * (size_t)idx + 1] = 0.0f; a.dL_dmean3D[3 * (size_t)idx + 2] = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * (size_t)idx + i] = 0.0f;
    if (block_sh)
      for (int i = 0; i < 3 * M; i++) a.dL_dsh[(size_t)idx * M * 3 + i] = 0.0f;
    a.dL_dscale[3 * (size_t)idx + 0] = 0.0f; a.dL_dscale[3 * (size_t)idx + 1] = 0.0f; a.dL_dscale[3 * (size_t)idx + 2] = 0.0f;
    reinterpret_cast<float4*>(a.dL_drot)[idx] = make_float4(0, 0, 0, 0);
    return;
  }

  // ---- computeCov2DCUDA (backward.cu:145-279)
  const float* vm = a.viewmatrix;
  float c3[6];
  const float* c3src = (a.cov3D_precomp ? a.cov3D_precomp : g.cov3D) + 6 * (size_t)idx;
#pragma unroll
  for (int i = 0; i < 6; i++) c3[i] = c3src[i];
  const v3 mean = ld3(a.means3D + 3 * (size_t)idx);
  const v3 dL_dconic = {g0.w, g1.x, g1.y};
  const Cov2D s = cov2d(mean, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, c3, vm);
  const v3 t = s.t;
  const float h_x = a.focal_x, h_y = a.focal_y;
  const float x_grad_mul = (s.txtz < -s.limx || s.txtz > s.limx) ? 0 : 1;
  const float y_grad_mul = (s.tytz < -s.limy || s.tytz > s.limy) ? 0 : 1;
  const m3& T = s.T;
  const m3& Vrk = s.Vrk;
  const m3& Wm = s.Wm;
  const float ca = s.cov.m[0][0] + 0.3f;
  const float cb = s.cov.m[0][1];
  const float cc = s.cov.m[1][1] + 0.3f;
  const float denom = ca * cc - cb * cb;
  float dL_da = 0, dL_db = 0, dL_dc = 0;
  const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
  float dcov[6];
  if (denom2inv != 0) {
    dL_da = denom2inv * (-cc * cc * dL_dconic.x + 2 * cb * cc * dL_dconic.y + (denom - ca * cc) * dL_dconic.z);
    dL_dc = denom2inv * (-ca * ca * dL_dconic.z + 2 * ca * cb * dL_dconic.y + (denom - ca * cc) * dL_dconic.x);
    dL_db = denom2inv * 2 * (cb * cc * dL_dconic.x - (denom + 2 * cb * cb) * dL_dconic.y + ca * cb * dL_dconic.z);
    dcov[0] = (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
    dcov[3] = (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
    dcov[5] = (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);
    dcov[1] = 2 * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][1] * dL_dc;
    dcov[2] = 2 * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][2] * dL_dc;
    dcov[4] = 2 * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db + 2 * T.m[1][1] * T.m[1][2] * dL_dc;
  } else {
#pragma unroll
    for (int i = 0; i < 6; i++) dcov[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];

  const float dL_dT00 = 2 * (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_da +
                        (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_db;
  const float dL_dT01 = 2 * (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_da +
                        (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_db;
  const float dL_dT02 = 2 * (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_da +
                        (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_db;
  const float dL_dT10 = 2 * (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc +
                        (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_db;
  const float dL_dT11 = 2 * (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc +
                        (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_db;
  const float dL_dT12 = 2 * (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc +
                        (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_db;
  const float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
  const float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
  const float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
  const float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
  const float tz = 1.f / t.z;
  const float tz2 = tz * tz;
  const float tz3 = tz2 * tz;
  const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
  const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
  const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
  v3 dmeans = xform_vec_4x3_T({dL_dtx, dL_dty, dL_dtz}, vm);
  dmeans.x += vm[2] * g4.z;
  dmeans.y += vm[6] * g4.z;
  dmeans.z += vm[10] * g4.z;

  // ---- preprocessCUDA backward (backward.cu:375-392)
  const float* proj = a.projmatrix;
  const v3 m = mean;
  const v4 m_hom = xform_point_4x4(m, proj);
  const float m_w = 1.0f / (m_hom.w + 0.0000001f);
  const float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
  const float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
  v3 dm2;
  dm2.x = (proj[0] * m_w - proj[3] * mul1) * g0.x + (proj[1] * m_w - proj[3] * mul2) * g0.y;
  dm2.y = (proj[4] * m_w - proj[7] * mul1) * g0.x + (proj[5] * m_w - proj[7] * mul2) * g0.y;
  dm2.z = (proj[8] * m_w - proj[11] * mul1) * g0.x + (proj[9] * m_w - proj[11] * mul2) * g0.y;
  dmeans = dmeans + dm2;

  // ---- computeColorFromSH backward (backward.cu:21-140)
  if (block_sh) {
    const float* sh = sh_lds + threadIdx.x * sh_stride(M);
#define SH(k) (v3{sh[3 * (k)], sh[3 * (k) + 1], sh[3 * (k) + 2]})
    const float* cp = a.cam_pos;
    const v3 dir_orig = {m.x - cp[0], m.y - cp[1], m.z - cp[2]};
    const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    const v3 dir = {dir_orig.x / len, dir_orig.y / len, dir_orig.z / len};
    v3 dRGB = {g1.w, g2.x, g2.y};
    dRGB.x *= g.clamped[3 * (size_t)idx + 0] ? 0 : 1;
    dRGB.y *= g.clamped[3 * (size_t)idx + 1] ? 0 : 1;
    dRGB.z *= g.clamped[3 * (size_t)idx + 2] ? 0 : 1;
    v3 dRGBdx = {0, 0, 0}, dRGBdy = {0, 0, 0}, dRGBdz = {0, 0, 0};
    const float x = dir.x, y = dir.y, z = dir.z;
    float* dsh = a.dL_dsh + (size_t)idx * M * 3;
#define W3(k, v) { const v3 _v = (v); dsh[3 * (k)] = _v.x; dsh[3 * (k) + 1] = _v.y; dsh[3 * (k) + 2] = _v.z; }
    for (int i = 3 * (D + 1) * (D + 1); i < 3 * M; i++) dsh[i] = 0.0f;  // coefficients above the active degree
    W3(0, dRGB * SHC0);
    if (D > 0) {
      const float d1 = -SHC1 * y, d2 = SHC1 * z, d3 = -SHC1 * x;
      W3(1, dRGB * d1);
      W3(2, dRGB * d2);
      W3(3, dRGB * d3);
      dRGBdx = SH(3) * (-SHC1);
      dRGBdy = SH(1) * (-SHC1);
      dRGBdz = SH(2) * SHC1;
      if (D > 1) {
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xy = x * y, yz = y * z, xz = x * z;
        W3(4, dRGB * (SHC2_0 * xy));
        W3(5, dRGB * (SHC2_1 * yz));
        W3(6, dRGB * (SHC2_2 * (2.f * zz - xx - yy)));
        W3(7, dRGB * (SHC2_3 * xz));
        W3(8, dRGB * (SHC2_4 * (xx - yy)));
        dRGBdx = dRGBdx + (SH(4) * (SHC2_0 * y) + SH(6) * (SHC2_2 * 2.f * -x) + SH(7) * (SHC2_3 * z) + SH(8) * (SHC2_4 * 2.f * x));
        dRGBdy = dRGBdy + (SH(4) * (SHC2_0 * x) + SH(5) * (SHC2_1 * z) + SH(6) * (SHC2_2 * 2.f * -y) + SH(8) * (SHC2_4 * 2.f * -y));
        dRGBdz = dRGBdz + (SH(5) * (SHC2_1 * y) + SH(6) * (SHC2_2 * 2.f * 2.f * z) + SH(7) * (SHC2_3 * x));
        if (D > 2) {
          W3(9, dRGB * (SHC3_0 * y * (3.f * xx - yy)));
          W3(10, dRGB * (SHC3_1 * xy * z));
          W3(11, dRGB * (SHC3_2 * y * (4.f * zz - xx - yy)));
          W3(12, dRGB * (SHC3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy)));
          W3(13, dRGB * (SHC3_4 * x * (4.f * zz - xx - yy)));
          W3(14, dRGB * (SHC3_5 * z * (xx - yy)));
          W3(15, dRGB * (SHC3_6 * x * (xx - 3.f * yy)));
          dRGBdx = dRGBdx + (SH(9) * SHC3_0 * 3.f * 2.f * xy + SH(10) * SHC3_1 * yz + SH(11) * SHC3_2 * -2.f * xy +
                             SH(12) * SHC3_3 * -3.f * 2.f * xz + SH(13) * SHC3_4 * (-3.f * xx + 4.f * zz - yy) +
                             SH(14) * SHC3_5 * 2.f * xz + SH(15) * SHC3_6 * 3.f * (xx - yy));
          dRGBdy = dRGBdy + (SH(9) * SHC3_0 * 3.f * (xx - yy) + SH(10) * SHC3_1 * xz + SH(11) * SHC3_2 * (-3.f * yy + 4.f * zz - xx) +
                             SH(12) * SHC3_3 * -3.f * 2.f * yz + SH(13) * SHC3_4 * -2.f * xy + SH(14) * SHC3_5 * -2.f * yz +
                             SH(15) * SHC3_6 * -3.f * 2.f * xy);
          dRGBdz = dRGBdz + (SH(10) * SHC3_1 * xy + SH(11) * SHC3_2 * 4.f * 2.f * yz + SH(12) * SHC3_3 * 3.f * (2.f * zz - xx - yy) +
                             SH(13) * SHC3_4 * 4.f * 2.f * xz + SH(14) * SHC3_5 * (xx - yy));
        }
      }
    }
#undef W3
#undef SH
    const v3 dL_ddir = {dot3(dRGBdx, dRGB), dot3(dRGBdy, dRGB), dot3(dRGBdz, dRGB)};
    dmeans = dmeans + dnormvdv3(dir_orig, dL_ddir);
  }
  a.dL_dmean3D[3 * (size_t)idx + 0] = dmeans.x;
  a.dL_dmean3D[3 * (size_t)idx + 1] = dmeans.y;
  a.dL_dmean3D[3 * (size_t)idx + 2] = dmeans.z;

  // ---- computeCov3D backward (backward.cu:283-346)
  if (!a.scales) {
    a.dL_dscale[3 * (size_t)idx + 0] = 0.0f; a.dL_dscale[3 * (size_t)idx + 1] = 0.0f; a.dL_dscale[3 * (size_t)idx + 2] = 0.0f;
    reinterpret_cast<float4*>(a.dL_drot)[idx] = make_float4(0, 0, 0, 0);
  }
  if (a.scales) {
    const float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    const m3 Rm = quat_to_R(r, x, y, z);
    const v3 sc = ld3(a.scales + 3 * (size_t)idx);
    const v3 sv = {a.scale_modifier * sc.x, a.scale_modifier * sc.y, a.scale_modifier * sc.z};
    m3 S = make_m3(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = sv.x;
    S.m[1][1] = sv.y;
    S.m[2][2] = sv.z;
    const m3 Mm = mul3(S, Rm);
    const m3 dL_dSigma = make_m3(dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3],
                                 0.5f * dcov[4], 0.5f * dcov[2], 0.5f * dcov[4], dcov[5]);
    m3 M2;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) M2.m[i][j] = Mm.m[i][j] * 2.0f;
    const m3 dL_dM = mul3(M2, dL_dSigma);
    const m3 Rt = transpose3(Rm);
    m3 d = transpose3(dL_dM);
    a.dL_dscale[3 * (size_t)idx + 0] = Rt.m[0][0] * d.m[0][0] + Rt.m[0][1] * d.m[0][1] + Rt.m[0][2] * d.m[0][2];
    a.dL_dscale[3 * (size_t)idx + 1] = Rt.m[1][0] * d.m[1][0] + Rt.m[1][1] * d.m[1][1] + Rt.m[1][2] * d.m[1][2];
    a.dL_dscale[3 * (size_t)idx + 2] = Rt.m[2][0] * d.m[2][0] + Rt.m[2][1] * d.m[2][1] + Rt.m[2][2] * d.m[2][2];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      d.m[0][j] *= sv.x;
      d.m[1][j] *= sv.y;
      d.m[2][j] *= sv.z;
    }
    float4 dq;
    dq.x = 2 * z * (d.m[0][1] - d.m[1][0]) + 2 * y * (d.m[2][0] - d.m[0][2]) + 2 * x * (d.m[1][2] - d.m[2][1]);
    dq.y = 2 * y * (d.m[1][0] + d.m[0][1]) + 2 * z * (d.m[2][0] + d.m[0][2]) + 2 * r * (d.m[1][2] - d.m[2][1]) - 4 * x * (d.m[2][2] + d.m[1][1]);
    dq.z = 2 * x * (d.m[1][0] + d.m[0][1]) + 2 * r * (d.m[2][0] - d.m[0][2]) + 2 * z * (d.m[1][2] + d.m[2][1]) - 4 * y * (d.m[2][2] + d.m[0][0]);
    dq.w = 2 * r * (d.m[0][1] - d.m[1][0]) + 2 * x * (d.m[2][0] + d.m[0][2]) + 2 * y * (d.m[1][2] + d.m[2][1]) - 4 * z * (d.m[1][1] + d.m[0][0]);
    reinterpret_cast<float4*>(a.dL_drot)[idx] = dq;
  }
}

void launch_preprocess_bwd(const BwdArgs& a, const GeomState& g, int sh_skip, hipStream_t s, unsigned* materials_only) {
  const int blocks = (a.P + kPreBlock - 1) / kPreBlock;
  const size_t lds = a.shs ? (size_t)kPreBlock * sh_stride(a.M) * sizeof(float) : 0;
  // sh_skip = 0 (gigs_options.pre_bwd_sh_skip, diagnostic): evaluate every visible Gaussian and read the SH block of every
  // group, whatever the incoming gradients (as before round 3)
  const int sh_always = sh_skip ? 0 : 1;
  hipLaunchKernelGGL(preprocess_bwd_kernel, dim3(blocks), dim3(kPreBlock), materials_only ? 0 : lds, s, a, g, sh_always, materials_only);
}

}  // namespace gigs
