// gigs_oracle.cpp -- CPU restatement of the GI-GS rasterizer hot path.
//
// *** TEST INFRASTRUCTURE ONLY ***  Nothing in the product path (gi-gs_amd/) may
// link, import or call this file.  It is used by tests/, by
// __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the CHECKER.
//
// What it is: a scalar fp32 restatement, in the reference's own operation
// order, of the CUDA kernels of piotrmwojcik/GI-GS
// (submodules/diff-gaussian-rasterization, "R/" below).  Build with
//   g++ -O2 -ffp-contract=off -fopenmp -shared -fPIC   (see oracle/Makefile)
// so that no FMA contraction or fast-math reassociation happens.
//
// Pinning status: the reference CUDA cannot be built or run here (no nvcc, no
// NVIDIA device) and the reference holds no golden vectors for this path
// (SURVEY.md section 8c).  The restatement is pinned (i) piecewise against the
// reference's importable Python (eval_sh, projection matrices: see
// tests/golden/make_golden.py) and (ii) its hand-written backward against
// autograd of an independent PyTorch float64 restatement of the forward
// (oracle/torch_ref.py).  kornia filters are "parity unpinned".
//
// GPU-vs-x86 semantics that are emulated on purpose:
//   * float->int conversion saturates and maps NaN to 0 (f2i below);
//   * min/max on floats are fminf/fmaxf (CUDA overloads).
// Reference line citations are given per function as R/<file>:<lines>.

#include <algorithm>
#include <climits>
#include <cstddef>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int BLOCK_X = 16;  // R/cuda_rasterizer/config.h:15-17
constexpr int BLOCK_Y = 16;
constexpr float PI_F = 3.14159265358979323846f;  // M_PIf

// float -> int as the GPU does it (cvt.rzi.s32.f32 / v_cvt_i32_f32).
inline int f2i(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return INT_MAX;
  if (f <= -2147483648.0f) return INT_MIN;
  return (int)f;
}
inline int f2i(double f) {
  if (f != f) return 0;
  if (f >= 2147483648.0) return INT_MAX;
  if (f <= -2147483648.0) return INT_MIN;
  return (int)f;
}

struct f3 { float x, y, z; };
inline f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
// R/cuda_rasterizer/vec_math.h:476-479 (float3 / float multiplies by the reciprocal)
inline f3 div_s(f3 a, float s) { float inv = 1.0f / s; return a * inv; }
inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vec_math.h:518
inline f3 cross(f3 a, f3 b) {                                                 // vec_math.h:523
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline f3 normalize(f3 v) {  // vec_math.h:538-541 : v * (1/sqrtf(dot))
  float invLen = 1.0f / sqrtf(dot(v, v));
  return v * invLen;
}

// Column-major 3x3 with GLM's semantics: m[c][r]; the 9-scalar constructor fills
// columns; product order as in glm/detail/type_mat3x3.inl:486-519.
struct M3 {
  float m[3][3];
};
inline M3 mat3(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
  M3 r;
  r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
  r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
  r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
  return r;
}
inline M3 mul(const M3& A, const M3& B) {
  M3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++)
      R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
  return R;
}
inline M3 transpose(const M3& A) {
  M3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
  return R;
}

// R/cuda_rasterizer/auxiliary.h:58-108
inline f3 transformPoint4x3(f3 p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
          m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
struct f4 { float x, y, z, w; };
inline f4 transformPoint4x4(f3 p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
          m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
          m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}
inline f3 transformVec4x3(f3 p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z,
          m[1] * p.x + m[5] * p.y + m[9] * p.z,
          m[2] * p.x + m[6] * p.y + m[10] * p.z};
}
inline f3 transformVec3x3(f3 p, const float* m) {
  return {m[0] * p.x + m[3] * p.y + m[6] * p.z,
          m[1] * p.x + m[4] * p.y + m[7] * p.z,
          m[2] * p.x + m[5] * p.y + m[8] * p.z};
}
inline f3 transformVec4x3Transpose(f3 p, const float* m) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z,
          m[4] * p.x + m[5] * p.y + m[6] * p.z,
          m[8] * p.x + m[9] * p.y + m[10] * p.z};
}

// auxiliary.h:41-44 -- evaluated in double because of the 1.0 / 0.5 literals.
inline float ndc2Pix(float v, int S) { return (float)((((double)v + 1.0) * S - 1.0) * 0.5); }

// auxiliary.h:46-56
inline void getRect(float px, float py, int max_radius, unsigned gx, unsigned gy,
                    unsigned& minx, unsigned& miny, unsigned& maxx, unsigned& maxy) {
  minx = std::min(gx, (unsigned)std::max(0, f2i((px - max_radius) / BLOCK_X)));
  miny = std::min(gy, (unsigned)std::max(0, f2i((py - max_radius) / BLOCK_Y)));
  maxx = std::min(gx, (unsigned)std::max(0, f2i((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
  maxy = std::min(gy, (unsigned)std::max(0, f2i((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

const float SH_C0 = 0.28209479177387814f;  // auxiliary.h:22-39
const float SH_C1 = 0.4886025119029199f;
const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                       -1.0925484305920792f, 0.5462742152960396f};
const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                       0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                       -0.5900435899266435f};

struct Ctx {
  int P = 0, W = 0, H = 0, M = 0, D = 0, R = 0;
  unsigned gx = 0, gy = 0;
  std::vector<float> depths, pos_view, means2D, cov3D, conic_opacity, rgb;
  std::vector<uint8_t> clamped;
  std::vector<int> radii;
  std::vector<uint32_t> tiles_touched, point_offsets;
  std::vector<uint64_t> keys_unsorted, keys;
  std::vector<uint32_t> vals_unsorted, point_list;
  std::vector<uint32_t> ranges;  // uint2 per tile
  std::vector<float> final_T;
  std::vector<uint32_t> n_contrib;
  // scratch grads of the backward that the reference allocates but does not return
  std::vector<float> dL_dconic, dL_ddepth;
  uint64_t pairs_evaluated = 0, pairs_contributing = 0;
};

// ---------------------------------------------------------------------------------
// A1  preprocessCUDA forward                               R/cuda_rasterizer/forward.cu:164-276
// ---------------------------------------------------------------------------------
// computeColorFromSH: forward.cu:22-80
inline f3 sh_to_rgb(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                    const float* shs, uint8_t* clamped) {
  f3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
  f3 dir = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
  float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);  // glm::length
  dir = {dir.x / len, dir.y / len, dir.z / len};                     // glm vec / scalar
  const float* sh = shs + (size_t)idx * max_coeffs * 3;
  auto S = [&](int k) { return f3{sh[3 * k], sh[3 * k + 1], sh[3 * k + 2]}; };
  f3 result = S(0) * SH_C0;
  if (deg > 0) {
    float x = dir.x, y = dir.y, z = dir.z;
    result = result - S(1) * (SH_C1 * y) + S(2) * (SH_C1 * z) - S(3) * (SH_C1 * x);
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z;
      float xy = x * y, yz = y * z, xz = x * z;
      result = result + S(4) * (SH_C2[0] * xy) + S(5) * (SH_C2[1] * yz) +
               S(6) * (SH_C2[2] * (2.0f * zz - xx - yy)) + S(7) * (SH_C2[3] * xz) +
               S(8) * (SH_C2[4] * (xx - yy));
      if (deg > 2) {
        result = result + S(9) * (SH_C3[0] * y * (3.0f * xx - yy)) + S(10) * (SH_C3[1] * xy * z) +
                 S(11) * (SH_C3[2] * y * (4.0f * zz - xx - yy)) +
                 S(12) * (SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) +
                 S(13) * (SH_C3[4] * x * (4.0f * zz - xx - yy)) + S(14) * (SH_C3[5] * z * (xx - yy)) +
                 S(15) * (SH_C3[6] * x * (xx - 3.0f * yy));
      }
    }
  }
  result = {result.x + 0.5f, result.y + 0.5f, result.z + 0.5f};
  clamped[3 * idx + 0] = (result.x < 0);
  clamped[3 * idx + 1] = (result.y < 0);
  clamped[3 * idx + 2] = (result.z < 0);
  // glm::max(x, 0) == (x < 0) ? 0 : x
  return {result.x < 0.0f ? 0.0f : result.x, result.y < 0.0f ? 0.0f : result.y,
          result.z < 0.0f ? 0.0f : result.z};
}

// computeCov3D: forward.cu:127-161 (quaternion NOT normalised, :136)
inline void cov3d_from_scale_rot(const float* scale, float mod, const float* rot, float* cov3D) {
  M3 S = mat3(1, 0, 0, 0, 1, 0, 0, 0, 1);
  S.m[0][0] = mod * scale[0];
  S.m[1][1] = mod * scale[1];
  S.m[2][2] = mod * scale[2];
  float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  M3 Rm = mat3(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
               2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
               2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
  M3 Mm = mul(S, Rm);
  M3 Sigma = mul(transpose(Mm), Mm);
  cov3D[0] = Sigma.m[0][0];
  cov3D[1] = Sigma.m[0][1];
  cov3D[2] = Sigma.m[0][2];
  cov3D[3] = Sigma.m[1][1];
  cov3D[4] = Sigma.m[1][2];
  cov3D[5] = Sigma.m[2][2];
}

// shared by computeCov2D fwd (forward.cu:83-122) and bwd (backward.cu:166-196)
struct Cov2DState {
  f3 t;
  float txtz, tytz, limx, limy;
  M3 J, Wm, T, Vrk, cov;
};
inline Cov2DState cov2d_state(f3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                              const float* cov3D, const float* vm) {
  Cov2DState s;
  f3 t = transformPoint4x3(mean, vm);
  s.limx = 1.3f * tan_fovx;
  s.limy = 1.3f * tan_fovy;
  s.txtz = t.x / t.z;
  s.tytz = t.y / t.z;
  t.x = fminf(s.limx, fmaxf(-s.limx, s.txtz)) * t.z;
  t.y = fminf(s.limy, fmaxf(-s.limy, s.tytz)) * t.z;
  s.t = t;
  s.J = mat3(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z), 0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z), 0, 0, 0);
  s.Wm = mat3(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
  s.T = mul(s.Wm, s.J);
  s.Vrk = mat3(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
  s.cov = mul(mul(transpose(s.T), transpose(s.Vrk)), s.T);
  return s;
}

void preprocess_fwd(Ctx& c, int P, int D, int M, const float* means3D, const float* scales,
                    float scale_modifier, const float* rotations, const float* opacities,
                    const float* shs, const float* cov3D_precomp, const float* colors_precomp,
                    const float* viewmatrix, const float* projmatrix, const float* campos, int W,
                    int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y, int* radii) {
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    radii[idx] = 0;
    c.tiles_touched[idx] = 0;
    // in_frustum: auxiliary.h:150-176 -- only the near cull z_view <= 0.2
    f3 p_orig = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    f3 p_view = transformPoint4x3(p_orig, viewmatrix);
    if (p_view.z <= 0.2f) continue;
    f4 p_hom = transformPoint4x4(p_orig, projmatrix);
    float p_w = 1.0f / (p_hom.w + 0.0000001f);
    f3 p_proj = {p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w};
    const float* cov3D;
    if (cov3D_precomp != nullptr) {
      cov3D = cov3D_precomp + (size_t)idx * 6;
    } else {
      cov3d_from_scale_rot(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx,
                           c.cov3D.data() + (size_t)idx * 6);
      cov3D = c.cov3D.data() + (size_t)idx * 6;
    }
    Cov2DState s = cov2d_state(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix);
    // forward.cu:119-121 low-pass
    float cx = s.cov.m[0][0] + 0.3f, cy = s.cov.m[0][1], cz = s.cov.m[1][1] + 0.3f;
    float det = (cx * cz - cy * cy);
    if (det == 0.0f) continue;
    float det_inv = 1.f / det;
    float conx = cz * det_inv, cony = -cy * det_inv, conz = cx * det_inv;
    float mid = 0.5f * (cx + cz);
    float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    float pix = ndc2Pix(p_proj.x, W), piy = ndc2Pix(p_proj.y, H);
    unsigned minx, miny, maxx, maxy;
    getRect(pix, piy, f2i(my_radius), c.gx, c.gy, minx, miny, maxx, maxy);
    if ((maxx - minx) * (maxy - miny) == 0) continue;
    if (colors_precomp == nullptr) {
      f3 col = sh_to_rgb(idx, D, M, means3D, campos, shs, c.clamped.data());
      c.rgb[3 * idx + 0] = col.x;
      c.rgb[3 * idx + 1] = col.y;
      c.rgb[3 * idx + 2] = col.z;
    }
    c.depths[idx] = p_view.z;
    radii[idx] = f2i(my_radius);
    c.means2D[2 * idx] = pix;
    c.means2D[2 * idx + 1] = piy;
    c.conic_opacity[4 * idx + 0] = conx;
    c.conic_opacity[4 * idx + 1] = cony;
    c.conic_opacity[4 * idx + 2] = conz;
    c.conic_opacity[4 * idx + 3] = opacities[idx];
    c.pos_view[3 * idx + 0] = p_view.x;
    c.pos_view[3 * idx + 1] = p_view.y;
    c.pos_view[3 * idx + 2] = p_view.z;
    c.tiles_touched[idx] = (maxy - miny) * (maxx - minx);
  }
}

// getHigherMsb: R/cuda_rasterizer/rasterizer_impl.cu:35-50
uint32_t getHigherMsb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4;
  uint32_t step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb) msb += step;
    else msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

// ---------------------------------------------------------------------------------
// A2-A4  scan, duplicateWithKeys, stable sort, identifyTileRanges
//        R/cuda_rasterizer/rasterizer_impl.cu:585-629, 70-138
// ---------------------------------------------------------------------------------
void bin_and_sort(Ctx& c, const int* radii) {
  const int P = c.P;
  uint32_t acc = 0;
  for (int i = 0; i < P; i++) {  // cub::DeviceScan::InclusiveSum
    acc += c.tiles_touched[i];
    c.point_offsets[i] = acc;
  }
  c.R = P > 0 ? (int)c.point_offsets[P - 1] : 0;
  const int R = c.R;
  c.keys_unsorted.assign(R, 0);
  c.vals_unsorted.assign(R, 0);
  c.keys.assign(R, 0);
  c.point_list.assign(R, 0);
  for (int idx = 0; idx < P; idx++) {  // duplicateWithKeys :70-112
    if (radii[idx] > 0) {
      uint32_t off = (idx == 0) ? 0 : c.point_offsets[idx - 1];
      unsigned minx, miny, maxx, maxy;
      getRect(c.means2D[2 * idx], c.means2D[2 * idx + 1], radii[idx], c.gx, c.gy, minx, miny, maxx, maxy);
      uint32_t dbits;
      std::memcpy(&dbits, &c.depths[idx], 4);
      for (int y = (int)miny; y < (int)maxy; y++)
        for (int x = (int)minx; x < (int)maxx; x++) {
          uint64_t key = (uint64_t)(y * c.gx + x);
          key <<= 32;
          key |= dbits;
          c.keys_unsorted[off] = key;
          c.vals_unsorted[off] = (uint32_t)idx;
          off++;
        }
    }
  }
  // cub::DeviceRadixSort::SortPairs(..., 0, 32 + bit): stable, on the low 32+bit bits only.
  const int bit = (int)getHigherMsb(c.gx * c.gy);
  const uint64_t mask = (32 + bit) >= 64 ? ~0ull : ((1ull << (32 + bit)) - 1);
  std::vector<uint32_t> perm(R);
  for (int i = 0; i < R; i++) perm[i] = (uint32_t)i;
  std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
    return (c.keys_unsorted[a] & mask) < (c.keys_unsorted[b] & mask);
  });
  for (int i = 0; i < R; i++) {
    c.keys[i] = c.keys_unsorted[perm[i]];
    c.point_list[i] = c.vals_unsorted[perm[i]];
  }
  // cudaMemset(ranges, 0) + identifyTileRanges :117-138
  std::fill(c.ranges.begin(), c.ranges.end(), 0u);
  for (int idx = 0; idx < R; idx++) {
    uint32_t currtile = (uint32_t)(c.keys[idx] >> 32);
    if (idx == 0) c.ranges[2 * currtile] = 0;
    else {
      uint32_t prevtile = (uint32_t)(c.keys[idx - 1] >> 32);
      if (currtile != prevtile) {
        c.ranges[2 * prevtile + 1] = idx;
        c.ranges[2 * currtile] = idx;
      }
    }
    if (idx == R - 1) c.ranges[2 * currtile + 1] = R;
  }
}

// ---------------------------------------------------------------------------------
// A5  renderCUDA forward                                   R/cuda_rasterizer/forward.cu:423-633
// Each pixel is independent; the block-level vote (:506) only stops fetching once
// every pixel is done and does not change any pixel's result.
// ---------------------------------------------------------------------------------
void render_fwd(Ctx& c, const float* viewmatrix, const float* features, const float* normals,
                const float* albedo, const float* roughness, const float* metallic,
                const float* bg_color, bool argmax_depth, bool inference, float* out_color,
                float* out_opacity, float* out_depth, float* out_normal, float* out_normal_view,
                float* out_pos, float* out_albedo, float* out_roughness, float* out_metallic) {
  const int W = c.W, H = c.H;
  const size_t HW = (size_t)H * W;
  uint64_t n_eval = 0, n_contr = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : n_eval, n_contr)
  for (int tile = 0; tile < (int)(c.gx * c.gy); tile++) {
    const int ty = tile / (int)c.gx, tx = tile % (int)c.gx;
    const uint32_t rx = c.ranges[2 * tile], ry = c.ranges[2 * tile + 1];
    for (int ly = 0; ly < BLOCK_Y; ly++)
      for (int lx = 0; lx < BLOCK_X; lx++) {
        const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (!(px < W && py < H)) continue;
        const size_t pix_id = (size_t)W * py + px;
        const float pixfx = (float)px, pixfy = (float)py;
        float T = 1.0f;
        uint32_t contributor = 0, last_contributor = 0;
        float C[3] = {0, 0, 0}, N[3] = {0, 0, 0}, A[3] = {0, 0, 0};
        float Rr = 0, Mm = 0, Dd = 0, O = 0;
        f3 POS = {0, 0, 0};
        float max_weight = 0.0f, except_depth = 0.0f;
        f3 except_pos = {0, 0, 0};
        for (uint32_t k = rx; k < ry; k++) {
          contributor++;
          n_eval++;
          const uint32_t id = c.point_list[k];
          const float dx = c.means2D[2 * id] - pixfx, dy = c.means2D[2 * id + 1] - pixfy;
          const float* co = &c.conic_opacity[4 * (size_t)id];
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0.0f) continue;
          const float alpha = fminf(0.99f, co[3] * expf(power));
          if (alpha < 1.0f / 255.0f) continue;
          const float test_T = T * (1 - alpha);
          if (test_T < 0.0001f) break;  // done = true
          const float weight = alpha * T;
          n_contr++;
          for (int ch = 0; ch < 3; ch++) {
            C[ch] += features[id * 3 + ch] * weight;
            A[ch] += albedo[id * 3 + ch] * weight;
            N[ch] += normals[id * 3 + ch] * weight;
          }
          Rr += roughness[id] * weight;
          Mm += metallic[id] * weight;
          Dd += c.depths[id] * weight;
          POS.x += c.pos_view[3 * id + 0] * weight;
          POS.y += c.pos_view[3 * id + 1] * weight;
          POS.z += c.pos_view[3 * id + 2] * weight;
          O += weight;
          if (weight > max_weight) {
            except_depth = c.depths[id];
            except_pos = {c.pos_view[3 * id], c.pos_view[3 * id + 1], c.pos_view[3 * id + 2]};
            max_weight = weight;
          }
          T = test_T;
          last_contributor = contributor;
        }
        c.final_T[pix_id] = T;
        c.n_contrib[pix_id] = last_contributor;
        f3 N_view = normalize(transformVec4x3({N[0], N[1], N[2]}, viewmatrix));  // NaN if N == 0
        out_normal_view[pix_id] = N_view.x;
        out_normal_view[HW + pix_id] = N_view.y;
        out_normal_view[2 * HW + pix_id] = N_view.z;
        for (int ch = 0; ch < 3; ch++) {
          out_color[ch * HW + pix_id] = C[ch] + T * bg_color[ch];
          out_normal[ch * HW + pix_id] = N[ch];
          out_albedo[ch * HW + pix_id] = A[ch];
        }
        out_roughness[pix_id] = inference ? (Rr + T) : Rr;
        out_metallic[pix_id] = Mm;
        if ((double)O > 1e-6) {
          out_depth[pix_id] = argmax_depth ? except_depth : Dd / O;
          out_pos[pix_id] = argmax_depth ? except_pos.x : POS.x / O;
          out_pos[HW + pix_id] = argmax_depth ? except_pos.y : POS.y / O;
          out_pos[2 * HW + pix_id] = argmax_depth ? except_pos.z : POS.z / O;
        } else {
          out_depth[pix_id] = 0.0f;
          out_pos[pix_id] = 0.0f;
          out_pos[HW + pix_id] = 0.0f;
          out_pos[2 * HW + pix_id] = 0.0f;
        }
        out_opacity[pix_id] = O;
      }
  }
  c.pairs_evaluated = n_eval;
  c.pairs_contributing = n_contr;
}

// ---------------------------------------------------------------------------------
// A9  renderCUDA backward                                  R/cuda_rasterizer/backward.cu:404-630
// Per-Gaussian sums are kept in double (the reference uses fp32 atomicAdd in an
// undefined order; the double sum is the order-free representative).
// ---------------------------------------------------------------------------------
struct RenderGrads {
  std::vector<double> mean2D, conic, opacity, color, normal, albedo, rough, metal, depth;
};

void render_bwd(const Ctx& c, const float* bg_color, const float* colors, const float* /*normals*/,
                const float* dL_dpix_depth, const float* dL_dpix, const float* dL_dpix_opacity,
                const float* dL_dpix_normal, const float* dL_dpix_albedo, const float* dL_dpix_rough,
                const float* dL_dpix_metal, RenderGrads& g) {
  const int W = c.W, H = c.H, P = c.P;
  const size_t HW = (size_t)H * W;
  g.mean2D.assign(3 * (size_t)P, 0.0);
  g.conic.assign(4 * (size_t)P, 0.0);
  g.opacity.assign(P, 0.0);
  g.color.assign(3 * (size_t)P, 0.0);
  g.normal.assign(3 * (size_t)P, 0.0);
  g.albedo.assign(3 * (size_t)P, 0.0);
  g.rough.assign(P, 0.0);
  g.metal.assign(P, 0.0);
  g.depth.assign(P, 0.0);
  const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);
  for (int tile = 0; tile < (int)(c.gx * c.gy); tile++) {
    const int ty = tile / (int)c.gx, tx = tile % (int)c.gx;
    const uint32_t rx = c.ranges[2 * tile], ry = c.ranges[2 * tile + 1];
    const int toDo = (int)(ry - rx);
    for (int ly = 0; ly < BLOCK_Y; ly++)
      for (int lx = 0; lx < BLOCK_X; lx++) {
        const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (!(px < W && py < H)) continue;
        const size_t pix_id = (size_t)W * py + px;
        const float pixfx = (float)px, pixfy = (float)py;
        const float T_final = c.final_T[pix_id];
        float T = T_final;
        const int last_contributor = (int)c.n_contrib[pix_id];
        float last_alpha = 0.0f, accum_opacity = 0.0f;
        float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0};
        float dpix[3], dnrm[3], dalb[3];
        for (int i = 0; i < 3; i++) {
          dpix[i] = dL_dpix[i * HW + pix_id];
          dnrm[i] = dL_dpix_normal[i * HW + pix_id];
          dalb[i] = dL_dpix_albedo[i * HW + pix_id];
        }
        const float dop = dL_dpix_opacity[pix_id], drg = dL_dpix_rough[pix_id];
        const float dmt = dL_dpix_metal[pix_id], ddp = dL_dpix_depth[pix_id];
        if (px == 0 || px == W - 1 || py == 0 || py == H - 1)  // :497-501
          for (int i = 0; i < 3; i++) dnrm[i] = 0.0f;
        for (int q = 0; q < toDo; q++) {
          // reference walks the list from the back; contributor-- happens first (:532)
          const uint32_t contributor = (uint32_t)(toDo - 1 - q);
          if ((int)contributor >= last_contributor) continue;
          const uint32_t id = c.point_list[ry - 1 - q];
          const float dx = c.means2D[2 * id] - pixfx, dy = c.means2D[2 * id + 1] - pixfy;
          const float* co = &c.conic_opacity[4 * (size_t)id];
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0.0f) continue;
          const float G = expf(power);
          const float alpha = fminf(0.99f, co[3] * G);
          if (alpha < 1.0f / 255.0f) continue;
          T = T / (1.f - alpha);
          const float dchannel_dcolor = alpha * T;
          float dL_dalpha = 0.0f;
          for (int ch = 0; ch < 3; ch++) {
            const float cc = colors[id * 3 + ch];
            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
            last_color[ch] = cc;
            dL_dalpha += (cc - accum_rec[ch]) * dpix[ch];
            g.color[3 * (size_t)id + ch] += dchannel_dcolor * dpix[ch];
            g.normal[3 * (size_t)id + ch] += dchannel_dcolor * dnrm[ch];
            g.albedo[3 * (size_t)id + ch] += dchannel_dcolor * dalb[ch];
          }
          g.rough[id] += dchannel_dcolor * drg;
          g.metal[id] += dchannel_dcolor * dmt;
          g.depth[id] += dchannel_dcolor * ddp;
          accum_opacity = last_alpha + (1.f - last_alpha) * accum_opacity;
          dL_dalpha += (1.0f - accum_opacity) * dop;
          dL_dalpha *= T;
          last_alpha = alpha;
          float bg_dot_dpixel = 0;
          for (int i = 0; i < 3; i++) bg_dot_dpixel += bg_color[i] * dpix[i];
          dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
          const float dL_dG = co[3] * dL_dalpha;
          const float gdx = G * dx, gdy = G * dy;
          const float dG_ddelx = -gdx * co[0] - gdy * co[1];
          const float dG_ddely = -gdy * co[2] - gdx * co[1];
          g.mean2D[3 * (size_t)id + 0] += dL_dG * dG_ddelx * ddelx_dx;
          g.mean2D[3 * (size_t)id + 1] += dL_dG * dG_ddely * ddely_dy;
          g.mean2D[3 * (size_t)id + 2] +=
              fabsf(dL_dG * dG_ddelx * ddelx_dx) + fabsf(dL_dG * dG_ddely * ddely_dy);
          g.conic[4 * (size_t)id + 0] += -0.5f * gdx * dx * dL_dG;
          g.conic[4 * (size_t)id + 1] += -0.5f * gdx * dy * dL_dG;
          g.conic[4 * (size_t)id + 3] += -0.5f * gdy * dy * dL_dG;
          g.opacity[id] += G * dL_dalpha;
        }
      }
  }
}

// ---------------------------------------------------------------------------------
// A10  computeCov2DCUDA (backward.cu:145-279), preprocessCUDA bwd (:351-401),
//      computeColorFromSH bwd (:21-140), computeCov3D bwd (:283-346)
// ---------------------------------------------------------------------------------
inline f3 dnormvdv(f3 v, f3 dv) {  // auxiliary.h:118-128
  float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  f3 r;
  r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
  r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
  r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
  return r;
}

void preprocess_bwd(const Ctx& c, int P, int D, int M, const float* means3D, const int* radii,
                    const float* shs, const float* scales, const float* rotations,
                    float scale_modifier, const float* cov3Ds, const float* vm, const float* proj,
                    const float* campos, float h_x, float h_y, float tan_fovx, float tan_fovy,
                    const float* dL_dmean2D /*P,3*/, const float* dL_dconics /*P,4*/,
                    const float* dL_ddepth /*P*/, float* dL_dmeans /*P,3*/, float* dL_dcolor /*P,3*/,
                    float* dL_dcov /*P,6*/, float* dL_dsh /*P,M,3*/, float* dL_dscale, float* dL_drot) {
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    if (!(radii[idx] > 0)) continue;
    // ---- computeCov2DCUDA
    const float* cov3D = cov3Ds + 6 * (size_t)idx;
    f3 mean = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    f3 dL_dconic = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
    Cov2DState s = cov2d_state(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, vm);
    const f3 t = s.t;
    const float x_grad_mul = (s.txtz < -s.limx || s.txtz > s.limx) ? 0 : 1;
    const float y_grad_mul = (s.tytz < -s.limy || s.tytz > s.limy) ? 0 : 1;
    const M3& T = s.T;
    const M3& Vrk = s.Vrk;
    const M3& Wm = s.Wm;
    float a = s.cov.m[0][0] + 0.3f;
    float b = s.cov.m[0][1];
    float cc = s.cov.m[1][1] + 0.3f;
    float denom = a * cc - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float* dcov = dL_dcov + 6 * (size_t)idx;
    if (denom2inv != 0) {
      dL_da = denom2inv * (-cc * cc * dL_dconic.x + 2 * b * cc * dL_dconic.y + (denom - a * cc) * dL_dconic.z);
      dL_dc = denom2inv * (-a * a * dL_dconic.z + 2 * a * b * dL_dconic.y + (denom - a * cc) * dL_dconic.x);
      dL_db = denom2inv * 2 * (b * cc * dL_dconic.x - (denom + 2 * b * b) * dL_dconic.y + a * b * dL_dconic.z);
      dcov[0] = (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
      dcov[3] = (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
      dcov[5] = (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);
      dcov[1] = 2 * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][1] * dL_dc;
      dcov[2] = 2 * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][2] * dL_dc;
      dcov[4] = 2 * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db + 2 * T.m[1][1] * T.m[1][2] * dL_dc;
    } else {
      for (int i = 0; i < 6; i++) dcov[i] = 0;
    }
    float dL_dT00 = 2 * (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_da +
                    (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_db;
    float dL_dT01 = 2 * (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_da +
                    (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_db;
    float dL_dT02 = 2 * (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_da +
                    (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_db;
    float dL_dT10 = 2 * (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc +
                    (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_db;
    float dL_dT11 = 2 * (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc +
                    (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_db;
    float dL_dT12 = 2 * (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc +
                    (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_db;
    float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
    float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
    float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
    float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
    float tz = 1.f / t.z;
    float tz2 = tz * tz;
    float tz3 = tz2 * tz;
    float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
    f3 dL_dmean = transformVec4x3Transpose({dL_dtx, dL_dty, dL_dtz}, vm);
    dL_dmean.x += vm[2] * dL_ddepth[idx];
    dL_dmean.y += vm[6] * dL_ddepth[idx];
    dL_dmean.z += vm[10] * dL_ddepth[idx];
    f3 dmeans = dL_dmean;  // backward.cu:278 overwrites

    // ---- preprocessCUDA (backward.cu:375-392)
    f3 m = mean;
    f4 m_hom = transformPoint4x4(m, proj);
    float m_w = 1.0f / (m_hom.w + 0.0000001f);
    float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    const float g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
    f3 dm2;
    dm2.x = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    dm2.y = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    dm2.z = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    dmeans = dmeans + dm2;

    // ---- computeColorFromSH backward (backward.cu:21-140)
    if (shs) {
      f3 dir_orig = {m.x - campos[0], m.y - campos[1], m.z - campos[2]};
      float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
      f3 dir = {dir_orig.x / len, dir_orig.y / len, dir_orig.z / len};
      const float* sh = shs + (size_t)idx * M * 3;
      auto S = [&](int k) { return f3{sh[3 * k], sh[3 * k + 1], sh[3 * k + 2]}; };
      f3 dL_dRGB = {dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
      dL_dRGB.x *= c.clamped[3 * idx + 0] ? 0 : 1;
      dL_dRGB.y *= c.clamped[3 * idx + 1] ? 0 : 1;
      dL_dRGB.z *= c.clamped[3 * idx + 2] ? 0 : 1;
      f3 dRGBdx = {0, 0, 0}, dRGBdy = {0, 0, 0}, dRGBdz = {0, 0, 0};
      float x = dir.x, y = dir.y, z = dir.z;
      float* dsh = dL_dsh + (size_t)idx * M * 3;
      auto W3 = [&](int k, f3 v) { dsh[3 * k] = v.x; dsh[3 * k + 1] = v.y; dsh[3 * k + 2] = v.z; };
      W3(0, dL_dRGB * SH_C0);
      if (D > 0) {
        float dRGBdsh1 = -SH_C1 * y, dRGBdsh2 = SH_C1 * z, dRGBdsh3 = -SH_C1 * x;
        W3(1, dL_dRGB * dRGBdsh1);
        W3(2, dL_dRGB * dRGBdsh2);
        W3(3, dL_dRGB * dRGBdsh3);
        dRGBdx = S(3) * (-SH_C1);
        dRGBdy = S(1) * (-SH_C1);
        dRGBdz = S(2) * SH_C1;
        if (D > 1) {
          float xx = x * x, yy = y * y, zz = z * z;
          float xy = x * y, yz = y * z, xz = x * z;
          W3(4, dL_dRGB * (SH_C2[0] * xy));
          W3(5, dL_dRGB * (SH_C2[1] * yz));
          W3(6, dL_dRGB * (SH_C2[2] * (2.f * zz - xx - yy)));
          W3(7, dL_dRGB * (SH_C2[3] * xz));
          W3(8, dL_dRGB * (SH_C2[4] * (xx - yy)));
          dRGBdx = dRGBdx + (S(4) * (SH_C2[0] * y) + S(6) * (SH_C2[2] * 2.f * -x) + S(7) * (SH_C2[3] * z) + S(8) * (SH_C2[4] * 2.f * x));
          dRGBdy = dRGBdy + (S(4) * (SH_C2[0] * x) + S(5) * (SH_C2[1] * z) + S(6) * (SH_C2[2] * 2.f * -y) + S(8) * (SH_C2[4] * 2.f * -y));
          dRGBdz = dRGBdz + (S(5) * (SH_C2[1] * y) + S(6) * (SH_C2[2] * 2.f * 2.f * z) + S(7) * (SH_C2[3] * x));
          if (D > 2) {
            W3(9, dL_dRGB * (SH_C3[0] * y * (3.f * xx - yy)));
            W3(10, dL_dRGB * (SH_C3[1] * xy * z));
            W3(11, dL_dRGB * (SH_C3[2] * y * (4.f * zz - xx - yy)));
            W3(12, dL_dRGB * (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)));
            W3(13, dL_dRGB * (SH_C3[4] * x * (4.f * zz - xx - yy)));
            W3(14, dL_dRGB * (SH_C3[5] * z * (xx - yy)));
            W3(15, dL_dRGB * (SH_C3[6] * x * (xx - 3.f * yy)));
            // float * vec3 products evaluate the scalar chain first (left-to-right)
            dRGBdx = dRGBdx + (S(9) * SH_C3[0] * 3.f * 2.f * xy + S(10) * SH_C3[1] * yz + S(11) * SH_C3[2] * -2.f * xy +
                               S(12) * SH_C3[3] * -3.f * 2.f * xz + S(13) * SH_C3[4] * (-3.f * xx + 4.f * zz - yy) +
                               S(14) * SH_C3[5] * 2.f * xz + S(15) * SH_C3[6] * 3.f * (xx - yy));
            dRGBdy = dRGBdy + (S(9) * SH_C3[0] * 3.f * (xx - yy) + S(10) * SH_C3[1] * xz + S(11) * SH_C3[2] * (-3.f * yy + 4.f * zz - xx) +
                               S(12) * SH_C3[3] * -3.f * 2.f * yz + S(13) * SH_C3[4] * -2.f * xy + S(14) * SH_C3[5] * -2.f * yz +
                               S(15) * SH_C3[6] * -3.f * 2.f * xy);
            dRGBdz = dRGBdz + (S(10) * SH_C3[1] * xy + S(11) * SH_C3[2] * 4.f * 2.f * yz + S(12) * SH_C3[3] * 3.f * (2.f * zz - xx - yy) +
                               S(13) * SH_C3[4] * 4.f * 2.f * xz + S(14) * SH_C3[5] * (xx - yy));
          }
        }
      }
      f3 dL_ddir = {dot(dRGBdx, dL_dRGB), dot(dRGBdy, dL_dRGB), dot(dRGBdz, dL_dRGB)};
      f3 dsm = dnormvdv(dir_orig, dL_ddir);
      dmeans = dmeans + dsm;
    }
    dL_dmeans[3 * idx] = dmeans.x;
    dL_dmeans[3 * idx + 1] = dmeans.y;
    dL_dmeans[3 * idx + 2] = dmeans.z;

    // ---- computeCov3D backward (backward.cu:283-346)
    if (scales) {
      const float* rot = rotations + 4 * (size_t)idx;
      float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
      M3 Rm = mat3(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
      M3 S = mat3(1, 0, 0, 0, 1, 0, 0, 0, 1);
      f3 sv = {scale_modifier * scales[3 * idx], scale_modifier * scales[3 * idx + 1], scale_modifier * scales[3 * idx + 2]};
      S.m[0][0] = sv.x;
      S.m[1][1] = sv.y;
      S.m[2][2] = sv.z;
      M3 Mm = mul(S, Rm);
      const float* d3 = dL_dcov + 6 * (size_t)idx;
      M3 dL_dSigma = mat3(d3[0], 0.5f * d3[1], 0.5f * d3[2], 0.5f * d3[1], d3[3], 0.5f * d3[4],
                          0.5f * d3[2], 0.5f * d3[4], d3[5]);
      M3 M2;  // 2.0f * M
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M2.m[i][j] = Mm.m[i][j] * 2.0f;
      M3 dL_dM = mul(M2, dL_dSigma);
      M3 Rt = transpose(Rm);
      M3 dL_dMt = transpose(dL_dM);
      auto coldot = [](const M3& A, int ca, const M3& B, int cb) {
        return A.m[ca][0] * B.m[cb][0] + A.m[ca][1] * B.m[cb][1] + A.m[ca][2] * B.m[cb][2];
      };
      dL_dscale[3 * idx + 0] = coldot(Rt, 0, dL_dMt, 0);
      dL_dscale[3 * idx + 1] = coldot(Rt, 1, dL_dMt, 1);
      dL_dscale[3 * idx + 2] = coldot(Rt, 2, dL_dMt, 2);
      for (int j = 0; j < 3; j++) {
        dL_dMt.m[0][j] *= sv.x;
        dL_dMt.m[1][j] *= sv.y;
        dL_dMt.m[2][j] *= sv.z;
      }
      const M3& d = dL_dMt;
      float qx = 2 * z * (d.m[0][1] - d.m[1][0]) + 2 * y * (d.m[2][0] - d.m[0][2]) + 2 * x * (d.m[1][2] - d.m[2][1]);
      float qy = 2 * y * (d.m[1][0] + d.m[0][1]) + 2 * z * (d.m[2][0] + d.m[0][2]) + 2 * r * (d.m[1][2] - d.m[2][1]) - 4 * x * (d.m[2][2] + d.m[1][1]);
      float qz = 2 * x * (d.m[1][0] + d.m[0][1]) + 2 * r * (d.m[2][0] - d.m[0][2]) + 2 * z * (d.m[1][2] + d.m[2][1]) - 4 * y * (d.m[2][2] + d.m[0][0]);
      float qw = 2 * r * (d.m[0][1] - d.m[1][0]) + 2 * x * (d.m[2][0] + d.m[0][2]) + 2 * y * (d.m[1][2] + d.m[2][1]) - 4 * z * (d.m[1][1] + d.m[0][0]);
      dL_drot[4 * idx + 0] = qx;
      dL_drot[4 * idx + 1] = qy;
      dL_drot[4 * idx + 2] = qz;
      dL_drot[4 * idx + 3] = qw;
    }
  }
}

// ---------------------------------------------------------------------------------
// GI pass helpers: ssr.h:103-135
// ---------------------------------------------------------------------------------
inline f3 get_position(int x, int y, float cx, float cy, float fx, float fy, float depth) {
  f3 dir = {((float)x - cx) / fx, ((float)y - cy) / fy, 1.0f};
  return dir * depth;
}
inline void get_coord(float cx, float cy, float fx, float fy, f3 pos, int& ox, int& oy) {
  f3 dir = {pos.x / (pos.z + 0.0000001f), pos.y / (pos.z + 0.0000001f), 1.0f};
  ox = f2i(roundf(dir.x * fx + cx));
  oy = f2i(roundf(dir.y * fy + cy));
}

}  // namespace

// =====================================================================================
// C ABI used by the tests (ctypes)
// =====================================================================================
extern "C" {

void* orc_create() { return new Ctx(); }
void orc_destroy(void* h) { delete (Ctx*)h; }
void orc_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int orc_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
uint32_t orc_higher_msb(uint32_t n) { return getHigherMsb(n); }

// Rasterizer::forward, R/cuda_rasterizer/rasterizer_impl.cu:486-672. Returns num_rendered.
int orc_forward(void* h, int P, int D, int M, const float* background, int width, int height,
                const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* normal, const float* albedo,
                const float* roughness, const float* metallic, const float* scales,
                float scale_modifier, const float* rotations, const float* cov3D_precomp,
                const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                float tan_fovx, float tan_fovy, int argmax_depth, int inference, float* out_color,
                float* out_opacity, float* out_depth, float* out_normal, float* out_normal_view,
                float* out_pos, float* out_albedo, float* out_roughness, float* out_metallic,
                int* radii) {
  Ctx& c = *(Ctx*)h;
  c.P = P; c.W = width; c.H = height; c.M = M; c.D = D;
  c.gx = (width + BLOCK_X - 1) / BLOCK_X;
  c.gy = (height + BLOCK_Y - 1) / BLOCK_Y;
  const float focal_y = height / (2.0f * tan_fovy);
  const float focal_x = width / (2.0f * tan_fovx);
  const size_t N = (size_t)width * height;
  c.depths.assign(P, 0); c.pos_view.assign(3 * (size_t)P, 0); c.means2D.assign(2 * (size_t)P, 0);
  c.cov3D.assign(6 * (size_t)P, 0); c.conic_opacity.assign(4 * (size_t)P, 0); c.rgb.assign(3 * (size_t)P, 0);
  c.clamped.assign(3 * (size_t)P, 0); c.radii.assign(P, 0);
  c.tiles_touched.assign(P, 0); c.point_offsets.assign(P, 0);
  c.ranges.assign(2 * (size_t)c.gx * c.gy, 0);
  c.final_T.assign(N, 0); c.n_contrib.assign(N, 0);
  if (P == 0) { c.R = 0; return 0; }
  preprocess_fwd(c, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs,
                 cov3D_precomp, colors_precomp, viewmatrix, projmatrix, cam_pos, width, height,
                 tan_fovx, tan_fovy, focal_x, focal_y, radii);
  std::memcpy(c.radii.data(), radii, sizeof(int) * P);
  bin_and_sort(c, radii);
  const float* feature_ptr = colors_precomp != nullptr ? colors_precomp : c.rgb.data();
  render_fwd(c, viewmatrix, feature_ptr, normal, albedo, roughness, metallic, background,
             argmax_depth != 0, inference != 0, out_color, out_opacity, out_depth, out_normal,
             out_normal_view, out_pos, out_albedo, out_roughness, out_metallic);
  return c.R;
}

// Rasterizer::backward, rasterizer_impl.cu:676-803 (uses the state kept in the context)
void orc_backward(void* h, int P, int D, int M, const float* background, int width, int height,
                  const float* means3D, const float* shs, const float* colors_precomp,
                  const float* normal, const float* scales, const float* rotations,
                  const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                  const float* cam_pos, float scale_modifier, float tan_fovx, float tan_fovy,
                  const float* dL_dpix_depth, const float* dL_dpix, const float* dL_dpix_opacity,
                  const float* dL_dpix_normal, const float* dL_dpix_albedo,
                  const float* dL_dpix_roughness, const float* dL_dpix_metallic, float* dL_dmean2D,
                  float* dL_dopacity, float* dL_dnormal, float* dL_dalbedo, float* dL_droughness,
                  float* dL_dmetallic, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                  float* dL_dsh, float* dL_dscale, float* dL_drot) {
  Ctx& c = *(Ctx*)h;
  (void)width; (void)height;
  const float focal_y = c.H / (2.0f * tan_fovy);
  const float focal_x = c.W / (2.0f * tan_fovx);
  const float* color_ptr = colors_precomp ? colors_precomp : c.rgb.data();
  RenderGrads g;
  render_bwd(c, background, color_ptr, normal, dL_dpix_depth, dL_dpix, dL_dpix_opacity,
             dL_dpix_normal, dL_dpix_albedo, dL_dpix_roughness, dL_dpix_metallic, g);
  c.dL_dconic.assign(4 * (size_t)P, 0);
  c.dL_ddepth.assign(P, 0);
  for (size_t i = 0; i < 3 * (size_t)P; i++) {
    dL_dmean2D[i] = (float)g.mean2D[i];
    dL_dcolor[i] = (float)g.color[i];
    dL_dnormal[i] = (float)g.normal[i];
    dL_dalbedo[i] = (float)g.albedo[i];
  }
  for (size_t i = 0; i < 4 * (size_t)P; i++) c.dL_dconic[i] = (float)g.conic[i];
  for (int i = 0; i < P; i++) {
    dL_dopacity[i] = (float)g.opacity[i];
    dL_droughness[i] = (float)g.rough[i];
    dL_dmetallic[i] = (float)g.metal[i];
    c.dL_ddepth[i] = (float)g.depth[i];
  }
  const float* cov3D_ptr = cov3D_precomp ? cov3D_precomp : c.cov3D.data();
  preprocess_bwd(c, P, D, M, means3D, c.radii.data(), shs, scales, rotations, scale_modifier,
                 cov3D_ptr, viewmatrix, projmatrix, cam_pos, focal_x, focal_y, tan_fovx, tan_fovy,
                 dL_dmean2D, c.dL_dconic.data(), c.dL_ddepth.data(), dL_dmean3D, dL_dcolor,
                 dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
}

// state accessors: which = index below; returns element count, copies into dst if non-null
// 0 depths f32[P] 1 pos_view f32[3P] 2 means2D f32[2P] 3 cov3D f32[6P] 4 conic_opacity f32[4P]
// 5 rgb f32[3P] 6 clamped u8[3P] 7 tiles_touched u32[P] 8 point_offsets u32[P]
// 9 keys_unsorted u64[R] 10 vals_unsorted u32[R] 11 keys u64[R] 12 point_list u32[R]
// 13 ranges u32[2T] 14 final_T f32[N] 15 n_contrib u32[N] 16 dL_dconic f32[4P] 17 dL_ddepth f32[P]
size_t orc_state(void* h, int which, void* dst) {
  Ctx& c = *(Ctx*)h;
#define CP(v)                                                            \
  {                                                                      \
    if (dst && !c.v.empty()) std::memcpy(dst, c.v.data(), c.v.size() * sizeof(c.v[0])); \
    return c.v.size();                                                   \
  }
  switch (which) {
    case 0: CP(depths) case 1: CP(pos_view) case 2: CP(means2D) case 3: CP(cov3D)
    case 4: CP(conic_opacity) case 5: CP(rgb) case 6: CP(clamped) case 7: CP(tiles_touched)
    case 8: CP(point_offsets) case 9: CP(keys_unsorted) case 10: CP(vals_unsorted) case 11: CP(keys)
    case 12: CP(point_list) case 13: CP(ranges) case 14: CP(final_T) case 15: CP(n_contrib)
    case 16: CP(dL_dconic) case 17: CP(dL_ddepth)
  }
#undef CP
  return 0;
}
void orc_counters(void* h, uint64_t* out3) {
  Ctx& c = *(Ctx*)h;
  out3[0] = (uint64_t)c.R;
  out3[1] = c.pairs_evaluated;
  out3[2] = c.pairs_contributing;
}

// checkFrustum / markVisible: rasterizer_impl.cu:54-66, 141-153
void orc_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present) {
  for (int idx = 0; idx < P; idx++) {
    f3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    f3 pv = transformPoint4x3(p, viewmatrix);
    present[idx] = !(pv.z <= 0.2f);
  }
}

// A6 depthmapToNormalCUDA: forward.cu:914-1032.  Outputs must be pre-zeroed by the caller
// (rasterize_points.cu:394-395 allocates them with torch::full(0)).
void orc_depth_to_normal(int W, int H, float focal_x, float focal_y, const float* viewmatrix,
                         const float* out_depth, float* normal_from_depth, float* depth_pos) {
  const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const size_t pix_id = (size_t)W * y + x;
      if (x <= 0 || x >= W - 1 || y <= 0 || y >= H - 1) continue;
      const float depth_thresh = 0.01f;
      const float depth = out_depth[pix_id];
      float cx = float(W) / 2.0f, cy = float(H) / 2.0f;
      f3 pos = get_position(x, y, cx, cy, focal_x, focal_y, depth);
      depth_pos[pix_id] = pos.x;
      depth_pos[HW + pix_id] = pos.y;
      depth_pos[2 * HW + pix_id] = pos.z;
      if (depth < depth_thresh) continue;
      bool skip = false;
      const int pad = 2;
      for (int dx = -pad; dx < pad + 1 && !skip; ++dx) {
        if (x + dx < 0 || x + dx > W - 1) { skip = true; break; }
        for (int dy = -pad; dy < pad + 1; ++dy) {
          if (y + dy < 0 || y + dy > H - 1) { skip = true; break; }
          if (out_depth[(std::ptrdiff_t)pix_id + (std::ptrdiff_t)W * dy + dx] < depth_thresh) { skip = true; break; }
        }
      }
      if (skip) continue;
      auto Dp = [&](int dx, int dy) { return out_depth[(std::ptrdiff_t)pix_id + (std::ptrdiff_t)W * dy + dx]; };
      f3 pos_aa = get_position(x, y - 1, cx, cy, focal_x, focal_y, Dp(0, -1));
      f3 pos_bb = get_position(x + 1, y, cx, cy, focal_x, focal_y, Dp(1, 0));
      f3 pos_cc = get_position(x, y + 1, cx, cy, focal_x, focal_y, Dp(0, 1));
      f3 pos_dd = get_position(x - 1, y, cx, cy, focal_x, focal_y, Dp(-1, 0));
      f3 pos_ab = get_position(x + 1, y - 1, cx, cy, focal_x, focal_y, Dp(1, -1));
      f3 pos_bc = get_position(x + 1, y + 1, cx, cy, focal_x, focal_y, Dp(1, 1));
      f3 pos_cd = get_position(x - 1, y + 1, cx, cy, focal_x, focal_y, Dp(-1, 1));
      f3 pos_da = get_position(x - 1, y - 1, cx, cy, focal_x, focal_y, Dp(-1, -1));
      f3 edge_a = pos_da - pos_ab, edge_b = pos_ab - pos_bc, edge_c = pos_bc - pos_cd, edge_d = pos_cd - pos_da;
      f3 edge_ac = pos_cc - pos_aa, edge_bd = pos_dd - pos_bb;
      f3 edge_cdab = pos_ab - pos_cd, edge_bcad = pos_da - pos_bc;
      f3 n1 = cross(edge_a, edge_d), n2 = cross(edge_d, edge_c), n3 = cross(edge_c, edge_b);
      f3 n4 = cross(edge_b, edge_a), n5 = cross(edge_ac, edge_bd), n6 = cross(edge_bcad, edge_cdab);
      f3 normal = div_s(normalize(n1) + normalize(n2) + normalize(n3) + normalize(n4) + normalize(n5) + normalize(n6), 6);
      const float* vm = viewmatrix;
      normal_from_depth[pix_id] = vm[0] * normal.x + vm[1] * normal.y + vm[2] * normal.z;
      normal_from_depth[HW + pix_id] = vm[4] * normal.x + vm[5] * normal.y + vm[6] * normal.z;
      normal_from_depth[2 * HW + pix_id] = vm[8] * normal.x + vm[9] * normal.y + vm[10] * normal.z;
    }
}

// A7 SSAOCUDA: forward.cu:635-724.  `occlusion` is pre-filled with 1.0 by the caller
// (rasterize_points.cu:420); every in-image pixel is overwritten anyway.
void orc_ssao(int W, int H, float focal_x, float focal_y, float radius, float bias, float thick,
              float delta, int step, int start, const float* out_normal, const float* out_pos,
              float* occlusion) {
  const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const size_t pix_id = (size_t)W * y + x;
      f3 normal = normalize({out_normal[pix_id], out_normal[HW + pix_id], out_normal[2 * HW + pix_id]});
      f3 pos = {out_pos[pix_id], out_pos[HW + pix_id], out_pos[2 * HW + pix_id]};
      f3 up = {0.0f, 1.0f, 0.0f};
      float rndot = dot(up, normal);
      f3 untangent = {up.x - normal.x * rndot, up.y - normal.y * rndot, up.z - normal.z * rndot};
      f3 tangent = normalize(untangent);
      f3 bitangent = normalize(cross(normal, tangent));
      float TBN[9] = {tangent.x, tangent.y, tangent.z, bitangent.x, bitangent.y, bitangent.z, normal.x, normal.y, normal.z};
      float occ = 0.0f;
      float sampleDelta = delta * PI_F;
      float nrSamples = 0.0f;
      for (float phi = 0.0f; (double)phi < 2.0 * (double)PI_F; phi += sampleDelta) {
        for (float theta = 0.0f; (double)theta <= 0.5 * (double)PI_F; theta = (float)((double)theta + (double)sampleDelta * 0.5)) {
          float cosh_ = cosf(theta);
          f3 tangentSample = normalize({sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta)});
          f3 sampleVec = transformVec3x3(tangentSample, TBN);
          nrSamples += cosh_ * sinf(theta);
          for (int j = start; j < step; ++j) {
            f3 sp;
            sp.x = pos.x + sampleVec.x * j * (1 + pos.z / 100) * (1 + pos.z / 100) * radius / step;
            sp.y = pos.y + sampleVec.y * j * (1 + pos.z / 100) * (1 + pos.z / 100) * radius / step;
            sp.z = pos.z + sampleVec.z * j * (1 + pos.z / 100) * (1 + pos.z / 100) * radius / step;
            float cx = float(W) / 2.0f, cy = float(H) / 2.0f;
            int ix, iy;
            get_coord(cx, cy, focal_x, focal_y, sp, ix, iy);
            if (ix < 0 || ix > W - 1) break;
            if (iy < 0 || iy > H - 1) break;
            float sampleDepth = out_pos[2 * HW + (size_t)W * iy + ix];
            if (sampleDepth <= sp.z + bias && sampleDepth >= sp.z - thick) {
              occ += cosh_ * sinf(theta);
              break;
            }
          }
        }
      }
      if (nrSamples > 0.0f)
        occlusion[pix_id] = fmaxf(0.0f, fminf(1.0f, (float)(1.0 - (double)(occ / nrSamples))));
      else
        occlusion[pix_id] = 1.0f;
    }
}

// A8 SSRCUDA forward: forward.cu:726-909, fresnelSchlick ssr.h:13-16
void orc_ssr(int W, int H, float focal_x, float focal_y, float radius, float bias, float thick,
             float delta, int step, int start, const float* out_normal, const float* out_pos,
             const float* out_rgb, const float* out_albedo, const float* /*out_roughness*/,
             const float* out_metallic, const float* out_F0, float* color, float* abd) {
  const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const size_t pix_id = (size_t)W * y + x;
      f3 pos = {out_pos[pix_id], out_pos[HW + pix_id], out_pos[2 * HW + pix_id]};
      f3 diffuse = {0, 0, 0}, gd = {0, 0, 0};
      f3 normal = normalize({out_normal[pix_id], out_normal[HW + pix_id], out_normal[2 * HW + pix_id]});
      f3 N = normal;
      f3 up = {0.0f, 1.0f, 0.0f};
      float rndot = dot(up, normal);
      f3 untangent = {up.x - normal.x * rndot, up.y - normal.y * rndot, up.z - normal.z * rndot};
      f3 tangent = normalize(untangent);
      f3 bitangent = normalize(cross(normal, tangent));
      float TBN[9] = {tangent.x, tangent.y, tangent.z, bitangent.x, bitangent.y, bitangent.z, normal.x, normal.y, normal.z};
      f3 albedo = {out_albedo[pix_id], out_albedo[HW + pix_id], out_albedo[2 * HW + pix_id]};
      f3 F0 = {out_F0[pix_id], out_F0[HW + pix_id], out_F0[2 * HW + pix_id]};
      float metallic = out_metallic[pix_id];
      f3 V = normalize({-pos.x, -pos.y, -pos.z});
      // fresnelSchlick(cosTheta, F0) = F0 + (1.0 - F0) * pow(fminf(fmaxf(1.0 - cosTheta, 0.000001), 1.0), 5.0)
      float cosTheta = fmaxf(dot(N, V), (float)0.0000001);
      float pw = (float)pow((double)fminf(fmaxf((float)(1.0 - (double)cosTheta), (float)0.000001), 1.0f), 5.0);
      f3 F = {F0.x + (1.0f - F0.x) * pw, F0.y + (1.0f - F0.y) * pw, F0.z + (1.0f - F0.z) * pw};
      f3 kD = {(float)(1.0 - (double)F.x), (float)(1.0 - (double)F.y), (float)(1.0 - (double)F.z)};
      kD.x = (float)((double)kD.x * (1.0 - (double)metallic));
      kD.y = (float)((double)kD.y * (1.0 - (double)metallic));
      kD.z = (float)((double)kD.z * (1.0 - (double)metallic));
      float sampleDelta = delta * PI_F;
      float nrSamples = 0.0f;
      for (float phi = 0.0f; (double)phi < 2.0 * (double)PI_F; phi += sampleDelta) {
        for (float theta = 0.0f; (double)theta <= 0.5 * (double)PI_F; theta = (float)((double)theta + (double)sampleDelta * 0.5)) {
          f3 tangentSample = normalize({sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta)});
          f3 sampleVec = transformVec3x3(tangentSample, TBN);
          nrSamples += 1;
          for (int j = start; j < step; ++j) {
            f3 sp;
            sp.x = pos.x + sampleVec.x * j * (1 + pos.z / 100) * (1 + pos.z / 100) * radius / step;
            sp.y = pos.y + sampleVec.y * j * (1 + pos.z / 100) * (1 + pos.z / 100) * radius / step;
            sp.z = pos.z + sampleVec.z * j * (1 + pos.z / 100) * (1 + pos.z / 100) * radius / step;
            float cx = float(W) / 2.0f, cy = float(H) / 2.0f;
            int ix, iy;
            get_coord(cx, cy, focal_x, focal_y, sp, ix, iy);
            if (ix < 0 || ix > W - 1) break;
            if (iy < 0 || iy > H - 1) break;
            const size_t q = (size_t)W * iy + ix;
            f3 rgb = {out_rgb[q], out_rgb[HW + q], out_rgb[2 * HW + q]};
            float sampleDepth = out_pos[2 * HW + q];
            if (sampleDepth <= sp.z + bias && sampleDepth >= sp.z - thick) {
              diffuse.x += rgb.x * cosf(theta) * sinf(theta);
              diffuse.y += rgb.y * cosf(theta) * sinf(theta);
              diffuse.z += rgb.z * cosf(theta) * sinf(theta);
              break;
            }
          }
        }
      }
      if (nrSamples > 0.0f) {
        gd.x = (float)((double)(PI_F * diffuse.x) * (1.0 / (double)nrSamples) * (double)kD.x);
        gd.y = (float)((double)(PI_F * diffuse.y) * (1.0 / (double)nrSamples) * (double)kD.y);
        gd.z = (float)((double)(PI_F * diffuse.z) * (1.0 / (double)nrSamples) * (double)kD.z);
        diffuse = {gd.x * albedo.x, gd.y * albedo.y, gd.z * albedo.z};
      } else {
        diffuse = {(float)0.0000001, (float)0.0000001, (float)0.0000001};
        gd = diffuse;
      }
      color[pix_id] = diffuse.x;
      color[HW + pix_id] = diffuse.y;
      color[2 * HW + pix_id] = diffuse.z;
      abd[pix_id] = gd.x;
      abd[HW + pix_id] = gd.y;
      abd[2 * HW + pix_id] = gd.z;
    }
}

// Number of (phi, theta) rays the fp32-accumulated loops of SSAO/SSR visit (forward.cu:679-681)
void orc_gi_ray_counts(float delta, int* n_phi, int* n_theta) {
  float sampleDelta = delta * PI_F;
  int a = 0, b = 0;
  for (float phi = 0.0f; (double)phi < 2.0 * (double)PI_F; phi += sampleDelta) a++;
  for (float theta = 0.0f; (double)theta <= 0.5 * (double)PI_F; theta = (float)((double)theta + (double)sampleDelta * 0.5)) b++;
  *n_phi = a;
  *n_theta = b;
}

// A11 kornia.filters.median_blur(x, (3,3)): zero padding, torch.median over the 9 taps
// (5th smallest, NaN-propagating).  kornia is absent from the container and unpinned in
// the reference's environment.yml -> parity unpinned; this is the build's own definition.
void orc_median3x3(int C, int H, int W, const float* in, float* out) {
#pragma omp parallel for schedule(static)
  for (int cy = 0; cy < C * H; cy++) {
    const int ch = cy / H, y = cy % H;
    const float* src = in + (size_t)ch * H * W;
    float* dst = out + (size_t)ch * H * W;
    for (int x = 0; x < W; x++) {
      float v[9];
      bool has_nan = false;
      int k = 0;
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          int yy = y + dy, xx = x + dx;
          float t = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0.0f : src[(size_t)yy * W + xx];
          if (t != t) has_nan = true;
          v[k++] = t;
        }
      if (has_nan) {
        dst[(size_t)y * W + x] = NAN;
      } else {
        std::sort(v, v + 9);
        dst[(size_t)y * W + x] = v[4];
      }
    }
  }
}

// A11 kornia.filters.bilateral_blur(x, (3,3), sigma_color, (sigma_sx, sigma_sy)),
// border_type='reflect', color_distance_type='l1'.  Parity unpinned (see above).
void orc_bilateral3x3(int C, int H, int W, float sigma_color, float sigma_sx, float sigma_sy,
                      const float* in, float* out) {
  // get_gaussian_kernel1d(3, sigma): x = {-1,0,1}; exp(-x^2/(2 sigma^2)); normalised
  auto k1d = [](float sigma, float* k) {
    float s = 0;
    for (int i = 0; i < 3; i++) {
      float xv = (float)(i - 1);
      k[i] = expf(-(xv * xv) / (2.0f * sigma * sigma));
      s += k[i];
    }
    for (int i = 0; i < 3; i++) k[i] /= s;
  };
  float ky[3], kx[3];
  k1d(sigma_sy, ky);
  k1d(sigma_sx, kx);
  auto refl = [](int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); };
  const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      float num[8] = {0}, den = 0;
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          const int yy = refl(y + dy, H), xx = refl(x + dx, W);
          float dist = 0;
          for (int ch = 0; ch < C; ch++) dist += fabsf(in[ch * HW + (size_t)yy * W + xx] - in[ch * HW + (size_t)y * W + x]);
          const float color_k = expf((-0.5f / (sigma_color * sigma_color)) * (dist * dist));
          const float kk = (ky[dy + 1] * kx[dx + 1]) * color_k;
          for (int ch = 0; ch < C; ch++) num[ch] += in[ch * HW + (size_t)yy * W + xx] * kk;
          den += kk;
        }
      for (int ch = 0; ch < C; ch++) out[ch * HW + (size_t)y * W + x] = num[ch] / den;
    }
}

}  // extern "C"
