// gi.hip -- screen-space passes over the G-buffer: depth->normal, SSAO, SSR (indirect
// diffuse) and the two 3x3 filters the operator applies between them.
//
// Reference behaviour restated (R/ = submodules/diff-gaussian-rasterization):
//   depthmapToNormalCUDA  R/cuda_rasterizer/forward.cu:914-1032  (get_position ssr.h:103-118)
//   SSAOCUDA              forward.cu:635-724                      (get_coord   ssr.h:120-135)
//   SSRCUDA               forward.cu:726-909                      (fresnelSchlick ssr.h:13-16)
//   kornia median_blur / bilateral_blur as called at
//   R/diff_gaussian_rasterization/__init__.py:478, 491, 504 (kornia is third party and
//   unpinned in the reference: parity unpinned, definitions in include/gigs_hip.h)
//
// MI355X design for SSAO / SSR (the dominant cost at start < step: rays * steps dependent
// gathers per pixel):
//   * the (phi, theta) ray set is pixel-invariant.  The host builds it ONCE per `delta` with
//     the reference's fp32 loop accumulation (`phi += d*pi`, `theta += d*pi/2`) and libm
//     sinf/cosf, and keeps it in a small per-device table.  In the kernel the ray index is
//     wave-uniform, so the table is read with scalar loads (s_load_dwordx4/x8) into SGPRs
//     and costs no vector memory or LDS traffic; the loop-count subtleties (16 not 17 theta
//     samples at delta = 0.0625) are decided on the host exactly as the oracle decides them;
//   * 8x8 pixels per wave so the z-plane gathers of neighbouring lanes share cache lines (the
//     z plane is 2.5 MB at 800x800 and stays L2 resident);
//   * per-step arithmetic keeps the reference's operation order and IEEE division, so pixel
//     coordinates (roundf of a projected float) are bit-identical to the oracle's; the only
//     exact rewrite is `x / step` -> `x * (1/step)` when step is a power of two;
//   * SSR reads the rgb planes only on a hit (the reference reads them every step but uses
//     them only on a hit: forward.cu:820-826).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "gigs_common.h"
#include "pixel_ops.h"

namespace gigs {

constexpr float kPiF = 3.14159265358979323846f;

// ------------------------------------------------------------------------------------------
// depth -> normal
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ v3 get_position(int x, int y, float cx, float cy, float fx, float fy,
                                           float depth) {
  const v3 dir = {((float)x - cx) / fx, ((float)y - cy) / fy, 1.0f};
  return dir * depth;
}

__global__ void __launch_bounds__(256)
depth_to_normal_kernel(int W, int H, float fx, float fy, const float* __restrict__ vm,
                       const float* __restrict__ depth_map, float* __restrict__ normal_out,
                       float* __restrict__ depth_pos) {
  const int x = blockIdx.x * 16 + (threadIdx.x & 15);
  const int y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= W || y >= H) return;
  const size_t HW = (size_t)H * W;
  const size_t pix_id = (size_t)W * y + x;
  // The reference returns early in three places and relies on the caller's zero fill
  // (rasterize_points.cu:394-395); the zeros are written here so the caller can pass raw memory.
  normal_out[pix_id] = 0.0f; normal_out[HW + pix_id] = 0.0f; normal_out[2 * HW + pix_id] = 0.0f;
  if (x <= 0 || x >= W - 1 || y <= 0 || y >= H - 1) {
    depth_pos[pix_id] = 0.0f; depth_pos[HW + pix_id] = 0.0f; depth_pos[2 * HW + pix_id] = 0.0f;
    return;
  }
  const float depth_thresh = 0.01f;
  const float depth = depth_map[pix_id];
  const float cx = float(W) / 2.0f, cy = float(H) / 2.0f;
  const v3 pos = get_position(x, y, cx, cy, fx, fy, depth);
  depth_pos[pix_id] = pos.x;
  depth_pos[HW + pix_id] = pos.y;
  depth_pos[2 * HW + pix_id] = pos.z;
  if (depth < depth_thresh) return;
  // 5x5 validity window (forward.cu:979-986)
  const int pad = 2;
  for (int dx = -pad; dx < pad + 1; ++dx) {
    if (x + dx < 0 || x + dx > W - 1) return;
    for (int dy = -pad; dy < pad + 1; ++dy) {
      if (y + dy < 0 || y + dy > H - 1) return;
      if (depth_map[(ptrdiff_t)pix_id + (ptrdiff_t)W * dy + dx] < depth_thresh) return;
    }
  }
#define DP(dx, dy) depth_map[(ptrdiff_t)pix_id + (ptrdiff_t)W * (dy) + (dx)]
  const v3 pos_aa = get_position(x, y - 1, cx, cy, fx, fy, DP(0, -1));
  const v3 pos_bb = get_position(x + 1, y, cx, cy, fx, fy, DP(1, 0));
  const v3 pos_cc = get_position(x, y + 1, cx, cy, fx, fy, DP(0, 1));
  const v3 pos_dd = get_position(x - 1, y, cx, cy, fx, fy, DP(-1, 0));
  const v3 pos_ab = get_position(x + 1, y - 1, cx, cy, fx, fy, DP(1, -1));
  const v3 pos_bc = get_position(x + 1, y + 1, cx, cy, fx, fy, DP(1, 1));
  const v3 pos_cd = get_position(x - 1, y + 1, cx, cy, fx, fy, DP(-1, 1));
  const v3 pos_da = get_position(x - 1, y - 1, cx, cy, fx, fy, DP(-1, -1));
#undef DP
  const v3 edge_a = pos_da - pos_ab, edge_b = pos_ab - pos_bc, edge_c = pos_bc - pos_cd,
           edge_d = pos_cd - pos_da;
  const v3 edge_ac = pos_cc - pos_aa, edge_bd = pos_dd - pos_bb;
  const v3 edge_cdab = pos_ab - pos_cd, edge_bcad = pos_da - pos_bc;
  const v3 n1 = cross3(edge_a, edge_d), n2 = cross3(edge_d, edge_c), n3 = cross3(edge_c, edge_b);
  const v3 n4 = cross3(edge_b, edge_a), n5 = cross3(edge_ac, edge_bd), n6 = cross3(edge_bcad, edge_cdab);
  const v3 sum = normalize3(n1) + normalize3(n2) + normalize3(n3) + normalize3(n4) + normalize3(n5) + normalize3(n6);
  const float inv6 = 1.0f / 6.0f;  // float3 / float multiplies by the reciprocal (vec_math.h:476)
  const v3 normal = sum * inv6;
  normal_out[pix_id] = vm[0] * normal.x + vm[1] * normal.y + vm[2] * normal.z;
  normal_out[HW + pix_id] = vm[4] * normal.x + vm[5] * normal.y + vm[6] * normal.z;
  normal_out[2 * HW + pix_id] = vm[8] * normal.x + vm[9] * normal.y + vm[10] * normal.z;
}

void launch_depth_to_normal(int W, int H, float fx, float fy, const float* viewmatrix,
                            const float* depth, float* normal, float* depth_pos, hipStream_t s) {
  dim3 grid((W + 15) / 16, (H + 15) / 16);
  hipLaunchKernelGGL(depth_to_normal_kernel, grid, dim3(256), 0, s, W, H, fx, fy, viewmatrix,
                     depth, normal, depth_pos);
}

// ------------------------------------------------------------------------------------------
// Ray table: per ray two float4 -- {ts.x, ts.y, ts.z, cos(theta)}, {sin(theta), w, 0, 0}
// with ts = normalize(sin t cos p, sin t sin p, cos t) and w = cos t * sin t.
//
// Zero-weight rays.  The reference's theta loop starts at 0 (forward.cu:681, 797): sin(0) = 0, so for every azimuth
// the first ray is the SAME direction (0, 0, 1) -- the pixel's normal -- with weight w = cos * sin = 0.  SSAO adds
// `w` for a hit (forward.cu:711: occ += 0, a no-op) and `w` to the sample sum (:690, again + 0); SSR adds
// rgb * cos * sin = +-0, or NaN when the hit pixel's rgb is not finite (forward.cu:824-826), identically for all of
// them.  So the table holds the rays with w != 0 first, in the reference's loop order (`n_live`), and the zero-weight
// ones behind them (`n_dead`; 32 of 512 at delta = 0.0625): SSAO marches the live ones only, SSR the live ones plus ONE
// of the dead ones (its +-0 / NaN applied once is the same as applied 32 times: x + 0 = x, NaN is sticky); the
// sample counts (SSAO's sum of weights, SSR's nrSamples = all rays) are unchanged.  gigs_options.gi_zero_rays = 1
// marches every dead ray as well, after the live ones -- same partial sums, same bits (tested): the proof that the
// cut is exact.  The dead rays are recognised by value (sin = 0 and ts.xy = 0, all with the same ts.z / cos), not by
// index; if a delta ever produced zero-weight rays of different directions they would simply count as live.
// The cache is keyed by (device, delta) and an entry is never modified once built, so concurrent streams and
// contexts share it safely.
// ------------------------------------------------------------------------------------------
struct RayTable {
  int device = -1;
  float delta = -1.0f;
  int nrays = 0;   // all rays of the reference's loops (SSR's nrSamples)
  int n_live = 0;  // rays with a non-zero weight: table entries [0, n_live)
  int n_dead = 0;  // identical zero-weight rays: entries [n_live, n_live + n_dead)
  float sum_w = 0.0f;  // SSAO's nrSamples, accumulated in fp32 in ray order
  float4* dev = nullptr;
};
static std::mutex g_ray_mu;
static std::vector<RayTable> g_ray;  // immutable entries

static int get_ray_table(float delta, hipStream_t s, RayTable& out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return -2;
  std::lock_guard<std::mutex> lk(g_ray_mu);
  for (const RayTable& t : g_ray)
    if (t.device == dev && t.delta == delta) { out = t; return 0; }
  if (!(delta > 1e-4f)) return -1;  // would not terminate / absurd table
  std::vector<float4> h;
  const float sampleDelta = delta * kPiF;
  float sum_w = 0.0f;
  for (float phi = 0.0f; (double)phi < 2.0 * (double)kPiF; phi += sampleDelta) {
    for (float theta = 0.0f; (double)theta <= 0.5 * (double)kPiF;
         theta = (float)((double)theta + (double)sampleDelta * 0.5)) {
      const float ct = cosf(theta), st = sinf(theta);
      float tx = st * cosf(phi), ty = st * sinf(phi), tz = ct;
      const float inv = 1.0f / sqrtf(tx * tx + ty * ty + tz * tz);
      tx *= inv; ty *= inv; tz *= inv;
      const float w = ct * st;
      sum_w += w;
      h.push_back(make_float4(tx, ty, tz, ct));
      h.push_back(make_float4(st, w, 0.0f, 0.0f));
      if (h.size() > (1u << 21)) return -1;
    }
  }
  // live rays first, the (identical) zero-weight rays behind them
  const size_t n = h.size() / 2;
  std::vector<size_t> dead;
  for (size_t r = 0; r < n; r++) {
    const float4 a = h[2 * r], b = h[2 * r + 1];
    if (b.x == 0.0f && b.y == 0.0f && a.x == 0.0f && a.y == 0.0f) dead.push_back(r);
  }
  for (size_t k = 1; k < dead.size(); k++)
    if (h[2 * dead[k]].z != h[2 * dead[0]].z || h[2 * dead[k]].w != h[2 * dead[0]].w) { dead.clear(); break; }
  std::vector<float4> ordered;
  ordered.reserve(h.size());
  {
    size_t k = 0;
    for (size_t r = 0; r < n; r++) {
      if (k < dead.size() && dead[k] == r) { k++; continue; }
      ordered.push_back(h[2 * r]); ordered.push_back(h[2 * r + 1]);
    }
    for (size_t r : dead) { ordered.push_back(h[2 * r]); ordered.push_back(h[2 * r + 1]); }
  }
  RayTable t;
  const size_t bytes = ordered.size() * sizeof(float4);
  // one-time (per device and delta) internal allocation; never freed, never rewritten
  if (hipMalloc(&t.dev, bytes ? bytes : 16) != hipSuccess) return -2;
  if (bytes) {
    // a blocking copy: the entry is complete before any stream can read it.  (First use of a delta on a device must
    // therefore happen outside a hipGraph capture -- every capture in this package is preceded by an eager warm-up.)
    if (hipMemcpy(t.dev, ordered.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return -2;
  }
  (void)s;
  t.device = dev;
  t.delta = delta;
  t.nrays = (int)n;
  t.n_dead = (int)dead.size();
  t.n_live = (int)(n - dead.size());
  t.sum_w = sum_w;
  g_ray.push_back(t);
  out = t;
  return 0;
}

struct GiParams {
  int W, H;
  float fx, fy, radius, bias, thick;
  int step, start;
  float inv_step;  // exact 1/step when step is a power of two
  int n_live;       // rays [0, n_live) are split across the four waves of a workgroup
  int nrays;        // rays [n_live, nrays) behind them go round-robin to the waves (zero-weight rays: none for SSAO, one for SSR)
  int nrays_total;  // all rays of the reference's loops (SSR's nrSamples)
  int tile_log2w;  // the 64 pixels of a workgroup form a (1 << tile_log2w) x (64 >> tile_log2w) rectangle
  int ray_interleave;  // fast marches: wave w takes ray pairs w, w + 4, ... instead of the w-th quarter of the ray set
  int cert_shift;      // certification blocks are (1 << cert_shift)^2 pixels; 0 = no certification table
  int cert_w, cert_h;  // blocks per row / column of the image; the table is (cert_w + 1) x (cert_h + 1): last column / row = border
  float cert_d0;       // certification needs den > cert_d0 (CertK::d0)
  // j / step for j = start .. start + 63 (exact for a power-of-two step): the fast path reads its per-step factor from
  // here with a scalar load instead of converting and multiplying on the vector ALU for every group (marches of more
  // than kFjTable - kGiGroup steps take the general path)
  float fjt[64];
  float fjc[64];  // fjt with the entries past the last step (step - 1 - start) repeating the last one
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// (x / d, y / d) with both quotients correctly rounded, i.e. bit-identical to two IEEE
// divisions.  hipcc expands an fp32 division into v_div_scale x2, v_rcp, a Newton step on the
// reciprocal, three residual FMAs, v_div_fmas and v_div_fixup.  When neither operand needs the
// exponent pre-scaling of v_div_scale (|values| within 2^-60 .. 2^60) the scale and fixup are
// identities and the quotient is exactly the FMA chain below; the reciprocal refinement is
// shared between the two numerators and the residual steps run as packed fp32 FMAs
// (v_pk_fma_f32): 1 rcp + 2 FMA + 5 packed ops instead of 22 instructions.
// (Checked bit-for-bit against `/` on the device: tests/test_gpu_parity.py::test_fast_div2.)
// `mag_ok` = the caller guarantees |n.x|, |n.y|, |d| < 2^60 (no overflow / pre-scaling); the
// denominator must also be away from zero.  A tiny or zero NUMERATOR is harmless here: the chain
// then returns the correctly signed tiny quotient or a zero of either sign, and the caller adds
// cx/cy (>= 0.5) to q * fx, which absorbs both.
__device__ __forceinline__ f32x2 div2_exact(f32x2 n, float d, bool mag_ok) {
  if (mag_ok && fabsf(d) > 0x1p-60f) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const f32x2 nd = {-d, -d}, rr = {r1, r1};
    const f32x2 q0 = n * rr;
    const f32x2 e1 = __builtin_elementwise_fma(nd, q0, n);
    const f32x2 q1 = __builtin_elementwise_fma(e1, rr, q0);
    const f32x2 e2 = __builtin_elementwise_fma(nd, q1, n);
    return __builtin_elementwise_fma(e2, rr, q1);
  }
  return f32x2{n.x / d, n.y / d};
}

// The FMA chain alone (valid under the conditions stated above; garbage otherwise).
__device__ __forceinline__ f32x2 div2_fast(f32x2 n, float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e0 = __builtin_fmaf(-d, r0, 1.0f);
  const float r1 = __builtin_fmaf(e0, r0, r0);
  const f32x2 nd = {-d, -d}, rr = {r1, r1};
  const f32x2 q0 = n * rr;
  const f32x2 e1 = __builtin_elementwise_fma(nd, q0, n);
  const f32x2 q1 = __builtin_elementwise_fma(e1, rr, q0);
  const f32x2 e2 = __builtin_elementwise_fma(nd, q1, n);
  return __builtin_elementwise_fma(e2, rr, q1);
}

// (int)roundf(t) for the GPU's saturating conversion: exhaustively verified over all 2^32 floats
// that trunc(t + copysign(0.5 - 2^-25, t)) rounds half away from zero like roundf.
__device__ __forceinline__ int round_to_int(float t) { return f2i(t + copysignf(0.49999997f, t)); }

// The march only asks "which pixel, if inside the image".  For that, floor(t + (0.5 - 2^-25)) serves as well as
// (int)roundf(t): the two agree for every t >= -0.5 + 2^-25 (above -0.5 the sum is >= 0 and floor = trunc, which is
// the identity above), at t <= -0.5 both are negative, and a NaN converts to 0 either way; beyond +-2^31 both
// saturate.  So they select the same pixel whenever one of them is a valid coordinate of an image side < 2^15, and
// are both outside otherwise.  v_floor_f32 (VOP1) replaces v_bfi_b32 (VOP3: ~1.5 cycles more per coordinate).
// Checked over ALL 2^32 floats on the device: gigs_selftest_round.
__device__ __forceinline__ int round_pix(float t) { return f2i(floorf(t + 0.49999997f)); }
// the same for a coordinate pair whose +(0.5 - 2^-25) is done as one packed add
__device__ __forceinline__ int floor_pix(float t_plus_half) { return f2i(floorf(t_plus_half)); }

// Buffer descriptor of one fp32 image plane with a 4-byte stride: an `idxen` load takes the pixel index itself (the
// hardware scales it) and range-checks it against the pixel count, an `offen` load with a byte offset works too.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t z_plane_rsrc(const float* plane, size_t HW) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(plane), /*stride*/ 4, (int)HW, 0x00020000);
}

// N indexed gathers in flight, then one wait.  Inline assembly because clang has no builtin for the indexed (struct)
// buffer load; the destinations are not tracked by the compiler's wait-count insertion, hence the explicit s_waitcnt
// that every result passes through before it is used.
__device__ __forceinline__ float gather_idx_issue(unsigned i, __amdgpu_buffer_rsrc_t rsrc) {
  float z;
  asm volatile("buffer_load_dword %0, %1, %2, 0 idxen" : "=&v"(z) : "v"(i), "s"(rsrc));
  return z;
}
template <int N>
__device__ __forceinline__ void gather_idx(float* z, const unsigned* i, __amdgpu_buffer_rsrc_t rsrc) {
  static_assert(N == 4 || N == 8 || N == 16, "group sizes of the march");
#pragma unroll
  for (int k = 0; k < N; k++) z[k] = gather_idx_issue(i[k], rsrc);
  if constexpr (N == 16) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]));
    asm volatile("" : "+v"(z[8]), "+v"(z[9]), "+v"(z[10]), "+v"(z[11]), "+v"(z[12]), "+v"(z[13]), "+v"(z[14]), "+v"(z[15]));
  } else if constexpr (N == 8)
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]));
  else
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]));
}

// Sample coordinates of one group of steps for kRays rays: byte offset into the z plane (0 when the sample
// is outside the image), the in-image flag and the sample's z.  kExact = false uses the shared-reciprocal
// FMA chain for every lane and returns the smallest |den| seen, so that the caller can decide ONCE per
// group whether any lane needed the IEEE slow path; kExact = true re-evaluates with per-lane selection.
template <bool kPow2, int kGroup, int kRays, bool kExact>
__device__ __forceinline__ float group_coords(const GiParams& p, v3 pos, float a, const v3* sv, float cx, float cy,
                                              bool mag_ok, int j0, unsigned (*off)[kGroup], bool (*inb)[kGroup],
                                              float (*spzv)[kGroup]) {
  const f32x2 posxy = {pos.x, pos.y}, fxy = {p.fx, p.fy}, cxy = {cx, cy};
  float min_den = __builtin_inff();
#pragma unroll
  for (int g = 0; g < kGroup; g++) {
    // pos + sampleVec * j * (1 + z/100) * (1 + z/100) * radius / step, left to right (forward.cu:693).  For a
    // power-of-two step the final division is an exact scaling, and an exact scaling commutes with every
    // rounding of the chain as long as no intermediate falls into the subnormal range: the fast path folds it
    // into j (j / step is exact) and saves three multiplies per sample; `sv_scale_ok` (checked per ray by the
    // caller) rules the subnormal case out, otherwise the exact path below runs.
    const float fj = (kPow2 && !kExact) ? p.fjt[j0 - p.start + g] : (float)(j0 + g);
    const bool in_range = (j0 + g) < p.step;
#pragma unroll
    for (int k = 0; k < kRays; k++) {
      f32x2 m = f32x2{sv[k].x, sv[k].y} * fj;
      float mz = sv[k].z * fj;
      m = m * a; mz = mz * a;
      m = m * a; mz = mz * a;
      m = m * p.radius; mz = mz * p.radius;
      if (kPow2) {
        if (kExact) { m = m * p.inv_step; mz = mz * p.inv_step; }
      } else {
        const float fs = (float)p.step;
        m = f32x2{m.x / fs, m.y / fs}; mz = mz / fs;
      }
      const f32x2 sp = posxy + m;
      const float spz = pos.z + mz;
      // get_coord (ssr.h:120-135)
      const float den = spz + 0.0000001f;
      f32x2 qv = div2_fast(sp, den);
      if (kExact) {
        if (!(mag_ok && fabsf(den) > 0x1p-60f)) qv = f32x2{sp.x / den, sp.y / den};
      } else {
        min_den = fminf(min_den, fabsf(den));  // a NaN den is not recorded: both paths then give NaN -> pixel (0, 0)
      }
      const f32x2 t = (qv * fxy + cxy) + 0.49999997f;  // round_pix's addend, packed
      const int ix = floor_pix(t.x);
      const int iy = floor_pix(t.y);
      inb[k][g] = in_range && (unsigned)ix < (unsigned)p.W && (unsigned)iy < (unsigned)p.H;
      // The gather goes through a buffer descriptor of exactly the z plane: an out-of-image sample (never
      // used: inb is false) yields some wrapped offset that the hardware range check either reads harmlessly
      // or answers with 0 -- no clamp, no predication.  W, H < 2^15 (checked by the C-ABI wrapper), so the
      // in-image pixel index is exact in the 24-bit multiply.
      off[k][g] = __umul24((unsigned)iy, (unsigned)p.W) + (unsigned)ix;
      spzv[k][g] = spz;
    }
  }
  return min_den;
}

// The same for exactly two rays, with every z-lane scalar of the pair (sample z, denominator, reciprocal
// refinement, the two depth-test bounds) held as one fp32 pair: the kernel is bound by VALU issue and a
// v_pk_* instruction (~4.7 cycles for two elements) is cheaper than two scalar ones that carry an SGPR / literal
// operand or a VOP3 encoding (~4 cycles each; only VGPR-only VOP2 forms reach ~2.5: tools/microbench/pk_forms.hip).  Same IEEE operations per element, same results.  hi2 / lo2 receive spz + bias and
// spz - thick of the two rays (ssr.h hit test operands).
template <bool kPow2, int kGroup, bool kExact>
__device__ __forceinline__ float group_coords2(const GiParams& p, v3 pos, float a, const v3* sv, float cx, float cy,
                                               bool mag_ok, int j0, unsigned (*off)[kGroup], bool (*inb)[kGroup],
                                               f32x2* hi2, f32x2* lo2) {
  const f32x2 posxy = {pos.x, pos.y}, fxy = {p.fx, p.fy}, cxy = {cx, cy};
  const f32x2 posz2 = {pos.z, pos.z}, svz = {sv[0].z, sv[1].z};
  const f32x2 sxy[2] = {f32x2{sv[0].x, sv[0].y}, f32x2{sv[1].x, sv[1].y}};
  float min_den = __builtin_inff();
#pragma unroll
  for (int g = 0; g < kGroup; g++) {
    const float fj = (kPow2 && !kExact) ? p.fjt[j0 - p.start + g] : (float)(j0 + g);  // see group_coords
    const bool in_range = (j0 + g) < p.step;
    f32x2 m[2] = {sxy[0] * fj, sxy[1] * fj};
    f32x2 mz = svz * fj;
    m[0] = m[0] * a; m[1] = m[1] * a; mz = mz * a;
    m[0] = m[0] * a; m[1] = m[1] * a; mz = mz * a;
    m[0] = m[0] * p.radius; m[1] = m[1] * p.radius; mz = mz * p.radius;
    if (kPow2) {
      if (kExact) { m[0] = m[0] * p.inv_step; m[1] = m[1] * p.inv_step; mz = mz * p.inv_step; }
    } else {
      const float fs = (float)p.step;
      m[0] = f32x2{m[0].x / fs, m[0].y / fs}; m[1] = f32x2{m[1].x / fs, m[1].y / fs}; mz = f32x2{mz.x / fs, mz.y / fs};
    }
    const f32x2 spz = posz2 + mz;
    const f32x2 den = spz + 0.0000001f;  // get_coord (ssr.h:120-135)
    // shared-reciprocal division (div2_fast) with the reciprocal refinement of both rays packed
    const f32x2 r0 = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const f32x2 e0 = __builtin_elementwise_fma(-den, r0, f32x2{1.0f, 1.0f});
    const f32x2 r1 = __builtin_elementwise_fma(e0, r0, r0);
    if (!kExact) min_den = fminf(fminf(fabsf(den.x), fabsf(den.y)), min_den);  // NaN dens are not recorded (see group_coords)
    hi2[g] = spz + p.bias;
    lo2[g] = spz - p.thick;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const f32x2 sp = posxy + m[k];
      const float d = k == 0 ? den.x : den.y, r = k == 0 ? r1.x : r1.y;
      const f32x2 nd = {-d, -d}, rr = {r, r};
      const f32x2 q0 = sp * rr;
      const f32x2 e1 = __builtin_elementwise_fma(nd, q0, sp);
      const f32x2 q1 = __builtin_elementwise_fma(e1, rr, q0);
      const f32x2 e2 = __builtin_elementwise_fma(nd, q1, sp);
      f32x2 qv = __builtin_elementwise_fma(e2, rr, q1);
      if (kExact) {
        if (!(mag_ok && fabsf(d) > 0x1p-60f)) qv = f32x2{sp.x / d, sp.y / d};
      }
      const f32x2 t = (qv * fxy + cxy) + 0.49999997f;  // round_pix's addend, packed
      const int ix = floor_pix(t.x);
      const int iy = floor_pix(t.y);
      inb[k][g] = in_range && (unsigned)ix < (unsigned)p.W && (unsigned)iy < (unsigned)p.H;
      off[k][g] = __umul24((unsigned)iy, (unsigned)p.W) + (unsigned)ix;  // see group_coords
    }
  }
  return min_den;
}

// Marches kRays rays of one pixel together; hit[k] = pixel index of ray k's hit or -1.
// The reference walks j = start..step-1 per ray and stops at the first sample that leaves the image
// or hits (forward.cu:691-714).  ~98 % of the rays of a real frame run all their steps, so:
//   * steps are evaluated in groups of kGroup: the coordinates of the whole group (times kRays) are
//     computed and their z-plane gathers issued together -- memory-level parallelism instead of one
//     dependent L2 round trip per step;
//   * the group is then resolved IN ORDER with per-lane selects, not branches (`open` = ray still
//     marching), which is exactly the sequential outcome; the only branches left are wave-uniform
//     (all rays of the wave resolved; the rare IEEE-division re-march, decided once per ray pair), so the
//     scalar unit is not spent on exec-mask bookkeeping.
template <bool kPow2, int kGroup, int kRays, bool kExact>
__device__ __forceinline__ bool march_impl(const GiParams& p, v3 pos, float a, const v3* sv, float cx, float cy,
                                           __amdgpu_buffer_rsrc_t pos_z, bool mag_ok, int* hit) {
  bool open[kRays];
#pragma unroll
  for (int k = 0; k < kRays; k++) { open[k] = true; hit[k] = -1; }
  float min_den = __builtin_inff();
  for (int j0 = p.start; j0 < p.step; j0 += kGroup) {
    unsigned off[kRays][kGroup];
    float zv[kRays][kGroup];
    bool inb[kRays][kGroup];
    f32x2 hi2[kGroup], lo2[kGroup];           // kRays == 2: depth-test bounds of the pair
    float spzv[kRays == 2 ? 1 : kRays][kGroup];  // otherwise: the sample z
    if constexpr (kRays == 2)
      min_den = fminf(min_den, group_coords2<kPow2, kGroup, kExact>(p, pos, a, sv, cx, cy, mag_ok, j0, off, inb, hi2, lo2));
    else
      min_den = fminf(min_den, group_coords<kPow2, kGroup, kRays, kExact>(p, pos, a, sv, cx, cy, mag_ok, j0, off, inb, spzv));
    {
      constexpr int kN = kRays * kGroup;
      unsigned in[kN];
      float zn[kN];
#pragma unroll
      for (int g = 0; g < kGroup; g++)
#pragma unroll
        for (int k = 0; k < kRays; k++) in[g * kRays + k] = off[k][g];
      gather_idx<kN>(zn, in, pos_z);
#pragma unroll
      for (int g = 0; g < kGroup; g++)
#pragma unroll
        for (int k = 0; k < kRays; k++) zv[k][g] = zn[g * kRays + k];
    }
    bool any_open = false;
#pragma unroll
    for (int k = 0; k < kRays; k++) {
#pragma unroll
      for (int g = 0; g < kGroup; g++) {
        float hi, lo;
        if constexpr (kRays == 2) {
          hi = k == 0 ? hi2[g].x : hi2[g].y;
          lo = k == 0 ? lo2[g].x : lo2[g].y;
        } else {
          hi = spzv[k][g] + p.bias;
          lo = spzv[k][g] - p.thick;
        }
        const bool h = inb[k][g] && (zv[k][g] <= hi) && (zv[k][g] >= lo);
        hit[k] = (open[k] && h) ? (int)off[k][g] : hit[k];
        open[k] = open[k] && inb[k][g] && !h;
      }
      any_open = any_open || open[k];
    }
    if (!__any(any_open)) break;
  }
  return !(mag_ok && min_den > 0x1p-60f);  // this lane needed an IEEE division somewhere
}

// The fast pass assumes no sample of the pixel needs the operand pre-scaling of an IEEE division and
// reports per lane whether that held; if any lane of the wave says no (never, on real frames) the
// rays are marched again with the per-lane exact selection -- two separate code paths, so the common
// one carries neither the branches nor the registers of the rare one.
template <bool kPow2, int kGroup, int kRays>
__device__ __forceinline__ void march(const GiParams& p, v3 pos, float a, const v3* sv, float cx, float cy,
                                      __amdgpu_buffer_rsrc_t pos_z, bool mag_ok, unsigned sv_min_bits, int* hit) {
  bool bad = march_impl<kPow2, kGroup, kRays, false>(p, pos, a, sv, cx, cy, pos_z, mag_ok, hit);
  if (kPow2) {
    // folding 1/step into j is exact unless a product of the chain is subnormal: every non-zero component of the
    // sample vectors must stay above sv_min (= 2^-120 / the smallest cumulative factor of the chain; a zero
    // component gives exact zeros either way).  As unsigned integers: (|bits| - 1) >= (sv_min_bits - 1).
    unsigned lowest = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < kRays; k++)
      lowest = min(lowest, min(min((__float_as_uint(sv[k].x) & 0x7fffffffu) - 1u, (__float_as_uint(sv[k].y) & 0x7fffffffu) - 1u),
                               (__float_as_uint(sv[k].z) & 0x7fffffffu) - 1u));
    bad = bad || lowest < sv_min_bits - 1u;
  }
  if (__builtin_expect(__any(bad), 0)) march_impl<kPow2, kGroup, kRays, true>(p, pos, a, sv, cx, cy, pos_z, mag_ok, hit);
}

struct Tbn { v3 t, b, n; };
__device__ __forceinline__ Tbn make_tbn(v3 normal_un) {
  Tbn r;
  r.n = normalize3(normal_un);
  const v3 up = {0.0f, 1.0f, 0.0f};
  const float rndot = dot3(up, r.n);
  const v3 untangent = {up.x - r.n.x * rndot, up.y - r.n.y * rndot, up.z - r.n.z * rndot};
  r.t = normalize3(untangent);
  r.b = normalize3(cross3(r.n, r.t));
  return r;
}
// transformVec3x3(ts, TBN) with TBN rows (t, b, n): auxiliary.h:90-98
__device__ __forceinline__ v3 tbn_apply(const Tbn& m, float x, float y, float z) {
  return {m.t.x * x + m.b.x * y + m.n.x * z, m.t.y * x + m.b.y * y + m.n.y * z,
          m.t.z * x + m.b.z * y + m.n.z * z};
}
// |sample position| <= |pos| + |sv| * (j/step) * a^2 * radius with |sv| <= sqrt(3): when this bound
// is far below 2^60 for the pixel, no step of any ray needs the operand pre-scaling of IEEE division.
__device__ __forceinline__ bool gi_mag_ok(v3 pos, float a, float radius) {
  const float bound = fmaxf(fmaxf(fabsf(pos.x), fabsf(pos.y)), fabsf(pos.z)) + 2.0f * fabsf(a * a * radius) + 1.0f;
  return bound < 0x1p59f;  // false for NaN / inf
}

// Smallest non-zero |sample-vector component| for which folding 1/step into j is exact (see group_coords): the
// chain multiplies by j/step >= 1/step, a, a, radius; its smallest cumulative factor must keep the product
// above 2^-120.  Returned as fp32 bits (positive floats order like unsigned integers); NaN / zero factors give
// +inf bits, i.e. "never exact" -> the exact path.
__device__ __forceinline__ unsigned gi_sv_min_bits(float a, float radius, float inv_step) {
  const float aa = fabsf(a), a2 = aa * aa;
  const float lo = fminf(fminf(1.0f, aa), fminf(a2, a2 * fabsf(radius))) * inv_step;
  const float sv_min = 0x1p-120f / lo;  // lo == 0 or NaN -> inf / NaN
  return (sv_min == sv_min) ? __float_as_uint(sv_min) : 0x7f800000u;
}

// Every sample direction contains t * ts.x, so a tangent that is NaN in all three components
// (empty pixel: N = 0 -> NaN normal; or N parallel to `up`) makes every sample position NaN:
// the projected pixel is (0, 0), the depth test against NaN is false, no ray ever hits.  Such
// pixels -- the whole background of an object-centric frame -- skip the march with the same result.
__device__ __forceinline__ bool tbn_never_hits(const Tbn& m) {
  return (m.t.x != m.t.x) && (m.t.y != m.t.y) && (m.t.z != m.t.z);
}


// ------------------------------------------------------------------------------------------
// Tolerance-spending marches (GIGS_GI_MARCH; kMode > 0)
// ------------------------------------------------------------------------------------------
// The exact march above reproduces the oracle's pixel choice bit for bit and pays for it with ~31 VALU
// instructions per ray-step.  north_star's bar for the fp planes is 1e-4 mean per-pixel L1, and the reference's
// own binary is an FMA-contracted compilation of the same lines (nvcc -fmad=true), so its projected coordinates
// already differ from the contraction-free oracle in the last place.  The variants below spend that tolerance:
//
//   kMode 1 "hoist"      per-ray step vector k = ((sv*a)*a*radius) (then * j/step), sp = pos + (j/step)*k with a
//                        separate multiply and add; both quotients IEEE-correct (the shared-reciprocal chain);
//                        q*f + c as multiply and add.  Only the order of the multiplies differs from forward.cu:693.
//   kMode 2 "hoist_fma"  the same with FMAs for pos + fj*k and q*f + c (what nvcc's contraction does to ssr.h:133).
//   kMode 3 "proj_nr"    projective form: numerators (pos.xy*f + fj*Bxy), denominator (pos.z + 1e-7 + fj*Bz), one
//                        v_rcp_f32 refined by one Newton step, t = n*r + (c + 0.5): 3 packed FMAs + rcp per sample.
//   kMode 4 "proj"       kMode 3 with the raw 1-ulp v_rcp_f32.
//
// In modes 3/4 the depth test z in [spz - thick, spz + bias] is evaluated as |z - (den + cm)| <= hh with
// cm = (bias - thick)/2 - 1e-7, hh = (bias + thick)/2 (same interval, different rounding of its ends).
// Every mode keeps the march's control flow: in-order resolution, break on leaving the image, first hit wins.
// tools/gi_variants.py measures each mode against mode 0 (= the oracle, bit for bit) on the C2 view and the GI test
// scenes; DESIGN.md section 5 holds the table and the choice of default.
struct FastPix {
  f32x2 Axy;      // modes 1/2: pos.xy;            modes 3/4: pos.xy * (fx, fy)
  float Dz;       // modes 1/2: pos.z;             modes 3/4: pos.z + 1e-7
  f32x2 cxy;      // (cx, cy) + round_pix's addend
  f32x2 c0xy;     // (cx, cy)
  f32x2 fxy;
  float cm, hh, bias, thick;
};

__device__ __forceinline__ int cvt_flr(float t) {  // (int)floor(t), saturating, NaN -> 0: one VOP1 instruction
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(t));
  return r;
}

template <int kMode, int kGroup>
__device__ __forceinline__ void march2_fast(const GiParams& p, const FastPix& c, const f32x2* Bxy, f32x2 Bz2,
                                            __amdgpu_buffer_rsrc_t pos_z, int* hit) {
  bool open[2] = {true, true};
  hit[0] = hit[1] = -1;
  const f32x2 Dz2 = {c.Dz, c.Dz};
  for (int j0 = p.start; j0 < p.step; j0 += kGroup) {
    unsigned off[2][kGroup];
    bool inb[2][kGroup];
    f32x2 ta[kGroup], tb[kGroup];  // modes 1/2: spz + bias, spz - thick; modes 3/4: mid (ta only)
#pragma unroll
    for (int g = 0; g < kGroup; g++) {
      const float fj = p.fjt[j0 - p.start + g];  // j / step
      const f32x2 fj2 = {fj, fj};
      const bool in_range = (j0 + g) < p.step;
      f32x2 r;
      f32x2 den;
      if constexpr (kMode >= 3) {
        den = __builtin_elementwise_fma(Bz2, fj2, Dz2);
        ta[g] = den + c.cm;
        r = f32x2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
        if constexpr (kMode == 3) {
          const f32x2 e0 = __builtin_elementwise_fma(-den, r, f32x2{1.0f, 1.0f});
          r = __builtin_elementwise_fma(e0, r, r);
        }
      } else {
        const f32x2 spz = (kMode == 2) ? __builtin_elementwise_fma(Bz2, fj2, Dz2) : Dz2 + Bz2 * fj2;
        den = spz + 0.0000001f;
        ta[g] = spz + c.bias;
        tb[g] = spz - c.thick;
        const f32x2 r0 = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
        const f32x2 e0 = __builtin_elementwise_fma(-den, r0, f32x2{1.0f, 1.0f});
        r = __builtin_elementwise_fma(e0, r0, r0);
      }
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const float rk = k == 0 ? r.x : r.y;
        const f32x2 rr = {rk, rk};
        f32x2 t;
        if constexpr (kMode >= 3) {
          const f32x2 n = __builtin_elementwise_fma(Bxy[k], fj2, c.Axy);
          t = __builtin_elementwise_fma(n, rr, c.cxy);
        } else {
          const f32x2 sp = (kMode == 2) ? __builtin_elementwise_fma(Bxy[k], fj2, c.Axy) : c.Axy + Bxy[k] * fj2;
          const float d = k == 0 ? den.x : den.y;
          const f32x2 nd = {-d, -d};
          const f32x2 q0 = sp * rr;
          const f32x2 e1 = __builtin_elementwise_fma(nd, q0, sp);
          const f32x2 q1 = __builtin_elementwise_fma(e1, rr, q0);
          const f32x2 e2 = __builtin_elementwise_fma(nd, q1, sp);
          const f32x2 qv = __builtin_elementwise_fma(e2, rr, q1);
          t = ((kMode == 2) ? __builtin_elementwise_fma(qv, c.fxy, c.c0xy) : qv * c.fxy + c.c0xy) + 0.49999997f;
        }
        const int ix = cvt_flr(t.x);
        const int iy = cvt_flr(t.y);
        inb[k][g] = in_range && (unsigned)ix < (unsigned)p.W && (unsigned)iy < (unsigned)p.H;
        off[k][g] = __umul24((unsigned)iy, (unsigned)p.W) + (unsigned)ix;
      }
    }
    float zn[2 * kGroup];
    {
      unsigned in[2 * kGroup];
#pragma unroll
      for (int g = 0; g < kGroup; g++) { in[2 * g] = off[0][g]; in[2 * g + 1] = off[1][g]; }
      gather_idx<2 * kGroup>(zn, in, pos_z);
    }
    bool any_open = false;
#pragma unroll
    for (int k = 0; k < 2; k++) {
#pragma unroll
      for (int g = 0; g < kGroup; g++) {
        const float z = zn[2 * g + k];
        bool h;
        if constexpr (kMode >= 3) {
          h = inb[k][g] && fabsf(z - (k == 0 ? ta[g].x : ta[g].y)) <= c.hh;
        } else {
          h = inb[k][g] && (z <= (k == 0 ? ta[g].x : ta[g].y)) && (z >= (k == 0 ? tb[g].x : tb[g].y));
        }
        hit[k] = (open[k] && h) ? (int)off[k][g] : hit[k];
        open[k] = open[k] && inb[k][g] && !h;
      }
      any_open = any_open || open[k];
    }
    if (!__any(any_open)) break;
  }
}

// Conservative certification in front of the projective march.  The z plane's {non-zero minimum, max(maximum, 0)} over
// square blocks of 2^cert_shift pixels (gi_minmax_kernel) sit in LDS, with a border of "never certify" entries around
// the image and on blocks the image only partly covers.  A sample is projected in BLOCK units first (the per-pixel
// constants are pre-scaled by the exact factor 2^-shift); if its hit interval, widened by 2^-16 relative + 1e-6, lies
// entirely above the block's maximum, or entirely below its non-zero minimum and above zero (empty pixels hold 0), then
// whichever pixel of that -- fully in-image -- block the sample selects, the reference's test fails by a margin far above
// its rounding error: the sample neither hits nor leaves the image, the ray simply stays open, and nothing else has to
// be computed for it.  Only when some lane of the wave is NOT certified for a sample does the wave run the exact part
// (pixel coordinates = block coordinates * 2^shift exactly, bounds test, gather, depth test).  ~92 % of the wave-samples
// of the bench view skip it (tools/gi_cert_rate.py); every ray's outcome, hence every output bit, is unchanged
// (tests/test_gpu_parity.py::test_gi_certification_is_exact).
struct CertPix {
  f32x2 Axy, cxy;  // FastPix::Axy / cxy scaled by 2^-shift
  float scale;     // 2^shift
  int bw, bh;      // blocks per row / column of the image (the table has one more column / row: the never-certify border)
};

typedef unsigned long long u64;

// certification thresholds on the sample's denominator den = spz + 1e-7 (the hit interval is [den + cm - hh, den + cm + hh]):
//   lo = (den + cm) * (1 - 2^-16) - (hh + 1e-6) > block maximum      <=>  den > t_hi   (stored per block)
//   hi = (den + cm) * (1 + 2^-16) + (hh + 1e-6) < block non-zero min <=>  den < t_lo   (stored per block)
//   lo > 0 (empty pixels hold z = 0)                                  <=>  den > d0     (a launch constant)
// each rounded away from certification (gi_minmax_kernel).
struct CertK { float lo_k, hi_k, d0; };
__host__ __device__ __forceinline__ CertK cert_consts(float bias, float thick) {
  const float cm = 0.5f * (bias - thick) - 0.0000001f, hh = 0.5f * (bias + thick);
  CertK k;
  k.lo_k = cm * (1.0f - 0x1p-16f) - hh - 1e-6f;
  k.hi_k = cm * (1.0f + 0x1p-16f) + hh + 1e-6f;
  const float d = -k.lo_k / (1.0f - 0x1p-16f);
  k.d0 = d + fabsf(d) * 0x1p-20f + 1e-30f;
  return k;
}

// kTabOff: LDS byte address of the table (the kernels' dynamic LDS starts right after their static arrays; checked at
// kernel entry).  The lookups are written as instructions because the compiler adds the -- constant -- base with a
// v_add per lookup instead of using the ds_read offset field.
template <int kMode, int kGroup, int kTabOff>
__device__ __forceinline__ void march2_cert(const GiParams& p, const FastPix& c, const CertPix& cp, const f32x2* Bxy16,
                                            f32x2 Bz2, __amdgpu_buffer_rsrc_t pos_z, int* hit) {
  static_assert(kMode >= 3, "certification is wired into the projective marches");
  // Lane predicates are kept as 64-bit wave masks in SGPRs (ballots of single compares, combined with scalar logic,
  // turned back into a lane predicate only where a select needs one): written with per-lane bools the compiler moves
  // them through VGPRs (v_cndmask 0/1 + v_cmp_ne per use) -- a third of the march's vector instructions.
  u64 open_m[2];
  open_m[0] = open_m[1] = __builtin_amdgcn_ballot_w64(true);  // the lanes that march (EXEC)
  hit[0] = hit[1] = -1;
  const f32x2 Dz2 = {c.Dz, c.Dz};
  const unsigned row8 = (unsigned)(cp.bw + 1) * 8u;
  // table entry of block (bx, by) at byte by * row8 + bx * 8; column bw / row bh never certify
  // block indices as unsigned: a negative one is a huge one, and both clamp to bw / bh (one v_min_u32 per coordinate)
  unsigned bw_v = (unsigned)cp.bw, bh_v = (unsigned)cp.bh;
  asm("" : "+v"(bw_v), "+v"(bh_v));  // in VGPRs: the VOP2 form with two VGPR operands issues faster than an SGPR-source one
  // den > d0 for every sample of a ray <=> at both of its ends (den is a correctly rounded, hence monotone, function of j):
  // decided once per ray instead of once per sample
  u64 pos[2];
  {
    const float f0 = p.fjt[0], f1 = p.fjt[p.step - 1 - p.start];
    const f32x2 da = __builtin_elementwise_fma(Bz2, f32x2{f0, f0}, Dz2), db = __builtin_elementwise_fma(Bz2, f32x2{f1, f1}, Dz2);
    pos[0] = __builtin_amdgcn_ballot_w64(da.x > p.cert_d0) & __builtin_amdgcn_ballot_w64(db.x > p.cert_d0);
    pos[1] = __builtin_amdgcn_ballot_w64(da.y > p.cert_d0) & __builtin_amdgcn_ballot_w64(db.y > p.cert_d0);
  }
  for (int j0 = p.start; j0 < p.step; j0 += kGroup) {
    f32x2 tb[2][kGroup];  // sample position in block units (+ the rounding addend, scaled)
    f32x2 dn[kGroup];
    f32x2 th[2][kGroup];  // {t_lo, t_hi} of the sample's block
    u64 cert[2][kGroup];
    // the group's j / step factors, in SGPRs before the first lookup is issued: a scalar load between the lookups would
    // bring a wait on the counter they share, i.e. on the lookups
    float fjs[kGroup];
#pragma unroll
    for (int g = 0; g < kGroup; g++) fjs[g] = p.fjc[j0 - p.start + g];
    asm volatile("" : "+s"(fjs[0]), "+s"(fjs[1]), "+s"(fjs[2]), "+s"(fjs[3]));
    // phase A: every sample of the group up to its table lookup; no branches
#pragma unroll
    for (int g = 0; g < kGroup; g++) {
      const float fj = fjs[g];  // j / step
      const f32x2 fj2 = {fj, fj};
      const f32x2 den = __builtin_elementwise_fma(Bz2, fj2, Dz2);
      dn[g] = den;
      f32x2 r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
      if constexpr (kMode == 3) {
        const f32x2 e0 = __builtin_elementwise_fma(-den, r, f32x2{1.0f, 1.0f});
        r = __builtin_elementwise_fma(e0, r, r);
      }
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const float rk = k == 0 ? r.x : r.y;
        const f32x2 n = __builtin_elementwise_fma(Bxy16[k], fj2, cp.Axy);
        tb[k][g] = __builtin_elementwise_fma(n, f32x2{rk, rk}, cp.cxy);
        const unsigned bx = min((unsigned)cvt_flr(tb[k][g].x), bw_v);
        const unsigned by = min((unsigned)cvt_flr(tb[k][g].y), bh_v);
        const unsigned toff = __umul24(by, row8) + (bx << 3);
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(th[k][g]) : "v"(toff), "n"(kTabOff));
      }
    }
    // the lookups above are invisible to the compiler's wait-count bookkeeping: wait for them here (operands tie the order)
    static_assert(kGroup == 4, "the wait below lists the lookups of a group of four");
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(th[0][0]), "+v"(th[0][1]), "+v"(th[0][2]), "+v"(th[0][3]), "+v"(th[1][0]), "+v"(th[1][1]), "+v"(th[1][2]),
                   "+v"(th[1][3]));
#pragma unroll
    for (int g = 0; g < kGroup; g++) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const float dk = k == 0 ? dn[g].x : dn[g].y;
        const u64 above = __builtin_amdgcn_ballot_w64(dk > th[k][g].y);
        const u64 below = __builtin_amdgcn_ballot_w64(dk < th[k][g].x);
        cert[k][g] = above | (below & pos[k]);
      }
    }
    // phase B: the exact part for the samples some lane of the wave still needs (~8 % of them).  A group that runs past the
    // last step repeats the last sample (fjc): repeating a sample changes nothing -- a ray it closed is closed, one it left
    // open it leaves open again -- so there is no per-sample range test.  Two stages, so that a group with several such
    // samples pays ONE round trip to the z plane instead of one per sample: every lookup that MAY be needed (judged with the
    // rays' state at the start of the group: a ray can only close) is requested, then the samples are resolved in order --
    // each ray's own (the two rays are independent) -- with the state as it evolves.
    u64 maybe[2][kGroup], any = 0ull;
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
      for (int g = 0; g < kGroup; g++) { maybe[k][g] = open_m[k] & ~cert[k][g]; any |= maybe[k][g]; }
    if (any != 0ull) {
      float z[2][kGroup];
      unsigned off[2][kGroup];
      u64 inb[2][kGroup];
#pragma unroll
      for (int k = 0; k < 2; k++)
#pragma unroll
        for (int g = 0; g < kGroup; g++) {
          z[k][g] = 0.0f; off[k][g] = 0u; inb[k][g] = 0ull;
          if (maybe[k][g] != 0ull) {
            const f32x2 t = tb[k][g] * cp.scale;  // exact: the pixel coordinates the uncertified march computes
            const int ix = cvt_flr(t.x);
            const int iy = cvt_flr(t.y);
            inb[k][g] = __builtin_amdgcn_ballot_w64((unsigned)ix < (unsigned)p.W) & __builtin_amdgcn_ballot_w64((unsigned)iy < (unsigned)p.H);
            off[k][g] = __umul24((unsigned)iy, (unsigned)p.W) + (unsigned)ix;
            z[k][g] = gather_idx_issue(__builtin_amdgcn_inverse_ballot_w64(maybe[k][g] & inb[k][g]) ? off[k][g] : 0xffffffffu, pos_z);
          }
        }
      static_assert(kGroup == 4, "the wait below lists the lookups of a group of four");
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(z[0][0]), "+v"(z[0][1]), "+v"(z[0][2]), "+v"(z[0][3]), "+v"(z[1][0]), "+v"(z[1][1]), "+v"(z[1][2]), "+v"(z[1][3]));
#pragma unroll
      for (int k = 0; k < 2; k++)
#pragma unroll
        for (int g = 0; g < kGroup; g++) {
          const u64 need = open_m[k] & maybe[k][g];  // the ray may have closed at an earlier sample of this group
          if (need != 0ull) {
            const u64 look = need & inb[k][g];
            const float mid = (k == 0 ? dn[g].x : dn[g].y) + c.cm;
            const u64 h = look & __builtin_amdgcn_ballot_w64(fabsf(z[k][g] - mid) <= c.hh);
            hit[k] = __builtin_amdgcn_inverse_ballot_w64(h) ? (int)off[k][g] : hit[k];
            open_m[k] &= ~((need & ~inb[k][g]) | h);  // left the image, or hit: the ray is closed
          }
        }
    }
    if ((open_m[0] | open_m[1]) == 0ull) break;
  }
}

// per-pixel constants of the fast marches and the per-ray vectors
template <int kMode>
__device__ __forceinline__ FastPix make_fast(const GiParams& p, v3 pos, float cx, float cy) {
  FastPix c;
  c.fxy = f32x2{p.fx, p.fy};
  c.cxy = f32x2{cx + 0.49999997f, cy + 0.49999997f};
  c.c0xy = f32x2{cx, cy};
  c.bias = p.bias; c.thick = p.thick;
  c.cm = 0.5f * (p.bias - p.thick) - 0.0000001f;
  c.hh = 0.5f * (p.bias + p.thick);
  if constexpr (kMode >= 3) {
    c.Axy = f32x2{pos.x * p.fx, pos.y * p.fy};
    c.Dz = pos.z + 0.0000001f;
  } else {
    c.Axy = f32x2{pos.x, pos.y};
    c.Dz = pos.z;
  }
  return c;
}
// rows of the tangent frame pre-scaled so that B = M * ts is the per-ray step vector (times j/step per sample)
struct FastTbn { v3 mx, my, mz; };
__device__ __forceinline__ CertPix make_cert(const GiParams& p, const FastPix& c, float inv_scale) {
  CertPix cp;
  cp.Axy = c.Axy * inv_scale;  // exact power-of-two scalings
  cp.cxy = c.cxy * inv_scale;
  cp.scale = 1.0f / inv_scale;
  cp.bw = p.cert_w;
  cp.bh = p.cert_h;
  return cp;
}
// xy_scale = 2^-cert_shift when the march runs in block units (exact), else 1
template <int kMode>
__device__ __forceinline__ FastTbn make_fast_tbn(const GiParams& p, const Tbn& m, float a, float xy_scale = 1.0f) {
  const float s = a * a * p.radius;
  const float sx = (kMode >= 3 ? s * p.fx : s) * xy_scale, sy = (kMode >= 3 ? s * p.fy : s) * xy_scale;
  FastTbn r;
  r.mx = {m.t.x * sx, m.b.x * sx, m.n.x * sx};
  r.my = {m.t.y * sy, m.b.y * sy, m.n.y * sy};
  r.mz = {m.t.z * s, m.b.z * s, m.n.z * s};
  return r;
}
template <int kMode>
__device__ __forceinline__ void fast_ray(const GiParams& p, const Tbn& tbn, const FastTbn& ft, float a, float4 ra, f32x2& bxy,
                                         float& bz) {
  float bx, by;
  if constexpr (kMode >= 3) {
    bx = __builtin_fmaf(ft.mx.z, ra.z, __builtin_fmaf(ft.mx.y, ra.y, ft.mx.x * ra.x));
    by = __builtin_fmaf(ft.my.z, ra.z, __builtin_fmaf(ft.my.y, ra.y, ft.my.x * ra.x));
    bz = __builtin_fmaf(ft.mz.z, ra.z, __builtin_fmaf(ft.mz.y, ra.y, ft.mz.x * ra.x));
  } else {
    const v3 sv = tbn_apply(tbn, ra.x, ra.y, ra.z);  // as the reference computes sampleVec
    bx = ((sv.x * a) * a) * p.radius;
    by = ((sv.y * a) * a) * p.radius;
    bz = ((sv.z * a) * a) * p.radius;
  }
  bxy = f32x2{bx, by};
}

#ifndef GIGS_GI_GROUP
#define GIGS_GI_GROUP 4
#endif
#ifndef GIGS_GI_RAYS
#define GIGS_GI_RAYS 2
#endif
constexpr int kGiGroup = GIGS_GI_GROUP;  // steps whose gathers are issued together
constexpr int kGiRays = GIGS_GI_RAYS;    // rays marched together per lane (independent instruction streams)
constexpr int kGiWaves = 4;
constexpr int kGiTileLog2W = 3;

// One 256-lane workgroup per 8x8 pixel tile: the four waves cover the SAME 64 pixels and split the
// ray set into four contiguous chunks (25 000 x 4 waves at 800x800 instead of 10 000 long-running
// ones: the tail of the launch is short and empty tiles retire immediately).  Partial sums are
// combined through LDS in the fixed order ((w0 + w1) + w2) + w3.
__device__ __forceinline__ bool gi_pixel(const GiParams& p, int& x, int& y, int& wave) {
  const int lane = threadIdx.x & 63;
  wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably uniform: the ray table is then read with s_load
  x = (blockIdx.x << p.tile_log2w) + (lane & ((1 << p.tile_log2w) - 1));
  y = blockIdx.y * (64 >> p.tile_log2w) + (lane >> p.tile_log2w);
  return x < p.W && y < p.H;
}

#ifndef GIGS_GI_WAVES
#define GIGS_GI_WAVES 0
#endif
#if GIGS_GI_WAVES > 0
#define GIGS_GI_OCC __attribute__((amdgpu_waves_per_eu(GIGS_GI_WAVES, GIGS_GI_WAVES)))
#define GIGS_SSR_OCC GIGS_GI_OCC
#else
#define GIGS_GI_OCC
// the certified SSR march allocates 100 VGPRs (4 waves/SIMD) on its own; asked for 5 it fits 96 without spilling and runs
// 9 % faster (1.29 -> 1.17 ms at C2; 6 waves spill and lose it again)
#define GIGS_SSR_OCC __attribute__((amdgpu_waves_per_eu(5)))
#endif

template <bool kPow2, int kMode, bool kCert>
__global__ void __launch_bounds__(256) GIGS_GI_OCC
ssao_kernel(GiParams p, const float4* __restrict__ rays, float sum_w,
            const float* __restrict__ nrm, const float* __restrict__ pos_map,
            float* __restrict__ occlusion, const float2* __restrict__ cert_tab) {
  __shared__ float s_part[kGiWaves][64];
  extern __shared__ float2 s_cert[];
  constexpr int kCertOff = sizeof(float) * kGiWaves * 64;  // dynamic LDS follows the static array
  if constexpr (kCert) {
    if ((unsigned)(size_t)s_cert != (unsigned)kCertOff) __builtin_trap();  // the march addresses the table by this constant
    const int nb = (p.cert_w + 1) * (p.cert_h + 1);
    for (int i = threadIdx.x; i < nb; i += 256) s_cert[i] = cert_tab[i];
    __syncthreads();
  }
  int x, y, wave;
  const bool inside = gi_pixel(p, x, y, wave);
  const int lane = threadIdx.x & 63;
  const size_t HW = (size_t)p.H * p.W;
  const size_t pix_id = inside ? (size_t)p.W * y + x : 0;
  float occ = 0.0f;
  if (inside && p.start < p.step) {
    const Tbn tbn = make_tbn({nrm[pix_id], nrm[HW + pix_id], nrm[2 * HW + pix_id]});
    if (!tbn_never_hits(tbn)) {
      const v3 pos = {pos_map[pix_id], pos_map[HW + pix_id], pos_map[2 * HW + pix_id]};
      const __amdgpu_buffer_rsrc_t pos_z = z_plane_rsrc(pos_map + 2 * HW, HW);
      const float a = 1 + pos.z / 100;
      const float cx = float(p.W) / 2.0f, cy = float(p.H) / 2.0f;
      // the live rays [0, n_live) are split across the waves; rays behind them (zero-weight ones a caller asked to be
      // marched all the same: none by default) go round-robin to the waves afterwards, so the live rays' partial sums do
      // not depend on them
      const int chunk = (p.n_live + kGiWaves - 1) / kGiWaves;
      const int r0 = wave * chunk, r1 = min(p.n_live, r0 + chunk);
      const bool mag_ok = gi_mag_ok(pos, a, p.radius);
      if constexpr (kMode > 0) {
        // absurd magnitudes (|pos| >= 2^59) are outside the fast marches' contract: such a pixel takes no hits
        if (mag_ok) {
          const FastPix c = make_fast<kMode>(p, pos, cx, cy);
          const float inv_scale = kCert ? __uint_as_float((127u - (unsigned)p.cert_shift) << 23) : 1.0f;  // 2^-shift
          const FastTbn ft = make_fast_tbn<kMode>(p, tbn, a, inv_scale);
          const CertPix cp = make_cert(p, c, inv_scale);
          // the four waves of a workgroup share its 64 pixels and split the ray set: contiguous quarters (a quarter of the
          // azimuths each), or interleaved pairs (every wave sees every direction: equal work per wave -- the workgroup
          // holds its slots until its slowest wave is done)
          const int rs = p.ray_interleave ? 2 * wave : r0, re = p.ray_interleave ? p.n_live : r1;
          const int rstep = p.ray_interleave ? 2 * kGiWaves : 2;
          auto pair = [&](int r, int rb) {
            f32x2 Bxy[2], Bz2;
            float bz0, bz1;
            fast_ray<kMode>(p, tbn, ft, a, rays[2 * r], Bxy[0], bz0);
            fast_ray<kMode>(p, tbn, ft, a, rays[2 * rb], Bxy[1], bz1);
            Bz2 = f32x2{bz0, bz1};
            int hit[2];
            if constexpr (kCert) march2_cert<kMode, kGiGroup, kCertOff>(p, c, cp, Bxy, Bz2, pos_z, hit);
            else march2_fast<kMode, kGiGroup>(p, c, Bxy, Bz2, pos_z, hit);
            occ += hit[0] >= 0 ? rays[2 * r + 1].y : 0.0f;
            occ += (hit[1] >= 0 && rb != r) ? rays[2 * rb + 1].y : 0.0f;
          };
          for (int r = rs; r < re; r += rstep) pair(r, min(r + 1, re - 1));  // an odd chunk marches its last ray twice and counts it once
          for (int r = p.n_live + wave; r < p.nrays; r += kGiWaves) pair(r, r);
        }
      } else {
      const unsigned sv_min_bits = gi_sv_min_bits(a, p.radius, p.inv_step);
      int r = r0;
      for (; r + kGiRays <= r1; r += kGiRays) {
        v3 sv[kGiRays];
        float w[kGiRays];
        int hit[kGiRays];
#pragma unroll
        for (int k = 0; k < kGiRays; k++) {
          const float4 ra = rays[2 * (r + k)];  // wave-uniform -> scalar loads
          w[k] = rays[2 * (r + k) + 1].y;
          sv[k] = tbn_apply(tbn, ra.x, ra.y, ra.z);
        }
        march<kPow2, kGiGroup, kGiRays>(p, pos, a, sv, cx, cy, pos_z, mag_ok, sv_min_bits, hit);
#pragma unroll
        for (int k = 0; k < kGiRays; k++) occ += hit[k] >= 0 ? w[k] : 0.0f;  // x + 0 is exact: ray order kept
      }
      auto single = [&](int q) {
        const float4 ra = rays[2 * q];
        const v3 sv = tbn_apply(tbn, ra.x, ra.y, ra.z);
        int hit;
        march<kPow2, kGiGroup, 1>(p, pos, a, &sv, cx, cy, pos_z, mag_ok, sv_min_bits, &hit);
        occ += hit >= 0 ? rays[2 * q + 1].y : 0.0f;
      };
      for (; r < r1; r++) single(r);
      for (r = p.n_live + wave; r < p.nrays; r += kGiWaves) single(r);
      }
    }
  }
  s_part[wave][lane] = occ;
  __syncthreads();
  if (wave == 0 && inside) {
    const float tot = ((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane];
    if (sum_w > 0.0f)
      occlusion[pix_id] = fmaxf(0.0f, fminf(1.0f, (float)(1.0 - (double)(tot / sum_w))));
    else
      occlusion[pix_id] = 1.0f;
  }
}

// The hit list of the indirect-light march (frozen-geometry reuse, pipeline.GeometryCache): WHICH pixel every ray of
// every pixel hits depends on normals and positions only -- not on the radiance that is gathered there -- so a view whose
// geometry does not change marches once and afterwards only gathers (ssr_apply_kernel).  kHits = 1 counts the hits of each
// (pixel, wave) -- the wave's rays are marched in a fixed order, so its hits form a sequence --, kHits = 2 writes them as
// (hit pixel, ray index) at the positions an exclusive prefix of the counts assigns.  Both also produce the normal outputs.
struct SsrHits {
  unsigned* counts;         // kHits 1: [4 N], entry 4 * pixel + wave
  const unsigned* offsets;  // kHits 2: [4 N + 1] exclusive prefix of the counts
  uint2* entries;           // kHits 2: {hit pixel, ray index}
  unsigned capacity;        // entries beyond it are dropped (the caller compares offsets[4 N] with it and repeats)
};

// the part of SSRCUDA behind the march (forward.cu:832-909; fresnelSchlick ssr.h:13-16): shared by the march and the gather
__device__ __forceinline__ void ssr_finish(v3 diffuse, const Tbn& tbn, v3 pos, size_t pix_id, size_t HW, int nrays_total,
                                           const float* __restrict__ albedo_map, const float* __restrict__ metallic_map,
                                           const float* __restrict__ F0_map, float* __restrict__ color, float* __restrict__ abd) {
  const v3 N = tbn.n;
  const v3 alb = {albedo_map[pix_id], albedo_map[HW + pix_id], albedo_map[2 * HW + pix_id]};
  const v3 F0 = {F0_map[pix_id], F0_map[HW + pix_id], F0_map[2 * HW + pix_id]};
  const float metallic = metallic_map[pix_id];
  const v3 V = normalize3({-pos.x, -pos.y, -pos.z});
  // fresnelSchlick (ssr.h:13-16): pow evaluated in double
  const float cosTheta = fmaxf(dot3(N, V), (float)0.0000001);
  const float pw = (float)pow((double)fminf(fmaxf((float)(1.0 - (double)cosTheta), (float)0.000001), 1.0f), 5.0);
  const v3 F = {F0.x + (1.0f - F0.x) * pw, F0.y + (1.0f - F0.y) * pw, F0.z + (1.0f - F0.z) * pw};
  v3 kD = {(float)(1.0 - (double)F.x), (float)(1.0 - (double)F.y), (float)(1.0 - (double)F.z)};
  kD.x = (float)((double)kD.x * (1.0 - (double)metallic));
  kD.y = (float)((double)kD.y * (1.0 - (double)metallic));
  kD.z = (float)((double)kD.z * (1.0 - (double)metallic));
  const float nrSamples = (float)nrays_total;  // += 1 per ray in fp32 is exact below 2^24
  v3 gd;
  if (nrSamples > 0.0f) {
    gd.x = (float)((double)(kPiF * diffuse.x) * (1.0 / (double)nrSamples) * (double)kD.x);
    gd.y = (float)((double)(kPiF * diffuse.y) * (1.0 / (double)nrSamples) * (double)kD.y);
    gd.z = (float)((double)(kPiF * diffuse.z) * (1.0 / (double)nrSamples) * (double)kD.z);
    diffuse = {gd.x * alb.x, gd.y * alb.y, gd.z * alb.z};
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

template <bool kPow2, int kMode, bool kCert, int kHits = 0>
__global__ void __launch_bounds__(256) GIGS_SSR_OCC
ssr_kernel(GiParams p, const float4* __restrict__ rays, const float* __restrict__ nrm,
           const float* __restrict__ pos_map, const float* __restrict__ rgb,
           const float* __restrict__ albedo_map, const float* __restrict__ metallic_map,
           const float* __restrict__ F0_map, float* __restrict__ color, float* __restrict__ abd,
           const float2* __restrict__ cert_tab, SsrHits hits) {
  __shared__ float s_part[kGiWaves][3][64];
  extern __shared__ float2 s_cert[];
  constexpr int kCertOff = sizeof(float) * kGiWaves * 3 * 64;  // dynamic LDS follows the static array
  if constexpr (kCert) {
    if ((unsigned)(size_t)s_cert != (unsigned)kCertOff) __builtin_trap();  // the march addresses the table by this constant
    const int nb = (p.cert_w + 1) * (p.cert_h + 1);
    for (int i = threadIdx.x; i < nb; i += 256) s_cert[i] = cert_tab[i];
    __syncthreads();
  }
  int x, y, wave;
  const bool inside = gi_pixel(p, x, y, wave);
  const int lane = threadIdx.x & 63;
  const size_t HW = (size_t)p.H * p.W;
  const size_t pix_id = inside ? (size_t)p.W * y + x : 0;
  const v3 pos = {pos_map[pix_id], pos_map[HW + pix_id], pos_map[2 * HW + pix_id]};
  const Tbn tbn = make_tbn({nrm[pix_id], nrm[HW + pix_id], nrm[2 * HW + pix_id]});
  v3 diffuse = {0, 0, 0};
  unsigned hpos = 0;  // kHits: hits of this (pixel, wave) so far / the next entry's position
  if (inside && p.start < p.step && !tbn_never_hits(tbn)) {
    const __amdgpu_buffer_rsrc_t pos_z = z_plane_rsrc(pos_map + 2 * HW, HW);
    const float a = 1 + pos.z / 100;
    const float cx = float(p.W) / 2.0f, cy = float(p.H) / 2.0f;
    const int chunk = (p.n_live + kGiWaves - 1) / kGiWaves;  // see ssao_kernel
    const int r0 = wave * chunk, r1 = min(p.n_live, r0 + chunk);
    const bool mag_ok = gi_mag_ok(pos, a, p.radius);
    if constexpr (kHits == 2) hpos = hits.offsets[4 * pix_id + wave];
    auto add_hit = [&](int q, float cos_t, float sin_t, int ray) {
      if (q >= 0) {
        // rgb * cosf(theta) * sinf(theta), left to right (forward.cu:824-826)
        diffuse.x += rgb[q] * cos_t * sin_t;
        diffuse.y += rgb[HW + q] * cos_t * sin_t;
        diffuse.z += rgb[2 * HW + q] * cos_t * sin_t;
        if constexpr (kHits == 1) hpos++;
        if constexpr (kHits == 2) {
          if (hpos < hits.capacity) hits.entries[hpos] = make_uint2((unsigned)q, (unsigned)ray);
          hpos++;
        }
      }
      (void)ray;
    };
    if constexpr (kMode > 0) {
      if (mag_ok) {  // see ssao_kernel
        const FastPix c = make_fast<kMode>(p, pos, cx, cy);
        const float inv_scale = kCert ? __uint_as_float((127u - (unsigned)p.cert_shift) << 23) : 1.0f;  // 2^-shift
        const FastTbn ft = make_fast_tbn<kMode>(p, tbn, a, inv_scale);
        const CertPix cp = make_cert(p, c, inv_scale);
        const int rs = p.ray_interleave ? 2 * wave : r0, re = p.ray_interleave ? p.n_live : r1;
        const int rstep = p.ray_interleave ? 2 * kGiWaves : 2;
        auto pair = [&](int r, int rb) {
          const float4 ra0 = rays[2 * r], ra1 = rays[2 * rb];
          f32x2 Bxy[2], Bz2;
          float bz0, bz1;
          fast_ray<kMode>(p, tbn, ft, a, ra0, Bxy[0], bz0);
          fast_ray<kMode>(p, tbn, ft, a, ra1, Bxy[1], bz1);
          Bz2 = f32x2{bz0, bz1};
          int hit[2];
          if constexpr (kCert) march2_cert<kMode, kGiGroup, kCertOff>(p, c, cp, Bxy, Bz2, pos_z, hit);
          else march2_fast<kMode, kGiGroup>(p, c, Bxy, Bz2, pos_z, hit);
          if (rb == r) hit[1] = -1;
          if (__any(hit[0] >= 0 || hit[1] >= 0)) {
            add_hit(hit[0], ra0.w, rays[2 * r + 1].x, r);
            add_hit(hit[1], ra1.w, rays[2 * rb + 1].x, rb);
          }
        };
        for (int r = rs; r < re; r += rstep) pair(r, min(r + 1, re - 1));
        // the zero-weight ray (ONE representative by default): rgb * cos * 0 = +-0, or NaN for a non-finite hit pixel
        for (int r = p.n_live + wave; r < p.nrays; r += kGiWaves) pair(r, r);
      }
    } else {
    const unsigned sv_min_bits = gi_sv_min_bits(a, p.radius, p.inv_step);
    int r = r0;
    for (; r + kGiRays <= r1; r += kGiRays) {
      v3 sv[kGiRays];
      float ct[kGiRays], st[kGiRays];
      int hit[kGiRays];
#pragma unroll
      for (int k = 0; k < kGiRays; k++) {
        const float4 ra = rays[2 * (r + k)];
        ct[k] = ra.w;
        st[k] = rays[2 * (r + k) + 1].x;
        sv[k] = tbn_apply(tbn, ra.x, ra.y, ra.z);
      }
      march<kPow2, kGiGroup, kGiRays>(p, pos, a, sv, cx, cy, pos_z, mag_ok, sv_min_bits, hit);
      bool any_hit = false;
#pragma unroll
      for (int k = 0; k < kGiRays; k++) any_hit = any_hit || hit[k] >= 0;
      if (__any(any_hit)) {  // hits are rare: one wave-uniform test per ray pair
#pragma unroll
        for (int k = 0; k < kGiRays; k++) add_hit(hit[k], ct[k], st[k], r + k);
      }
    }
    auto single = [&](int q) {
      const float4 ra = rays[2 * q];
      const v3 sv = tbn_apply(tbn, ra.x, ra.y, ra.z);
      int hit;
      march<kPow2, kGiGroup, 1>(p, pos, a, &sv, cx, cy, pos_z, mag_ok, sv_min_bits, &hit);
      add_hit(hit, ra.w, rays[2 * q + 1].x, q);
    };
    for (; r < r1; r++) single(r);
    for (r = p.n_live + wave; r < p.nrays; r += kGiWaves) single(r);
    }
  }
  if constexpr (kHits == 1) {
    if (inside) hits.counts[4 * pix_id + wave] = hpos;
  }
  s_part[wave][0][lane] = diffuse.x;
  s_part[wave][1][lane] = diffuse.y;
  s_part[wave][2][lane] = diffuse.z;
  __syncthreads();
  if (wave != 0 || !inside) return;
  diffuse.x = ((s_part[0][0][lane] + s_part[1][0][lane]) + s_part[2][0][lane]) + s_part[3][0][lane];
  diffuse.y = ((s_part[0][1][lane] + s_part[1][1][lane]) + s_part[2][1][lane]) + s_part[3][1][lane];
  diffuse.z = ((s_part[0][2][lane] + s_part[1][2][lane]) + s_part[2][2][lane]) + s_part[3][2][lane];
  ssr_finish(diffuse, tbn, pos, pix_id, HW, p.nrays_total, albedo_map, metallic_map, F0_map, color, abd);
}

// The gather that replaces the march for a view whose hit list is known (see SsrHits): per pixel the four waves' hit
// sequences are summed in their recorded order, the four partial sums combined as the march combines them, and the same
// tail evaluated -- the march's outputs bit for bit for the geometry the list was recorded with.
__global__ void __launch_bounds__(256)
ssr_apply_kernel(int W, int H, int nrays_total, const float4* __restrict__ rays, const unsigned* __restrict__ offsets,
                 const uint2* __restrict__ entries, const float* __restrict__ nrm, const float* __restrict__ pos_map,
                 const float* __restrict__ rgb, const float* __restrict__ albedo_map, const float* __restrict__ metallic_map,
                 const float* __restrict__ F0_map, float* __restrict__ color, float* __restrict__ abd) {
  const size_t HW = (size_t)H * W;
  const size_t pix_id = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix_id >= HW) return;
  const v3 pos = {pos_map[pix_id], pos_map[HW + pix_id], pos_map[2 * HW + pix_id]};
  const Tbn tbn = make_tbn({nrm[pix_id], nrm[HW + pix_id], nrm[2 * HW + pix_id]});
  v3 part[4];
#pragma unroll
  for (int w = 0; w < 4; w++) {
    v3 d = {0, 0, 0};
    const unsigned e0 = offsets[4 * pix_id + w], e1 = offsets[4 * pix_id + w + 1];
    for (unsigned e = e0; e < e1; e++) {
      const uint2 h = entries[e];
      const float cos_t = rays[2 * h.y].w, sin_t = rays[2 * h.y + 1].x;
      d.x += rgb[h.x] * cos_t * sin_t;
      d.y += rgb[HW + h.x] * cos_t * sin_t;
      d.z += rgb[2 * HW + h.x] * cos_t * sin_t;
    }
    part[w] = d;
  }
  const v3 diffuse = {((part[0].x + part[1].x) + part[2].x) + part[3].x, ((part[0].y + part[1].y) + part[2].y) + part[3].y,
                      ((part[0].z + part[1].z) + part[2].z) + part[3].z};
  ssr_finish(diffuse, tbn, pos, pix_id, HW, nrays_total, albedo_map, metallic_map, F0_map, color, abd);
}

static GiParams make_params(const Options& o, int W, int H, float fx, float fy, float radius, float bias, float thick,
                            int step, int start, const RayTable& t, bool ssr, bool& pow2) {
  GiParams p;
  p.W = W; p.H = H; p.fx = fx; p.fy = fy; p.radius = radius; p.bias = bias; p.thick = thick;
  p.step = step; p.start = start;
  p.n_live = t.n_live;
  // zero-weight rays (see the ray table): all of them on request, else one for SSR (NaN propagation) and none for SSAO
  p.nrays = t.n_live + (o.gi_zero_rays ? t.n_dead : (ssr && t.n_dead > 0 ? 1 : 0));
  p.nrays_total = t.nrays;
  pow2 = step > 0 && (step & (step - 1)) == 0 && step <= (1 << 20);
  p.inv_step = pow2 ? 1.0f / (float)step : 0.0f;
  if (step - start > 64 - kGiGroup) pow2 = false;  // a partial last group may index kGiGroup - 1 entries past step - 1
  // j / step: exact for a power-of-two step; otherwise the correctly rounded quotient (read by the fast marches only)
  for (int k = 0; k < 64; k++) p.fjt[k] = pow2 ? (float)(start + k) * p.inv_step : (float)(start + k) / (float)step;
  for (int k = 0; k < 64; k++) p.fjc[k] = p.fjt[k < step - 1 - start ? k : (step - 1 - start > 0 ? step - 1 - start : 0)];
  // gi_tile_log2w: tuning knob for the pixel rectangle of a workgroup (3 = 8x8 ... 6 = 64x1)
  p.tile_log2w = (o.gi_tile_log2w >= 0 && o.gi_tile_log2w <= 6) ? o.gi_tile_log2w : kGiTileLog2W;
  p.ray_interleave = o.gi_interleave ? 1 : 0;  // measured at C2: SSAO 0.94 -> 0.89, SSR 0.93 -> 0.90 ms (0: quarters)
  return p;
}
// Certification table, (bw + 1) x (bh + 1) entries.  Per (1 << shift)^2 block of the z plane: the minimum over its non-zero
// pixels (+inf if none) and max(maximum, 0) -- NaN pixels are ignored (a NaN never passes the depth test) -- turned into
// the two thresholds {t_lo, t_hi} on a sample's denominator (CertK): the march then certifies with three compares and no
// arithmetic.  The last column / row (indices bw / bh: where out-of-image block coordinates clamp to, negative ones
// included, being huge as unsigned) and blocks the image covers only partly hold {-inf, +inf}: nothing is ever certified
// there, so samples that may lie outside the image always take the exact path.  One wave per entry.
__global__ void __launch_bounds__(64)
gi_minmax_kernel(int W, int H, int shift, int bw, int bh, CertK ck, const float* __restrict__ z, float2* __restrict__ tab) {
  const int bx = (int)blockIdx.x, by = (int)blockIdx.y, B = 1 << shift;
  float mn = __builtin_inff(), mx = 0.0f;
  const bool full = bx < bw && by < bh && ((bx + 1) << shift) <= W && ((by + 1) << shift) <= H;
  if (full) {
    for (int i = threadIdx.x; i < B * B; i += 64) {
      const float v = z[(size_t)((by << shift) + (i >> shift)) * W + (bx << shift) + (i & (B - 1))];
      if (v != 0.0f) mn = fminf(mn, v);  // false for NaN too: fminf would ignore it anyway
      mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, o));
      mx = fmaxf(mx, __shfl_xor(mx, o));
    }
  } else {
    mn = -__builtin_inff();
    mx = __builtin_inff();
  }
  if (threadIdx.x == 0) {
    // thresholds on den (see CertK), each moved away from certification by 2^-20 relative
    float t_hi = (mx - ck.lo_k) / (1.0f - 0x1p-16f);
    t_hi = t_hi + fabsf(t_hi) * 0x1p-20f + 1e-30f;
    float t_lo = (mn - ck.hi_k) / (1.0f + 0x1p-16f);
    t_lo = t_lo - fabsf(t_lo) * 0x1p-20f - 1e-30f;
    if (!(mx == mx) || !(mn == mn)) { t_hi = __builtin_inff(); t_lo = -__builtin_inff(); }  // NaN depths: never certify
    tab[(size_t)by * (bw + 1) + bx] = make_float2(t_lo, t_hi);
  }
}

// block size of the certification table: the smallest of 16 / 32 / 64 pixels whose table fits kCertMaxBytes of LDS
constexpr int kCertMaxBytes = 40 * 1024;
static size_t cert_table_bytes(int W, int H, int sh) {
  return (size_t)(((W + (1 << sh) - 1) >> sh) + 2) * (((H + (1 << sh) - 1) >> sh) + 2) * sizeof(float2);
}
static int cert_shift_for(int W, int H) {
  for (int sh = 4; sh <= 6; sh++)
    if (cert_table_bytes(W, H, sh) <= (size_t)kCertMaxBytes) return sh;
  return 0;
}
size_t gi_scratch_bytes(int W, int H) {
  const int sh = cert_shift_for(W, H);
  return sh ? cert_table_bytes(W, H, sh) : 0;
}
// fills p.cert_*; returns the table's byte size (0 = run without certification)
static size_t prepare_cert(const Options& o, GiParams& p, int mode, const float* pos, void* scratch, hipStream_t s) {
  p.cert_shift = p.cert_w = p.cert_h = 0;
  p.cert_d0 = cert_consts(p.bias, p.thick).d0;
  if (!scratch || mode < 3 || !o.gi_cert) return 0;
  const int sh = cert_shift_for(p.W, p.H);
  if (!sh) return 0;
  p.cert_shift = sh;
  p.cert_w = (p.W + (1 << sh) - 1) >> sh;
  p.cert_h = (p.H + (1 << sh) - 1) >> sh;
  hipLaunchKernelGGL(gi_minmax_kernel, dim3(p.cert_w + 1, p.cert_h + 1), dim3(64), 0, s, p.W, p.H, sh, p.cert_w, p.cert_h,
                     cert_consts(p.bias, p.thick), pos + 2 * (size_t)p.H * p.W, (float2*)scratch);
  return cert_table_bytes(p.W, p.H, sh);
}

// gigs_options.gi_march: 0 exact | 1 hoist | 2 hoist_fma | 3 proj_nr | 4 proj (see the block comment above march2_fast;
// GIGS_GI_MARCH names them in the environment).  The fast marches read the j/step table, so marches of more than
// 64 - kGiGroup steps take the exact path.
#ifndef GIGS_GI_DEFAULT_MODE
#define GIGS_GI_DEFAULT_MODE 4  // "proj": measured 1.3e-7 mean L1 / 1e-4 changed pixels vs the exact march at C2
#endif
static int gi_march_mode(const Options& o, int step, int start) {
  int mode = (o.gi_march >= 0 && o.gi_march <= 4) ? o.gi_march : GIGS_GI_DEFAULT_MODE;
  if (step - start > 64 - kGiGroup || step <= 0) mode = 0;
  return mode;
}
static dim3 gi_grid(const GiParams& p) {
  const int tw = 1 << p.tile_log2w, th = 64 >> p.tile_log2w;
  return dim3((p.W + tw - 1) / tw, (p.H + th - 1) / th);
}

int launch_ssao(const Options& o, int W, int H, float fx, float fy, float radius, float bias, float thick,
                float delta, int step, int start, const float* normal, const float* pos,
                float* occlusion, void* scratch, hipStream_t s) {
  RayTable t;
  const int rc = get_ray_table(delta, s, t);
  if (rc) return rc;
  bool pow2;
  GiParams p = make_params(o, W, H, fx, fy, radius, bias, thick, step, start, t, /*ssr*/ false, pow2);
  if (W >= (1 << 15) || H >= (1 << 15)) return -2;  // rejected with a message by the C-ABI wrapper
  const dim3 grid = gi_grid(p);
  const int mode = gi_march_mode(o, step, start);
  const size_t cert = (start < step) ? prepare_cert(o, p, mode, pos, scratch, s) : 0;
  const float2* tab = (const float2*)scratch;
#define GIGS_SSAO_LAUNCH(POW2, MODE, CERT, LDS) \
  hipLaunchKernelGGL((ssao_kernel<POW2, MODE, CERT>), grid, dim3(256), LDS, s, p, t.dev, t.sum_w, normal, pos, occlusion, tab)
  switch (mode) {
    case 1: GIGS_SSAO_LAUNCH(false, 1, false, 0); break;
    case 2: GIGS_SSAO_LAUNCH(false, 2, false, 0); break;
    case 3: if (cert) GIGS_SSAO_LAUNCH(false, 3, true, cert); else GIGS_SSAO_LAUNCH(false, 3, false, 0); break;
    case 4: if (cert) GIGS_SSAO_LAUNCH(false, 4, true, cert); else GIGS_SSAO_LAUNCH(false, 4, false, 0); break;
    default:
      if (pow2) GIGS_SSAO_LAUNCH(true, 0, false, 0);
      else GIGS_SSAO_LAUNCH(false, 0, false, 0);
  }
#undef GIGS_SSAO_LAUNCH
  return 0;
}

int launch_ssr_apply(int W, int H, float delta, const unsigned* offsets, const void* entries, const float* normal,
                     const float* pos, const float* rgb, const float* albedo, const float* metallic, const float* F0,
                     float* color, float* abd, hipStream_t s) {
  RayTable t;
  const int rc = get_ray_table(delta, s, t);
  if (rc) return rc;
  const size_t HW = (size_t)W * H;
  hipLaunchKernelGGL(ssr_apply_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, s, W, H, t.nrays, t.dev, offsets,
                     (const uint2*)entries, normal, pos, rgb, albedo, metallic, F0, color, abd);
  return 0;
}

int launch_ssr(const Options& o, int W, int H, float fx, float fy, float radius, float bias, float thick,
               float delta, int step, int start, const float* normal, const float* pos,
               const float* rgb, const float* albedo, const float* /*roughness*/,
               const float* metallic, const float* F0, float* color, float* abd, void* scratch, hipStream_t s,
               int hits_mode, unsigned* hit_counts, const unsigned* hit_offsets, void* hit_entries, unsigned hit_capacity) {
  RayTable t;
  const int rc = get_ray_table(delta, s, t);
  if (rc) return rc;
  bool pow2;
  GiParams p = make_params(o, W, H, fx, fy, radius, bias, thick, step, start, t, /*ssr*/ true, pow2);
  if (W >= (1 << 15) || H >= (1 << 15)) return -2;  // rejected with a message by the C-ABI wrapper
  const dim3 grid = gi_grid(p);
  const int mode = gi_march_mode(o, step, start);
  const size_t cert = (start < step) ? prepare_cert(o, p, mode, pos, scratch, s) : 0;
  const float2* tab = (const float2*)scratch;
  const SsrHits hits = {hit_counts, hit_offsets, (uint2*)hit_entries, hit_capacity};
  if (hits_mode != 0) {
    // the hit-list variants exist for the default march only (mode 4, with or without certification); -3 = not available
    if (mode != 4 || (hits_mode != 1 && hits_mode != 2) || !(start < step)) return -3;
    if (hits_mode == 1) {
      if (cert) hipLaunchKernelGGL((ssr_kernel<false, 4, true, 1>), grid, dim3(256), cert, s, p, t.dev, normal, pos, rgb, albedo, metallic, F0, color, abd, tab, hits);
      else hipLaunchKernelGGL((ssr_kernel<false, 4, false, 1>), grid, dim3(256), 0, s, p, t.dev, normal, pos, rgb, albedo, metallic, F0, color, abd, tab, hits);
    } else {
      if (cert) hipLaunchKernelGGL((ssr_kernel<false, 4, true, 2>), grid, dim3(256), cert, s, p, t.dev, normal, pos, rgb, albedo, metallic, F0, color, abd, tab, hits);
      else hipLaunchKernelGGL((ssr_kernel<false, 4, false, 2>), grid, dim3(256), 0, s, p, t.dev, normal, pos, rgb, albedo, metallic, F0, color, abd, tab, hits);
    }
    return 0;
  }
#define GIGS_SSR_LAUNCH(POW2, MODE, CERT, LDS)                                                                          \
  hipLaunchKernelGGL((ssr_kernel<POW2, MODE, CERT>), grid, dim3(256), LDS, s, p, t.dev, normal, pos, rgb, albedo, metallic, \
                     F0, color, abd, tab, hits)
  switch (mode) {
    case 1: GIGS_SSR_LAUNCH(false, 1, false, 0); break;
    case 2: GIGS_SSR_LAUNCH(false, 2, false, 0); break;
    case 3: if (cert) GIGS_SSR_LAUNCH(false, 3, true, cert); else GIGS_SSR_LAUNCH(false, 3, false, 0); break;
    case 4: if (cert) GIGS_SSR_LAUNCH(false, 4, true, cert); else GIGS_SSR_LAUNCH(false, 4, false, 0); break;
    default:
      if (pow2) GIGS_SSR_LAUNCH(true, 0, false, 0);
      else GIGS_SSR_LAUNCH(false, 0, false, 0);
  }
#undef GIGS_SSR_LAUNCH
  return 0;
}

// ------------------------------------------------------------------------------------------
// 3x3 filters
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool load_taps(const float* __restrict__ src, int H, int W, int y, int x,
                                          float* v) {
  bool has_nan = false;
  int k = 0;
#pragma unroll
  for (int dy = -1; dy <= 1; dy++)
#pragma unroll
    for (int dx = -1; dx <= 1; dx++) {
      const int yy = y + dy, xx = x + dx;
      const float t = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0.0f : src[(size_t)yy * W + xx];
      has_nan |= (t != t);
      v[k++] = t;
    }
  return has_nan;
}

__global__ void __launch_bounds__(256)
median3x3_kernel(int H, int W, const float* __restrict__ in, float* __restrict__ out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const float* src = in + (size_t)blockIdx.z * H * W;
  float v[9];
  const bool has_nan = load_taps(src, H, W, y, x, v);
  out[(size_t)blockIdx.z * H * W + (size_t)y * W + x] = has_nan ? __builtin_nanf("") : median9(v);
}

// Gradient of the median: the whole output gradient goes to the first tap (row-major tap
// order) whose value equals the median; NaN windows pass no gradient.
__global__ void __launch_bounds__(256)
median3x3_bwd_kernel(int H, int W, const float* __restrict__ in, const float* __restrict__ gout,
                     float* __restrict__ gin) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const size_t plane = (size_t)blockIdx.z * H * W;
  const float* src = in + plane;
  float v[9], s[9];
  const bool has_nan = load_taps(src, H, W, y, x, v);
  if (has_nan) return;
#pragma unroll
  for (int k = 0; k < 9; k++) s[k] = v[k];
  const float med = median9(s);
  const float g = gout[plane + (size_t)y * W + x];
  if (g == 0.0f) return;
  int k = 0;
  for (int dy = -1; dy <= 1; dy++)
    for (int dx = -1; dx <= 1; dx++, k++) {
      if (v[k] == med) {
        const int yy = y + dy, xx = x + dx;
        if (!(yy < 0 || yy >= H || xx < 0 || xx >= W)) atomicAdd(gin + plane + (size_t)yy * W + xx, g);
        return;  // a padding tap selected: gradient is dropped
      }
    }
}

void launch_median3x3(int C, int H, int W, const float* in, float* out, hipStream_t s) {
  dim3 grid((W + 63) / 64, (H + 3) / 4, C);
  hipLaunchKernelGGL(median3x3_kernel, grid, dim3(256), 0, s, H, W, in, out);
}
void launch_median3x3_bwd(int C, int H, int W, const float* in, const float* gout, float* gin,
                          hipStream_t s) {
  dim3 grid((W + 63) / 64, (H + 3) / 4, C);
  hipLaunchKernelGGL(median3x3_bwd_kernel, grid, dim3(256), 0, s, H, W, in, gout, gin);
}

struct BilatK { float ky[3], kx[3]; float color_scale; };

template <int C>
__global__ void __launch_bounds__(256)
bilateral3x3_kernel(int H, int W, BilatK kk, const float* __restrict__ in, float* __restrict__ out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const size_t HW = (size_t)H * W;
  float ctr[C], num[C];
#pragma unroll
  for (int c = 0; c < C; c++) {
    ctr[c] = in[c * HW + (size_t)y * W + x];
    num[c] = 0.0f;
  }
  float den = 0.0f;
#pragma unroll
  for (int dy = -1; dy <= 1; dy++)
#pragma unroll
    for (int dx = -1; dx <= 1; dx++) {
      int yy = y + dy, xx = x + dx;
      yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);  // reflect
      xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
      float tap[C], dist = 0.0f;
#pragma unroll
      for (int c = 0; c < C; c++) {
        tap[c] = in[c * HW + (size_t)yy * W + xx];
        dist += fabsf(tap[c] - ctr[c]);
      }
      const float color_k = expf(kk.color_scale * (dist * dist));
      const float k = (kk.ky[dy + 1] * kk.kx[dx + 1]) * color_k;
#pragma unroll
      for (int c = 0; c < C; c++) num[c] += tap[c] * k;
      den += k;
    }
#pragma unroll
  for (int c = 0; c < C; c++) out[c * HW + (size_t)y * W + x] = num[c] / den;
}

void launch_bilateral3x3(int C, int H, int W, float sigma_color, float sx, float sy,
                         const float* in, float* out, hipStream_t s) {
  BilatK kk;
  auto k1d = [](float sigma, float* k) {
    float sum = 0;
    for (int i = 0; i < 3; i++) {
      const float xv = (float)(i - 1);
      k[i] = expf(-(xv * xv) / (2.0f * sigma * sigma));
      sum += k[i];
    }
    for (int i = 0; i < 3; i++) k[i] /= sum;
  };
  k1d(sy, kk.ky);
  k1d(sx, kk.kx);
  kk.color_scale = -0.5f / (sigma_color * sigma_color);
  dim3 grid((W + 63) / 64, (H + 3) / 4, 1);
  if (C == 1) hipLaunchKernelGGL(bilateral3x3_kernel<1>, grid, dim3(256), 0, s, H, W, kk, in, out);
  else if (C == 3) hipLaunchKernelGGL(bilateral3x3_kernel<3>, grid, dim3(256), 0, s, H, W, kk, in, out);
}

// ------------------------------------------------------------------------------------------
// The four passes between the blend kernel and SSAO as one launch
// ------------------------------------------------------------------------------------------
// GaussianRasterizer.forward (R/.../__init__.py:475-517) chains median3x3(depth) -> depth_to_normal -> bilateral3x3
// (normal) and median3x3(depth_pos): four small kernels (0.16 ms) with a launch gap after each, all on the critical path
// in front of the SSAO march.  Here a 32x4 pixel tile stages the raw depth with a 4-pixel halo in LDS and runs the same
// stages over shrinking halos (median: 3, normal / position: 1, outputs: 0).  Every stage evaluates exactly the
// expressions of the stand-alone kernels above (same taps, same padding rules, same operation order), so the outputs are
// bit-identical to the four-kernel chain (tests/test_gpu_parity.py compares them).  Small tiles win: the kernel is
// latency-bound (four dependent LDS stages), so more, smaller workgroups beat the smaller halo overhead of large ones.
#ifndef GIGS_DN_H
#define GIGS_DN_H 4  // measured at 800x800: tile heights 2 / 4 / 8 / 16 / 32 -> 100 / 70 / 97 / 114 / 135 us
#endif
constexpr int kDnW = 32, kDnH = GIGS_DN_H, kDnThreads = kDnW * kDnH;

__global__ void __launch_bounds__(kDnThreads)
derive_normal_fused_kernel(int W, int H, float fx, float fy, const float* __restrict__ vm, BilatK kk,
                           const float* __restrict__ depth_raw, float* __restrict__ normal_out,
                           float* __restrict__ pos_filter_out) {
  __shared__ float s_d[kDnH + 8][kDnW + 8];     // raw depth, halo 4; 0 outside the image (median_blur's zero padding)
  __shared__ float s_f[kDnH + 6][kDnW + 6];     // median-filtered depth, halo 3
  __shared__ float s_n[3][kDnH + 2][kDnW + 2];  // depth_to_normal's normal, halo 1
  __shared__ float s_p[3][kDnH + 2][kDnW + 2];  // depth_to_normal's position, halo 1; 0 outside the image
  const int x0 = blockIdx.x * kDnW, y0 = blockIdx.y * kDnH;
  const size_t HW = (size_t)H * W;

  for (int i = threadIdx.x; i < (kDnH + 8) * (kDnW + 8); i += kDnThreads) {
    const int ty = i / (kDnW + 8), tx = i - ty * (kDnW + 8);
    const int gx = x0 - 4 + tx, gy = y0 - 4 + ty;
    s_d[ty][tx] = (gx < 0 || gx >= W || gy < 0 || gy >= H) ? 0.0f : depth_raw[(size_t)gy * W + gx];
  }
  __syncthreads();

  for (int i = threadIdx.x; i < (kDnH + 6) * (kDnW + 6); i += kDnThreads) {  // = median3x3_kernel on the depth plane
    const int ty = i / (kDnW + 6), tx = i - ty * (kDnW + 6);
    const int gx = x0 - 3 + tx, gy = y0 - 3 + ty;
    float r = 0.0f;
    if (!(gx < 0 || gx >= W || gy < 0 || gy >= H)) {
      float v[9];
      bool has_nan = false;
      int k = 0;
#pragma unroll
      for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
          const float t = s_d[ty + 1 + dy][tx + 1 + dx];
          has_nan |= (t != t);
          v[k++] = t;
        }
      r = has_nan ? __builtin_nanf("") : median9(v);
    }
    s_f[ty][tx] = r;
  }
  __syncthreads();

  for (int i = threadIdx.x; i < (kDnH + 2) * (kDnW + 2); i += kDnThreads) {  // = depth_to_normal_kernel
    const int ty = i / (kDnW + 2), tx = i - ty * (kDnW + 2);
    const int x = x0 - 1 + tx, y = y0 - 1 + ty;
    v3 nrm = {0.0f, 0.0f, 0.0f}, pos = {0.0f, 0.0f, 0.0f};
    if (!(x < 0 || x >= W || y < 0 || y >= H) && !(x <= 0 || x >= W - 1 || y <= 0 || y >= H - 1)) {
#define DP(dx, dy) s_f[ty + 2 + (dy)][tx + 2 + (dx)]
      const float depth_thresh = 0.01f;
      const float depth = DP(0, 0);
      const float cx = float(W) / 2.0f, cy = float(H) / 2.0f;
      pos = get_position(x, y, cx, cy, fx, fy, depth);
      bool ok = !(depth < depth_thresh);
      const int pad = 2;
      for (int dx = -pad; ok && dx < pad + 1; ++dx) {
        if (x + dx < 0 || x + dx > W - 1) { ok = false; break; }
        for (int dy = -pad; dy < pad + 1; ++dy) {
          if (y + dy < 0 || y + dy > H - 1) { ok = false; break; }
          if (DP(dx, dy) < depth_thresh) { ok = false; break; }
        }
      }
      if (ok) {
        const v3 pos_aa = get_position(x, y - 1, cx, cy, fx, fy, DP(0, -1));
        const v3 pos_bb = get_position(x + 1, y, cx, cy, fx, fy, DP(1, 0));
        const v3 pos_cc = get_position(x, y + 1, cx, cy, fx, fy, DP(0, 1));
        const v3 pos_dd = get_position(x - 1, y, cx, cy, fx, fy, DP(-1, 0));
        const v3 pos_ab = get_position(x + 1, y - 1, cx, cy, fx, fy, DP(1, -1));
        const v3 pos_bc = get_position(x + 1, y + 1, cx, cy, fx, fy, DP(1, 1));
        const v3 pos_cd = get_position(x - 1, y + 1, cx, cy, fx, fy, DP(-1, 1));
        const v3 pos_da = get_position(x - 1, y - 1, cx, cy, fx, fy, DP(-1, -1));
        const v3 edge_a = pos_da - pos_ab, edge_b = pos_ab - pos_bc, edge_c = pos_bc - pos_cd, edge_d = pos_cd - pos_da;
        const v3 edge_ac = pos_cc - pos_aa, edge_bd = pos_dd - pos_bb;
        const v3 edge_cdab = pos_ab - pos_cd, edge_bcad = pos_da - pos_bc;
        const v3 n1 = cross3(edge_a, edge_d), n2 = cross3(edge_d, edge_c), n3 = cross3(edge_c, edge_b);
        const v3 n4 = cross3(edge_b, edge_a), n5 = cross3(edge_ac, edge_bd), n6 = cross3(edge_bcad, edge_cdab);
        const v3 sum = normalize3(n1) + normalize3(n2) + normalize3(n3) + normalize3(n4) + normalize3(n5) + normalize3(n6);
        const float inv6 = 1.0f / 6.0f;
        const v3 normal = sum * inv6;
        nrm = {vm[0] * normal.x + vm[1] * normal.y + vm[2] * normal.z, vm[4] * normal.x + vm[5] * normal.y + vm[6] * normal.z,
               vm[8] * normal.x + vm[9] * normal.y + vm[10] * normal.z};
      }
#undef DP
    }
    s_n[0][ty][tx] = nrm.x; s_n[1][ty][tx] = nrm.y; s_n[2][ty][tx] = nrm.z;
    s_p[0][ty][tx] = pos.x; s_p[1][ty][tx] = pos.y; s_p[2][ty][tx] = pos.z;
  }
  __syncthreads();

  const int lx = threadIdx.x & (kDnW - 1), ly = threadIdx.x / kDnW;
  const int x = x0 + lx, y = y0 + ly;
  if (x >= W || y >= H) return;
  const size_t p = (size_t)y * W + x;
  {  // = bilateral3x3_kernel<3> on the normal
    float ctr[3], num[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 3; c++) ctr[c] = s_n[c][ly + 1][lx + 1];
    float den = 0.0f;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
      for (int dx = -1; dx <= 1; dx++) {
        int yy = y + dy, xx = x + dx;
        yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);  // reflect
        xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
        float tap[3], dist = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
          tap[c] = s_n[c][yy - y0 + 1][xx - x0 + 1];
          dist += fabsf(tap[c] - ctr[c]);
        }
        const float color_k = expf(kk.color_scale * (dist * dist));
        const float k = (kk.ky[dy + 1] * kk.kx[dx + 1]) * color_k;
#pragma unroll
        for (int c = 0; c < 3; c++) num[c] += tap[c] * k;
        den += k;
      }
#pragma unroll
    for (int c = 0; c < 3; c++) normal_out[c * HW + p] = num[c] / den;
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {  // = median3x3_kernel on the three position planes
    float v[9];
    bool has_nan = false;
    int k = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
      for (int dx = -1; dx <= 1; dx++) {
        const float t = s_p[c][ly + 1 + dy][lx + 1 + dx];
        has_nan |= (t != t);
        v[k++] = t;
      }
    pos_filter_out[c * HW + p] = has_nan ? __builtin_nanf("") : median9(v);
  }
}

void launch_derive_normal_fused(int W, int H, float fx, float fy, const float* viewmatrix, float sigma_color, float sx,
                                float sy, const float* depth_raw, float* normal_out, float* pos_filter_out, hipStream_t s) {
  BilatK kk;
  auto k1d = [](float sigma, float* k) {
    float sum = 0;
    for (int i = 0; i < 3; i++) {
      const float xv = (float)(i - 1);
      k[i] = expf(-(xv * xv) / (2.0f * sigma * sigma));
      sum += k[i];
    }
    for (int i = 0; i < 3; i++) k[i] /= sum;
  };
  k1d(sy, kk.ky);
  k1d(sx, kk.kx);
  kk.color_scale = -0.5f / (sigma_color * sigma_color);
  hipLaunchKernelGGL(derive_normal_fused_kernel, dim3((W + kDnW - 1) / kDnW, (H + kDnH - 1) / kDnH), dim3(kDnThreads), 0, s, W, H, fx,
                     fy, viewmatrix, kk, depth_raw, normal_out, pos_filter_out);
}

// diagnostic: fast shared-reciprocal division vs the compiler's IEEE division
__global__ void __launch_bounds__(256)
selftest_div2_kernel(int n, const float* __restrict__ nx, const float* __restrict__ ny,
                     const float* __restrict__ d, float* __restrict__ out_fast, float* __restrict__ out_ref,
                     int* __restrict__ out_round) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float big = fmaxf(fmaxf(fabsf(nx[i]), fabsf(ny[i])), fabsf(d[i]));
  const float small = fminf(fabsf(nx[i]), fabsf(ny[i]));
  // numerators below 2^-60 are exercised only through the march (absorbed by + cx), not here
  const f32x2 q = div2_exact(f32x2{nx[i], ny[i]}, d[i], big < 0x1p60f && small > 0x1p-60f);
  out_fast[2 * i] = q.x;
  out_fast[2 * i + 1] = q.y;
  out_ref[2 * i] = nx[i] / d[i];
  out_ref[2 * i + 1] = ny[i] / d[i];
  out_round[2 * i] = round_to_int(nx[i]);
  out_round[2 * i + 1] = f2i(roundf(nx[i]));
}

// exhaustive: every fp32 bit pattern; counts the t for which round_pix and (int)roundf would decide differently
__global__ void __launch_bounds__(256) selftest_round_kernel(unsigned long long* __restrict__ bad) {
  unsigned long long local = 0;
  for (unsigned long long b = (unsigned long long)blockIdx.x * 256 + threadIdx.x; b < (1ull << 32);
       b += (unsigned long long)gridDim.x * 256) {
    const float t = __uint_as_float((unsigned)b);
    const int a = f2i(roundf(t)), r = round_pix(t);
    const bool same = a == r || (a < 0 && r < 0) || (a >= 32767 && r >= 32767);
    local += same ? 0 : 1;
  }
  if (local) atomicAdd(bad, local);
}

}  // namespace gigs

extern "C" int gigs_selftest_round(unsigned long long* mismatches, void* stream) {
  if (!mismatches) return -1;
  if (hipMemsetAsync(mismatches, 0, sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess) return -2;
  hipLaunchKernelGGL(gigs::selftest_round_kernel, dim3(16384), dim3(256), 0, (hipStream_t)stream, mismatches);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int gigs_selftest_div2(int n, const float* nx, const float* ny, const float* d, float* out_fast,
                                  float* out_ref, int* out_round, void* stream) {
  if (n <= 0 || !nx || !ny || !d || !out_fast || !out_ref || !out_round) return -1;
  hipLaunchKernelGGL(gigs::selftest_div2_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, nx, ny, d,
                     out_fast, out_ref, out_round);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
