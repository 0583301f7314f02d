// blend.hip -- per-tile alpha blending of the G-buffer (forward) and its gradient (backward).
//
// Reference behaviour restated (R/ = submodules/diff-gaussian-rasterization):
//   forward : renderCUDA   R/cuda_rasterizer/forward.cu:423-633
//   backward: renderCUDA   R/cuda_rasterizer/backward.cu:404-630
// Per pixel the sequence of fp32 operations is the reference's (same skip tests
// `power > 0`, `alpha < 1/255`, `T*(1-alpha) < 1e-4`, same recurrences), so n_contrib /
// final_T and every plane match the CPU oracle up to the ulp difference of expf.
//
// MI355X design (DESIGN.md section 5 has the measurements)
//   * one 256-lane workgroup per 16x16 tile (the tile is fixed by `ranges`), four wave64s, each owning an 8x8
//     pixel quadrant -- and each wave walks the tile's instance list ON ITS OWN: there is no workgroup barrier in
//     either kernel.  Tile lists are very skewed, so a kernel lasts as long as the busiest quadrant of its
//     longest tile; autonomous waves make that max_quadrant(sum over chunks) instead of sum(max_quadrant);
//   * the list is consumed in chunks of 64 instances (one per lane) through a two-deep register pipeline (ids two
//     chunks ahead, the packed 80-byte record `brec` -- five 16-byte loads, written by the preprocess kernel -- one
//     chunk ahead) and staged in a 5 KB per-wave LDS slab for the broadcast reads of the walk;
//   * quadrant cull, instance-parallel: the 64 lanes test 64 DIFFERENT Gaussians against the wave's pixel box
//     (minimum of the convex exponent over the box vs log(255 * opacity)), a ballot turns the survivors into an
//     SGPR mask and the wave iterates set bits in list order, so the blend order and every output bit are
//     unchanged; the forward records per instance which quadrants really blended it (`hit_mask`), the backward
//     walks exactly those pairs;
//   * backward: the 64 pixels' partial gradients of one Gaussian are summed with DPP row/bank reductions
//     (registers only, single v_add_f32_dpp instructions), lane 63 drops the 19 sums into 80 bytes of LDS and
//     lanes 0-19 add them to the Gaussian's packed gradient record `grec` with ONE global_atomic_add_f32
//     wave-instruction (consecutive lanes -> consecutive floats of one row, the shape the float-atomic path runs
//     at full rate); gradient groups that are zero over the quadrant are skipped.  This replaces the reference's
//     21 same-address atomics per (pixel, Gaussian) pair.
//   No MFMA: the per-pair work is a scalar recurrence over depth-ordered Gaussians.
#include <cstdlib>
#include "gigs_common.h"

namespace gigs {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Lane mask of a predicate.  HIP's ballot64(int) goes through a 0/1 VGPR and a compare (v_cndmask + v_cmp_ne per
// call); the builtin takes the predicate's own mask -- it matters in the walk, where every instruction of the one
// wave that holds up the kernel counts.
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

constexpr int kLongTile = 2048;  // lists longer than this run at raised wave priority

// ---- DPP helpers ---------------------------------------------------------------------------
template <int kCtrl, int kRowMask = 0xf, int kBankMask = 0xf>
__device__ __forceinline__ float dpp_mov0(float v) {
  // lanes without a valid source (or masked rows) read 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, kRowMask, kBankMask, false));
}
// Sums over the 64 lanes of a wave of N values at once; the totals are valid in lane 63.  Each value takes six
// v_add_f32_dpp (source lane permuted by the DPP modifier, lanes without a source add 0, masked rows keep their
// value); they are written as instructions because the compiler turns the builtin form into v_mov_b32_dpp +
// v_add_f32 pairs -- twice the work on the walk's serial path -- and the N chains are interleaved so that the two
// wait states a DPP read needs after a VALU write of the same register are filled with useful instructions (the
// leading s_nop 4 covers what may sit right before the block -- a VALU write of an operand needs 2 wait states
// before a DPP read, a VALU write of EXEC (v_cmpx) 5: the compiler's hazard recogniser does not look into inline
// assembly).
template <int N>
__device__ __forceinline__ void wave_sum_n(float* v);
template <>
__device__ __forceinline__ void wave_sum_n<1>(float* v) {
  asm("s_nop 4\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v[0]));
}
template <>
__device__ __forceinline__ void wave_sum_n<3>(float* v) {
  asm("s_nop 4\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
}
template <>
__device__ __forceinline__ void wave_sum_n<5>(float* v) {
  asm("s_nop 4\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]));
}
template <>
__device__ __forceinline__ void wave_sum_n<10>(float* v) {
  asm("s_nop 4\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %9, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %9, %9, %9 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %5, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %6, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %7, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %8, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %9, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %9, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %9, %9 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %9, %9 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
  wave_sum_n<1>(&v);
  return v;
}

// Conservative test "no pixel centre of a box of pixels can pass `alpha >= 1/255`" for one Gaussian.
// d = mean2D - pix ranges over [ua, ub] x [va, vb] (the box is the bounding box of the quadrant's pixels that are
// still blending, see the forward kernel); the exponent -power = Q(d) = 0.5 (A dx^2 + C dy^2) + B dx dy is convex when the
// conic is positive definite, and its minimum over the box is 0 (centre inside) or lies on one of the four
// edges.  alpha >= 1/255 needs Q <= log(255 * opacity); the comparison carries a slack far above the fp32
// error of both this bound and the kernel's own evaluation of `power`, so a culled Gaussian is one that
// the reference's per-pixel tests (forward.cu:533-541) would have skipped for every pixel of the quadrant.
__device__ __forceinline__ bool box_never_blends(float ua, float ub, float va, float vb, float A, float B, float C, float op) {
  if (op < 1.0f / 255.0f) return true;  // alpha <= opacity * exp(power <= 0)
  if (!(A > 0.0f && C > 0.0f && A * C - B * B > 0.0f)) return false;
  if (ua <= 0.0f && ub >= 0.0f && va <= 0.0f && vb >= 0.0f) return false;
  const float iA = __builtin_amdgcn_rcpf(A), iC = __builtin_amdgcn_rcpf(C);
  auto f = [&](float u, float v) { return 0.5f * (A * u * u + C * v * v) + B * u * v; };
  const float q1 = f(ua, fminf(fmaxf(-B * ua * iC, va), vb));
  const float q2 = f(ub, fminf(fmaxf(-B * ub * iC, va), vb));
  const float q3 = f(fminf(fmaxf(-B * va * iA, ua), ub), va);
  const float q4 = f(fminf(fmaxf(-B * vb * iA, ua), ub), vb);
  const float qmin = fminf(fminf(q1, q2), fminf(q3, q4));
  const float um = fmaxf(fabsf(ua), fabsf(ub)), vm = fmaxf(fabsf(va), fabsf(vb));
  const float smax = 0.5f * (A * um * um + C * vm * vm) + fabsf(B) * um * vm;
  const float tau = __logf(255.0f * op);
  return qmin > tau + 1e-5f * smax + 1e-2f;  // false for NaN
}

// exp(x) for the blend exponent (x <= 0 wherever the value is used; |x| < 6 wherever alpha passes 1/255).
// x * log2(e) is formed as an fp32 product plus its exact FMA residual and the low word of the constant, the
// hardware exp2 (v_exp_f32, ~1 ulp) takes the high part and the low part is applied as e * (1 + lo * ln 2):
// 6 instructions and <= 2 ulp, against ~18 for OCML's expf (1 ulp) -- the same accuracy class as the
// reference's CUDA expf (2 ulp).  The forward and the backward must (and do) use the SAME function: the
// backward re-derives the forward's alpha tests bit for bit.
__device__ __forceinline__ float blend_exp(float x) {
  const float L = 0x1.715476p+0f, L_lo = 0x1.4ae0cp-26f;  // log2(e) = L + L_lo
  const float y = x * L;
  const float lo = __builtin_fmaf(x, L_lo, __builtin_fmaf(x, L, -y));
  const float e = __builtin_amdgcn_exp2f(y);
  return __builtin_fmaf(e, lo * 0x1.62e43p-1f, e);
}

struct BlendOut {
  float *color, *opacity, *depth, *normal, *normal_view, *pos, *albedo, *roughness, *metallic;
};

__global__ void __launch_bounds__(GIGS_TILE)
blend_fwd_kernel(int W, int H, unsigned gx, const uint2* __restrict__ ranges,
                 const uint32_t* __restrict__ point_list, const float4* __restrict__ brec,
                 const float* __restrict__ viewmatrix, const float* __restrict__ bg_color,
                 uint32_t* __restrict__ n_contrib, float* __restrict__ final_T, BlendOut o,
                 int argmax_depth, int inference, uint8_t* __restrict__ hit_mask, size_t hit_stride, int cull,
                 const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ reuse_guard, uint2 reuse_expect) {
  // Each wave walks the tile's list on its own (no workgroup barrier anywhere): the time of a tile is the
  // time of its busiest quadrant, not the sum over batches of the slowest quadrant of each batch, and a wave
  // whose 64 pixels are saturated leaves at once.  The list is consumed in chunks of 64 instances, one per
  // lane, through a two-deep register pipeline (ids two chunks ahead, the 80-byte records one chunk ahead),
  // so the dependent index -> record gathers of the next chunk are in flight while this one is blended.
  __shared__ float4 s_rec[4][GIGS_BREC_F4 * 64];  // per wave, [k][slot]: 4 x 5 KB

  const unsigned tile = tile_order[blockIdx.x];  // longest lists first (binning.hip::tile_order_kernel)
  const unsigned ty = tile / gx, tx = tile - ty * gx;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const unsigned px = tx * GIGS_BLOCK_X + (wave & 1) * 8 + (lane & 7);
  const unsigned py = ty * GIGS_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < (unsigned)W && py < (unsigned)H;
  const size_t pix_id = (size_t)W * py + px;
  float pixfx = (float)px, pixfy = (float)py;
  asm volatile("" : "+v"(pixfx), "+v"(pixfy));  // not to be rematerialised (2 v_cvt per survivor) inside the walk
  bool done = !inside;

  uint2 range = ranges[tile];
  // reused lists (gigs_ctx_set_reuse_binning): blended only if they were made for this Gaussian count and capacity --
  // otherwise the tile is empty (background), and no stale index is ever dereferenced
  if (reuse_guard && (reuse_guard[2] != reuse_expect.x || reuse_guard[3] != reuse_expect.y || range.y > reuse_expect.y ||
                      range.x > range.y))
    range = make_uint2(0u, 0u);
  const int n = (int)(range.y - range.x);
  // the kernel ends when its longest tile does: waves of long tiles get the issue priority on their SIMD
  if (n > kLongTile) __builtin_amdgcn_s_setprio(3);
  float4* sw = s_rec[wave];
  // this quadrant's byte of instance i of the list: hit[i].  One plane of R bytes per quadrant ([4][R]): a wave's 64 bytes per
  // chunk are one contiguous run (interleaved [R][4] they were a 256-byte span of partial 32-byte writes: 16 B of L2 -> fabric
  // write traffic per instance at C4, 460 MB per forward; the backward read them with the same stride)
  uint8_t* hit = hit_mask + (size_t)wave * hit_stride + range.x;

  float T = 1.0f;
  uint32_t last_contributor = 0;
  // Accumulators paired the way the record words are laid out, so that every `acc += word * weight` (a multiply
  // and an add: the library is built without FMA contraction) is one v_pk_mul_f32 + one v_pk_add_f32 for two
  // planes -- same IEEE operations per element, half the instructions on the walk's serial path.
  f32x2 C01 = {0, 0}, C2P0 = {0, 0}, N01 = {0, 0}, N2P1 = {0, 0}, A01 = {0, 0}, A2P2 = {0, 0}, RM = {0, 0};
  float O = 0;  // P2 (pos_view.z) doubles as the depth accumulator
  float max_weight = 0.0f, e0 = 0, e1 = 0, e2 = 0;

  // pixel-centre box of this wave's quadrant
  const float qx0 = (float)(tx * GIGS_BLOCK_X + (wave & 1) * 8), qy0 = (float)(ty * GIGS_BLOCK_Y + (wave >> 1) * 8);

  // pipeline prologue: records of chunk 0, ids of chunk 1
  float4 rec[GIGS_BREC_F4];
#pragma unroll
  for (int k = 0; k < GIGS_BREC_F4; k++) rec[k] = make_float4(0, 0, 0, 0);
  if (lane < n) {
    const float4* src = brec + (size_t)point_list[range.x + lane] * GIGS_BREC_F4;
#pragma unroll
    for (int k = 0; k < GIGS_BREC_F4; k++) rec[k] = src[k];
  }
  uint32_t id_next = (64 + lane < n) ? point_list[range.x + 64 + lane] : 0u;

  int base = 0;
  bool wave_done = ballot64(!done) == 0ull;
  for (; base < n && !wave_done; base += 64) {
    const bool valid = base + lane < n;
    // this chunk: stage the records for the broadcast reads of the walk, keep mean/conic/opacity for the cull
    const float4 q0 = rec[0], q1 = rec[1];
#pragma unroll
    for (int k = 0; k < GIGS_BREC_F4; k++) sw[k * 64 + lane] = rec[k];
    // next chunk's records and the ids after that: in flight while this chunk is walked
    if (base + 64 + lane < n) {
      const float4* src = brec + (size_t)id_next * GIGS_BREC_F4;
#pragma unroll
      for (int k = 0; k < GIGS_BREC_F4; k++) rec[k] = src[k];
    }
    id_next = (base + 128 + lane < n) ? point_list[range.x + base + 128 + lane] : 0u;

    // instance-parallel cull: lane l tests instance base+l against the quadrant, the wave then walks only
    // the surviving bits (in list order, so the blend order is unchanged)
    // The box is the bounding box of the pixels of the quadrant that are still blending (not saturated, inside the
    // image): a quadrant on a silhouette keeps walking for its few open pixels only, and the Gaussians that cover
    // just its saturated part fall to the cull instead of being evaluated.  Scalar bit arithmetic on one ballot.
    const unsigned long long open_px = ballot64(!done);  // bit = lane = 8 * row + column; non-zero here
    unsigned long long fold = open_px | (open_px >> 32);
    fold |= fold >> 16;
    fold |= fold >> 8;
    const unsigned cols = (unsigned)fold & 0xffu;
    const float bx0 = qx0 + (float)__builtin_ctz(cols), bx1 = qx0 + (float)(31 - __builtin_clz(cols));
    const float by0 = qy0 + (float)(__builtin_ctzll(open_px) >> 3), by1 = qy0 + (float)((63 - __builtin_clzll(open_px)) >> 3);
    unsigned long long m = ballot64(
        valid && !(cull && box_never_blends(q0.x - bx1, q0.x - bx0, q0.y - by1, q0.y - by0, q1.x, q1.y, q1.z, q1.w)));
    unsigned long long hits = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    while (m != 0ull) {
      const int bit = __builtin_ctzll(m);
      m &= m - 1ull;
      const float4 r0 = sw[bit];       // mean2D.xy, roughness, metallic
      const float4 r1 = sw[64 + bit];  // conic xyz, opacity
      const float dx = r0.x - pixfx, dy = r0.y - pixfy;
      const float power = -0.5f * (r1.x * dx * dx + r1.z * dy * dy) - r1.y * dx * dy;
      // same decisions as the reference's `continue` chain (forward.cu:533-547), kept as predicates so
      // that the wave can record whether ANY of its pixels blends this Gaussian
      const float alpha = fminf(0.99f, r1.w * blend_exp(power));
      const float test_T = T * (1 - alpha);
      const bool cand = !done && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
      const bool stop = cand && test_T < 0.0001f;
      const bool contrib = cand && !stop;
      if (stop) done = true;
      if (ballot64(contrib) != 0ull) hits |= 1ull << bit;
      if (contrib) {
        const float weight = alpha * T;
        const float4 r2 = sw[2 * 64 + bit];  // rgb, pos.x
        const float4 r3 = sw[3 * 64 + bit];  // normal, pos.y
        const float4 r4 = sw[4 * 64 + bit];  // albedo, pos.z
        const f32x2 w2 = {weight, weight};
        C01 += f32x2{r2.x, r2.y} * w2; C2P0 += f32x2{r2.z, r2.w} * w2;
        N01 += f32x2{r3.x, r3.y} * w2; N2P1 += f32x2{r3.z, r3.w} * w2;
        A01 += f32x2{r4.x, r4.y} * w2; A2P2 += f32x2{r4.z, r4.w} * w2;
        RM += f32x2{r0.z, r0.w} * w2;
        O += weight;
        if (argmax_depth) {  // wave-uniform: only the argmax_depth outputs read e0..e2
          if (weight > max_weight) {
            e0 = r2.w; e1 = r3.w; e2 = r4.w;
            max_weight = weight;
          }
        }
        T = test_T;
        last_contributor = (uint32_t)(base + bit + 1);
      }
    }
    // "every pixel of the quadrant saturated" is looked at once per chunk, not per survivor: two ballots less on the
    // serial path of the long walks, at most one chunk of cheap non-blending evaluations more for the others
    wave_done = ballot64(!done) == 0ull;
    __builtin_amdgcn_wave_barrier();  // the walk's LDS reads precede the next chunk's staging stores
    if (valid) hit[base + lane] = (uint8_t)((hits >> lane) & 1ull);
  }
  // instances of chunks that were never fetched (every pixel of the quadrant saturated) blend nowhere here
  for (int i = base + lane; i < n; i += 64) hit[i] = 0;

  const float C0 = C01.x, C1 = C01.y, C2 = C2P0.x, P0 = C2P0.y, N0 = N01.x, N1 = N01.y, N2 = N2P1.x, P1 = N2P1.y;
  const float A0 = A01.x, A1 = A01.y, A2 = A2P2.x, P2 = A2P2.y, Rr = RM.x, Mm = RM.y;
  if (inside) {
    const size_t HW = (size_t)H * W;
    final_T[pix_id] = T;
    n_contrib[pix_id] = last_contributor;
    const v3 nv = normalize3(xform_vec_4x3({N0, N1, N2}, viewmatrix));  // NaN when N == 0
    o.normal_view[pix_id] = nv.x;
    o.normal_view[HW + pix_id] = nv.y;
    o.normal_view[2 * HW + pix_id] = nv.z;
    o.color[pix_id] = C0 + T * bg_color[0];
    o.color[HW + pix_id] = C1 + T * bg_color[1];
    o.color[2 * HW + pix_id] = C2 + T * bg_color[2];
    o.normal[pix_id] = N0; o.normal[HW + pix_id] = N1; o.normal[2 * HW + pix_id] = N2;
    o.albedo[pix_id] = A0; o.albedo[HW + pix_id] = A1; o.albedo[2 * HW + pix_id] = A2;
    o.roughness[pix_id] = inference ? (Rr + T) : Rr;
    o.metallic[pix_id] = Mm;
    if ((double)O > 1e-6) {
      o.depth[pix_id] = argmax_depth ? e2 : P2 / O;
      o.pos[pix_id] = argmax_depth ? e0 : P0 / O;
      o.pos[HW + pix_id] = argmax_depth ? e1 : P1 / O;
      o.pos[2 * HW + pix_id] = argmax_depth ? e2 : P2 / O;
    } else {
      o.depth[pix_id] = 0.0f;
      o.pos[pix_id] = 0.0f;
      o.pos[HW + pix_id] = 0.0f;
      o.pos[2 * HW + pix_id] = 0.0f;
    }
    o.opacity[pix_id] = O;
  }
}

void launch_blend_fwd(const FwdArgs& a, const GeomState& g, const BinningState& b,
                      const ImageState& im, float* out_color, float* out_opacity,
                      float* out_depth, float* out_normal, float* out_normal_view, float* out_pos,
                      float* out_albedo, float* out_roughness, float* out_metallic, int cull, size_t hit_stride,
                      hipStream_t s, bool reused_lists) {
  BlendOut o{out_color, out_opacity, out_depth, out_normal, out_normal_view,
             out_pos, out_albedo, out_roughness, out_metallic};
  // cull = 0 (gigs_options.blend_cull) disables the quadrant cull (diagnostic: the outputs must not change by a bit)
  hipLaunchKernelGGL(blend_fwd_kernel, dim3(a.gx * a.gy), dim3(GIGS_TILE), 0, s, a.W, a.H, a.gx,
                     im.ranges, b.point_list, g.brec, a.viewmatrix, a.background, im.n_contrib,
                     im.final_T, o, a.argmax_depth, a.inference, b.hit_mask, hit_stride, cull, im.tile_order,
                     reused_lists ? (const uint32_t*)im.bin_counters : (const uint32_t*)nullptr,
                     make_uint2((unsigned)a.P, (unsigned)hit_stride));
}

// ------------------------------------------------------------------------------------------
struct BlendGradIn {
  const float *depth, *color, *opacity, *normal, *albedo, *roughness, *metallic;
};

__global__ void __launch_bounds__(GIGS_TILE)
blend_bwd_kernel(int W, int H, unsigned gx, const uint2* __restrict__ ranges,
                 const uint32_t* __restrict__ point_list, const float4* __restrict__ brec,
                 const float* __restrict__ bg_color, const float* __restrict__ final_Ts,
                 const uint32_t* __restrict__ n_contrib, BlendGradIn gi, float* __restrict__ grec,
                 const uint8_t* __restrict__ hit_mask, size_t hit_stride, const uint32_t* __restrict__ tile_order) {
  // Wave-autonomous like the forward: each wave walks the tile's list back to front in chunks of 64 instances
  // (lane l of chunk c <-> instance n-1-(64c+l)), fetching only the instances its quadrant blended (the
  // forward's hit byte), two-deep pipelined, no workgroup barrier.  The 19 per-Gaussian sums of a wave are
  // reduced with DPP, bounced through 80 bytes of LDS so that lane k holds sum k, and leave as ONE
  // global_atomic_add_f32 wave-instruction over the contiguous 80-byte gradient record (non-zero lanes only).
  __shared__ float4 s_rec[4][3 * 64];            // per wave, [k][slot], k = 0..2 (mean2D, conic/opacity, rgb)
  __shared__ __align__(16) float s_sum[4][2][GIGS_GREC];  // per wave: the reduced sums of the current instance (two when paired)

  const unsigned tile = tile_order[blockIdx.x];  // longest lists first (binning.hip::tile_order_kernel)
  const unsigned ty = tile / gx, tx = tile - ty * gx;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const unsigned px = tx * GIGS_BLOCK_X + (wave & 1) * 8 + (lane & 7);
  const unsigned py = ty * GIGS_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < (unsigned)W && py < (unsigned)H;
  const size_t pix_id = (size_t)W * py + px;
  const size_t HW = (size_t)H * W;
  const float pixfx = (float)px, pixfy = (float)py;

  const uint2 range = ranges[tile];
  const int n = (int)(range.y - range.x);
  if (n > kLongTile) __builtin_amdgcn_s_setprio(3);
  float4* sw = s_rec[wave];
  float* ssum = s_sum[wave][0];
  float* ssum2 = s_sum[wave][1];
  const uint8_t* hit = hit_mask + (size_t)wave * hit_stride + range.x;

  const float T_final = inside ? final_Ts[pix_id] : 0;
  float T = T_final;
  const int last_contributor = inside ? (int)n_contrib[pix_id] : 0;

  float last_alpha = 0.0f, accum_opacity = 0.0f;
  float ar0 = 0, ar1 = 0, ar2 = 0, lc0 = 0, lc1 = 0, lc2 = 0;
  float dp0 = 0, dp1 = 0, dp2 = 0, dn0 = 0, dn1 = 0, dn2 = 0, da0 = 0, da1 = 0, da2 = 0;
  float dop = 0, drg = 0, dmt = 0, ddp = 0;
  if (inside) {
    // a NULL plane is an all-zero gradient (an output the loss does not use: no tensor is materialised for it)
    if (gi.color) { dp0 = gi.color[pix_id]; dp1 = gi.color[HW + pix_id]; dp2 = gi.color[2 * HW + pix_id]; }
    if (gi.normal) { dn0 = gi.normal[pix_id]; dn1 = gi.normal[HW + pix_id]; dn2 = gi.normal[2 * HW + pix_id]; }
    if (gi.albedo) { da0 = gi.albedo[pix_id]; da1 = gi.albedo[HW + pix_id]; da2 = gi.albedo[2 * HW + pix_id]; }
    if (gi.opacity) dop = gi.opacity[pix_id];
    if (gi.roughness) drg = gi.roughness[pix_id];
    if (gi.metallic) dmt = gi.metallic[pix_id];
    if (gi.depth) ddp = gi.depth[pix_id];
  }
  if (px == 0 || px == (unsigned)(W - 1) || py == 0 || py == (unsigned)(H - 1)) {  // backward.cu:497-501
    dn0 = dn1 = dn2 = 0.0f;
  }
  const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);
  const float bg_dot_dpixel = 0 + bg_color[0] * dp0 + bg_color[1] * dp1 + bg_color[2] * dp2;
  // Which incoming gradient planes are non-zero anywhere in this QUADRANT?  A plane that is zero for all 64
  // pixels contributes exact zeros to every sum, so its reductions are skipped (stage-2 training feeds only
  // albedo / roughness / metallic gradients: SURVEY Appendix D).
  //   geo: colour or opacity grads -> dL_dalpha -> mean2D, conic, opacity, colour (v[0..9])
  const bool any_geo = __any((dp0 != 0.0f) | (dp1 != 0.0f) | (dp2 != 0.0f) | (dop != 0.0f)) != 0;
  const bool any_nrm = __any((dn0 != 0.0f) | (dn1 != 0.0f) | (dn2 != 0.0f)) != 0;
  const bool any_mat = __any((da0 != 0.0f) | (da1 != 0.0f) | (da2 != 0.0f) | (drg != 0.0f) | (dmt != 0.0f)) != 0;
  const bool any_dep = __any(ddp != 0.0f) != 0;
  if (!(any_geo || any_nrm || any_mat || any_dep)) return;  // every sum of this quadrant is an exact zero

  // pipeline prologue (chunk 0 = the last 64 instances): ids + hit bytes of chunks 0 and 1, records of chunk 0
  auto fetch_id = [&](int base, uint32_t& id, bool& h) {
    const int idx = n - 1 - (base + lane);
    h = idx >= 0 && hit[idx] != 0;
    id = h ? point_list[range.x + idx] : 0u;
  };
  uint32_t id_cur, id_next;
  bool h_cur, h_next;
  fetch_id(0, id_cur, h_cur);
  fetch_id(64, id_next, h_next);
  float4 rec[3];
#pragma unroll
  for (int k = 0; k < 3; k++) rec[k] = make_float4(0, 0, 0, 0);
  if (h_cur) {
    const float4* src = brec + (size_t)id_cur * GIGS_BREC_F4;
#pragma unroll
    for (int k = 0; k < 3; k++) rec[k] = src[k];
  }

  if (lane < GIGS_GREC) { ssum[lane] = 0.0f; ssum2[lane] = 0.0f; }
  // Material gradients only (every stage-2 step): the per-instance work is one reduction of five sums and one atomic
  // instruction, each a chain of dependent steps (six DPP stages with their wait states, LDS bounce, atomic) that a lone wave
  // cannot hide.  Two surviving instances are then taken per iteration: their ten sums share ONE interleaved reduction
  // (wave_sum_n<10>: the same DPP tree per value, so every sum keeps its bits), one LDS bounce and one atomic instruction
  // (lanes 0-19 the first record, lanes 32-51 the second); the T recurrence stays in list order.
  const bool mat_only = any_mat && !any_geo && !any_nrm && !any_dep;
  for (int base = 0; base < n; base += 64) {
    unsigned long long m = __ballot(h_cur);
    const uint32_t my_id = id_cur;
    if (m != 0ull) {
#pragma unroll
      for (int k = 0; k < 3; k++) sw[k * 64 + lane] = rec[k];
    }
    // next chunk's records, and the ids / hit bytes of the chunk after it
    id_cur = id_next; h_cur = h_next;
    if (h_cur) {
      const float4* src = brec + (size_t)id_cur * GIGS_BREC_F4;
#pragma unroll
      for (int k = 0; k < 3; k++) rec[k] = src[k];
    }
    fetch_id(base + 128, id_next, h_next);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    while (mat_only && (m & (m - 1ull)) != 0ull) {  // at least two survivors left in this chunk
      const int bitA = __builtin_ctzll(m);
      m &= m - 1ull;
      const int bitB = __builtin_ctzll(m);
      m &= m - 1ull;
      float dcc[2];
#pragma unroll
      for (int q = 0; q < 2; q++) {  // in list order: the second instance sees the first one's T
        const int bit = q == 0 ? bitA : bitB;
        const int contributor = n - 1 - (base + bit);
        const float4 r0 = sw[bit];
        const float4 r1 = sw[64 + bit];
        const float dx = r0.x - pixfx, dy = r0.y - pixfy;
        const float power = -0.5f * (r1.x * dx * dx + r1.z * dy * dy) - r1.y * dx * dy;
        const float G = blend_exp(power);
        const float alpha = fminf(0.99f, r1.w * G);
        const bool act = inside && (contributor < last_contributor) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
        dcc[q] = 0.0f;
        if (act) {
          T = T / (1.f - alpha);
          dcc[q] = alpha * T;
        }
      }
      float v[10] = {dcc[0] * da0, dcc[0] * da1, dcc[0] * da2, dcc[0] * drg, dcc[0] * dmt,
                     dcc[1] * da0, dcc[1] * da1, dcc[1] * da2, dcc[1] * drg, dcc[1] * dmt};
      wave_sum_n<10>(v);
      if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 5; k++) { ssum[13 + k] = v[k]; ssum2[13 + k] = v[5 + k]; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint32_t gidA = (uint32_t)__builtin_amdgcn_readlane((int)my_id, bitA);
      const uint32_t gidB = (uint32_t)__builtin_amdgcn_readlane((int)my_id, bitB);
      {
        const int k = lane & 31, second = lane >> 5;
        if (k < GIGS_GREC) {
          const float val = second ? ssum2[k] : ssum[k];
          if (val != 0.0f) atomicAdd(grec + (size_t)(second ? gidB : gidA) * GIGS_GREC + k, val);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    while (m != 0ull) {
      const int bit = __builtin_ctzll(m);
      m &= m - 1ull;
      const int contributor = n - 1 - (base + bit);  // index of the instance in the tile list
      const float4 r0 = sw[bit];
      const float4 r1 = sw[64 + bit];
      const float dx = r0.x - pixfx, dy = r0.y - pixfy;
      const float power = -0.5f * (r1.x * dx * dx + r1.z * dy * dy) - r1.y * dx * dy;
      const float G = blend_exp(power);
      const float alpha = fminf(0.99f, r1.w * G);
      const bool act = inside && (contributor < last_contributor) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);

      // Per gradient group (wave-uniform switches): the 64 per-pixel terms, their wave sum, and lane 63's store of
      // the sums into the group's slots of the 80-byte LDS row (slots of groups that never run stay zero).  A pixel
      // that does not blend this Gaussian contributes dchannel_dcolor = 0, i.e. exact zeros (finite gradients).
      float dchannel_dcolor = 0.0f, one_minus_alpha = 1.0f;
      if (act) {
        one_minus_alpha = 1.f - alpha;
        T = T / one_minus_alpha;
        dchannel_dcolor = alpha * T;
      }
      if (any_mat) {
        float m[5] = {dchannel_dcolor * da0, dchannel_dcolor * da1, dchannel_dcolor * da2, dchannel_dcolor * drg,
                      dchannel_dcolor * dmt};
        wave_sum_n<5>(m);
        if (lane == 63) {
#pragma unroll
          for (int k = 0; k < 5; k++) ssum[13 + k] = m[k];
        }
      }
      if (any_nrm) {
        float m[3] = {dchannel_dcolor * dn0, dchannel_dcolor * dn1, dchannel_dcolor * dn2};
        wave_sum_n<3>(m);
        if (lane == 63) {
#pragma unroll
          for (int k = 0; k < 3; k++) ssum[10 + k] = m[k];
        }
      }
      if (any_dep) {
        float m = dchannel_dcolor * ddp;
        wave_sum_n<1>(&m);
        if (lane == 63) ssum[18] = m;
      }
      if (any_geo) {  // the dL_dalpha chain (and its running colour / opacity state) feeds sums 0..9 only
        float v[10];
#pragma unroll
        for (int k = 0; k < 10; k++) v[k] = 0.0f;
        if (act) {
          const float4 r2 = sw[2 * 64 + bit];
          float dL_dalpha = 0.0f;
          ar0 = last_alpha * lc0 + (1.f - last_alpha) * ar0; lc0 = r2.x;
          dL_dalpha += (r2.x - ar0) * dp0;
          ar1 = last_alpha * lc1 + (1.f - last_alpha) * ar1; lc1 = r2.y;
          dL_dalpha += (r2.y - ar1) * dp1;
          ar2 = last_alpha * lc2 + (1.f - last_alpha) * ar2; lc2 = r2.z;
          dL_dalpha += (r2.z - ar2) * dp2;
          v[7] = dchannel_dcolor * dp0; v[8] = dchannel_dcolor * dp1; v[9] = dchannel_dcolor * dp2;
          accum_opacity = last_alpha + (1.f - last_alpha) * accum_opacity;
          dL_dalpha += (1.0f - accum_opacity) * dop;
          dL_dalpha *= T;
          last_alpha = alpha;
          dL_dalpha += (-T_final / one_minus_alpha) * bg_dot_dpixel;
          const float dL_dG = r1.w * dL_dalpha;
          const float gdx = G * dx, gdy = G * dy;
          const float dG_ddelx = -gdx * r1.x - gdy * r1.y;
          const float dG_ddely = -gdy * r1.z - gdx * r1.y;
          v[0] = dL_dG * dG_ddelx * ddelx_dx;
          v[1] = dL_dG * dG_ddely * ddely_dy;
          v[2] = fabsf(v[0]) + fabsf(v[1]);
          v[3] = -0.5f * gdx * dx * dL_dG;
          v[4] = -0.5f * gdx * dy * dL_dG;
          v[5] = -0.5f * gdy * dy * dL_dG;
          v[6] = G * dL_dalpha;
        }
        wave_sum_n<10>(v);
        if (lane == 63) {
#pragma unroll
          for (int k = 0; k < 10; k++) ssum[k] = v[k];
        }
      }
      // lane k then adds sum k to float k of the Gaussian's gradient record
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint32_t gid = (uint32_t)__builtin_amdgcn_readlane((int)my_id, bit);
      if (lane < GIGS_GREC) {
        const float val = ssum[lane];
        if (val != 0.0f) atomicAdd(grec + (size_t)gid * GIGS_GREC + lane, val);
      }
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();  // the walk's LDS reads precede the next chunk's staging stores
  }
}

void launch_blend_bwd(const BwdArgs& a, const GeomState& g, const BinningState& b,
                      const ImageState& im, hipStream_t s) {
  BlendGradIn gi{a.dL_dpix_depth, a.dL_dpix, a.dL_dpix_opacity, a.dL_dpix_normal,
                 a.dL_dpix_albedo, a.dL_dpix_roughness, a.dL_dpix_metallic};
  hipLaunchKernelGGL(blend_bwd_kernel, dim3(a.gx * a.gy), dim3(GIGS_TILE), 0, s, a.W, a.H, a.gx,
                     im.ranges, b.point_list, g.brec, a.background, im.final_T,
                     im.n_contrib, gi, g.grec, b.hit_mask, (size_t)a.R, im.tile_order);
}

}  // namespace gigs
