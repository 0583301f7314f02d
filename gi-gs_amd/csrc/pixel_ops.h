// pixel_ops.h -- small per-pixel device helpers shared by the filter, shading and stage-2 kernels.
#pragma once
#include "gigs_common.h"

namespace gigs {

__device__ __forceinline__ void cswap(float& a, float& b) {
  const float lo = fminf(a, b), hi = fmaxf(a, b);
  a = lo;
  b = hi;
}
// median of 9 by the classic 19-exchange network (NaN-free inputs)
__device__ __forceinline__ float median9(float* v) {
  cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
  cswap(v[0], v[1]); cswap(v[3], v[4]); cswap(v[6], v[7]);
  cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
  cswap(v[0], v[3]); cswap(v[5], v[8]); cswap(v[4], v[7]);
  cswap(v[3], v[6]); cswap(v[1], v[4]); cswap(v[2], v[5]);
  cswap(v[4], v[7]); cswap(v[4], v[2]); cswap(v[6], v[4]);
  cswap(v[4], v[2]);
  return v[4];
}

// linear -> sRGB and its derivative (train.py:54-68 / pbr/shade.py:50-63)
__device__ __forceinline__ float lin2srgb(float x, float& d) {  // pbr/shade.py:50-63
  const float eps = 1.1920929e-07f;
  if (x <= 0.0031308f) { d = 323.0f / 25.0f; return 323.0f / 25.0f * x; }
  const float c = fmaxf(x, eps);
  const float p = powf(c, 5.0f / 12.0f);
  d = x >= eps ? 211.0f * (5.0f / 12.0f) * p / c / 200.0f : 0.0f;
  return (211.0f * p - 11.0f) / 200.0f;
}
// sRGB -> linear (train.py:70-81)
__device__ __forceinline__ float srgb2lin(float x) {
  if (x <= 0.04045f) return 25.0f / 323.0f * x;
  return powf((x + 0.055f) / 1.055f, 2.4f);
}

}  // namespace gigs
