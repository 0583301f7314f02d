// knn.hip -- SURVEY 8(f) rank 4: simple-knn's distCUDA2 (submodules/simple-knn/simple_knn.cu:165-224), the mean squared
// distance of every point to its three nearest neighbours, used once to initialise the Gaussian scales
// (scene/gaussian_model.py:277-281).
//
// The answer is defined exactly (3 nearest neighbours by index-exclusion, squared Euclidean distance in fp32), so the
// search structure is free.  The reference sorts by a 30-bit Morton code, cuts the order into boxes of 1024 points and
// lets every thread walk all boxes on its own (divergent, indirect `points[indices[i]]` loads).  Here:
//   * the sorted points are materialised once as float4 (xyz + original index): every later load is coalesced;
//   * boxes are one wavefront wide (64 points) under super-boxes of 64 boxes: two pruning levels instead of one;
//   * a wave owns one box and searches cooperatively: a candidate box is visited if ANY lane still needs it
//     (ballot), its 64 points are loaded once (one per lane) and broadcast lane by lane, so control flow is uniform
//     and the arithmetic is 64 lanes wide;
//   * each lane's bound starts from its own box and only shrinks (own super-box first).
// All stages are queued on one stream with no host read-back (the reference copies the bounding box to the host twice).
#include <cfloat>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/gigs_hip.h"
#include "gigs_common.h"

namespace gigs {

constexpr int kKnnBox = 64;
constexpr int kKnnSuper = 64;  // boxes per super-box

struct KnnBounds { float lo[3], hi[3]; };

// monotone float <-> uint mapping for atomicMin / atomicMax
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__global__ void knn_init_kernel(unsigned* ext) {
  if (threadIdx.x < 3) ext[threadIdx.x] = 0xffffffffu;       // running min (ordered)
  else if (threadIdx.x < 6) ext[threadIdx.x] = 0u;           // running max
}

__global__ void __launch_bounds__(256)
knn_extent_kernel(int P, const float* __restrict__ pts, unsigned* __restrict__ ext) {
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float v = pts[3 * (size_t)i + c];
      lo[c] = fminf(lo[c], v);
      hi[c] = fmaxf(hi[c], v);
    }
#pragma unroll
  for (int c = 0; c < 3; c++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[c] = fminf(lo[c], __shfl_xor(lo[c], off));
      hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], off));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(ext + c, f2ord(lo[c]));
      atomicMax(ext + 3 + c, f2ord(hi[c]));
    }
  }
}

// simple_knn.cu:42-57
__device__ __forceinline__ unsigned prep_morton(unsigned x) {
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}

__global__ void __launch_bounds__(256)
knn_morton_kernel(int P, const float* __restrict__ pts, const unsigned* __restrict__ ext, unsigned* __restrict__ codes,
                  unsigned* __restrict__ index) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  unsigned code = 0;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float lo = ord2f(ext[c]), hi = ord2f(ext[3 + c]);
    const float span = hi - lo;
    float t = span > 0.0f ? (pts[3 * (size_t)i + c] - lo) / span : 0.0f;
    t = fminf(fmaxf(t, 0.0f), 1.0f);  // NaN coordinates sort first instead of producing an undefined conversion
    code |= prep_morton((unsigned)(t * 1023.0f)) << c;
  }
  codes[i] = code;
  index[i] = (unsigned)i;
}

// sorted points as float4 (w = original index bits) + the bounds of every 64-point box; one wave per box
__global__ void __launch_bounds__(256)
knn_gather_kernel(int P, const float* __restrict__ pts, const unsigned* __restrict__ order, float4* __restrict__ sorted,
                  KnnBounds* __restrict__ boxes) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (i < P) {
    const unsigned src = order[i];
    const float x = pts[3 * (size_t)src], y = pts[3 * (size_t)src + 1], z = pts[3 * (size_t)src + 2];
    sorted[i] = make_float4(x, y, z, __uint_as_float(src));
    lo[0] = hi[0] = x; lo[1] = hi[1] = y; lo[2] = hi[2] = z;
  }
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[c] = fminf(lo[c], __shfl_xor(lo[c], off));
      hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], off));
    }
  const int box = i >> 6;
  if ((threadIdx.x & 63) == 0 && box * kKnnBox < P) {
    KnnBounds b;
#pragma unroll
    for (int c = 0; c < 3; c++) { b.lo[c] = lo[c]; b.hi[c] = hi[c]; }
    boxes[box] = b;
  }
}

// bounds of every super-box (64 boxes); one wave per super-box
__global__ void __launch_bounds__(256)
knn_super_kernel(int n_boxes, const KnnBounds* __restrict__ boxes, KnnBounds* __restrict__ supers) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (b < n_boxes) {
#pragma unroll
    for (int c = 0; c < 3; c++) { lo[c] = boxes[b].lo[c]; hi[c] = boxes[b].hi[c]; }
  }
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[c] = fminf(lo[c], __shfl_xor(lo[c], off));
      hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], off));
    }
  const int s = b >> 6;
  if ((threadIdx.x & 63) == 0 && s * kKnnSuper < n_boxes) {
    KnnBounds o;
#pragma unroll
    for (int c = 0; c < 3; c++) { o.lo[c] = lo[c]; o.hi[c] = hi[c]; }
    supers[s] = o;
  }
}

// simple_knn.cu:106-115
__device__ __forceinline__ float dist_box_point(const KnnBounds& box, float x, float y, float z) {
  float dx = 0.0f, dy = 0.0f, dz = 0.0f;
  if (x < box.lo[0] || x > box.hi[0]) dx = fminf(fabsf(x - box.lo[0]), fabsf(x - box.hi[0]));
  if (y < box.lo[1] || y > box.hi[1]) dy = fminf(fabsf(y - box.lo[1]), fabsf(y - box.hi[1]));
  if (z < box.lo[2] || z > box.hi[2]) dz = fminf(fabsf(z - box.lo[2]), fabsf(z - box.hi[2]));
  return dx * dx + dy * dy + dz * dz;
}

// simple_knn.cu:117-129 (K = 3)
__device__ __forceinline__ void update_best(float dist, float& b0, float& b1, float& b2) {
  if (b0 > dist) { const float t = b0; b0 = dist; dist = t; }
  if (b1 > dist) { const float t = b1; b1 = dist; dist = t; }
  if (b2 > dist) { b2 = dist; }
}

// every lane measures its point against the `count` points of one box, broadcast lane by lane
__device__ __forceinline__ void visit_box(const float4* __restrict__ sorted, int box, int P, int self, float x, float y,
                                          float z, float& b0, float& b1, float& b2) {
  const int lane = threadIdx.x & 63;
  const int first = box * kKnnBox;
  const int count = min(kKnnBox, P - first);
  float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (lane < count) q = sorted[first + lane];
  for (int j = 0; j < count; j++) {
    const float qx = __shfl(q.x, j), qy = __shfl(q.y, j), qz = __shfl(q.z, j);
    const float dx = qx - x, dy = qy - y, dz = qz - z;
    const float d = dx * dx + dy * dy + dz * dz;
    if (first + j != self) update_best(d, b0, b1, b2);
  }
}

__global__ void __launch_bounds__(256)
knn_search_kernel(int P, int n_boxes, int n_supers, const float4* __restrict__ sorted,
                  const KnnBounds* __restrict__ boxes, const KnnBounds* __restrict__ supers,
                  float* __restrict__ mean_dists) {
  const int my_box = blockIdx.x * 4 + (threadIdx.x >> 6);  // wave-uniform
  if (my_box >= n_boxes) return;
  const int lane = threadIdx.x & 63;
  const int self = my_box * kKnnBox + lane;
  const bool active = self < P;
  float4 me = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (active) me = sorted[self];
  float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
  visit_box(sorted, my_box, P, self, me.x, me.y, me.z, b0, b1, b2);
  const int my_super = my_box / kKnnSuper;
  // own super-box first: its boxes are the nearest in Morton order and shrink the bounds early
  for (int k = 0; k < n_supers; k++) {
    const int s = k == 0 ? my_super : (k <= my_super ? k - 1 : k);
    const bool need_s = active && !(dist_box_point(supers[s], me.x, me.y, me.z) > b2);
    if (__ballot(need_s) == 0) continue;
    const int b_end = min(n_boxes, (s + 1) * kKnnSuper);
    for (int b = s * kKnnSuper; b < b_end; b++) {
      if (b == my_box) continue;
      const bool need = need_s && !(dist_box_point(boxes[b], me.x, me.y, me.z) > b2);
      if (__ballot(need) == 0) continue;
      visit_box(sorted, b, P, self, me.x, me.y, me.z, b0, b1, b2);
    }
  }
  if (active) mean_dists[__float_as_uint(me.w)] = (b0 + b1 + b2) / 3.0f;
}

struct KnnScratch {
  unsigned *ext, *codes, *codes_sorted, *index, *order;
  float4* sorted;
  KnnBounds *boxes, *supers;
  void* sort_temp;
  size_t sort_bytes, total;
};

static size_t knn_sort_bytes(int P) {
  size_t bytes = 0;
  unsigned* d = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, d, d, d, d, (size_t)P, 0, 30);
  return bytes;
}

static KnnScratch knn_carve(char* base, int P) {
  KnnScratch k;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base + off; off += (bytes + 255) & ~(size_t)255; return p; };
  const int nb = (P + kKnnBox - 1) / kKnnBox, ns = (nb + kKnnSuper - 1) / kKnnSuper;
  k.ext = (unsigned*)take(6 * sizeof(unsigned));
  k.codes = (unsigned*)take((size_t)P * 4);
  k.codes_sorted = (unsigned*)take((size_t)P * 4);
  k.index = (unsigned*)take((size_t)P * 4);
  k.order = (unsigned*)take((size_t)P * 4);
  k.sorted = (float4*)take((size_t)P * 16);
  k.boxes = (KnnBounds*)take((size_t)nb * sizeof(KnnBounds));
  k.supers = (KnnBounds*)take((size_t)ns * sizeof(KnnBounds));
  k.sort_bytes = knn_sort_bytes(P);
  k.sort_temp = take(k.sort_bytes);
  k.total = off;
  return k;
}

}  // namespace gigs

extern "C" {
int gigs_internal_fail(int code, const char* fmt, ...);
void gigs_internal_stage_begin(int stage, void* stream, void** token);
void gigs_internal_stage_end(void* token);

size_t gigs_dist2_scratch_bytes(int P) {
  if (P <= 0) return 0;
  return gigs::knn_carve(nullptr, P).total;
}

int gigs_dist2(int P, const float* points, float* mean_dists, void* scratch, size_t scratch_bytes, void* stream) {
  if (P < 0 || (P > 0 && (!points || !mean_dists || !scratch)))
    return gigs_internal_fail(GIGS_ERR_INVALID, "dist2: bad argument");
  if (P == 0) return 0;
  gigs::KnnScratch k = gigs::knn_carve((char*)scratch, P);
  if (scratch_bytes < k.total) return gigs_internal_fail(GIGS_ERR_INVALID, "dist2: scratch too small");
  hipStream_t s = (hipStream_t)stream;
  void* tok; gigs_internal_stage_begin(29, stream, &tok);
  const int nb = (P + gigs::kKnnBox - 1) / gigs::kKnnBox, ns = (nb + gigs::kKnnSuper - 1) / gigs::kKnnSuper;
  const int pblocks = (P + 255) / 256;
  hipLaunchKernelGGL(gigs::knn_init_kernel, dim3(1), dim3(64), 0, s, k.ext);
  hipLaunchKernelGGL(gigs::knn_extent_kernel, dim3(pblocks < 1024 ? pblocks : 1024), dim3(256), 0, s, P, points, k.ext);
  hipLaunchKernelGGL(gigs::knn_morton_kernel, dim3(pblocks), dim3(256), 0, s, P, points, k.ext, k.codes, k.index);
  size_t bytes = k.sort_bytes;
  if (rocprim::radix_sort_pairs(k.sort_temp, bytes, k.codes, k.codes_sorted, k.index, k.order, (size_t)P, 0, 30, s) !=
      hipSuccess) {
    gigs_internal_stage_end(tok);
    return gigs_internal_fail(GIGS_ERR_HIP, "dist2: sort failed");
  }
  hipLaunchKernelGGL(gigs::knn_gather_kernel, dim3((nb * 64 + 255) / 256), dim3(256), 0, s, P, points, k.order, k.sorted,
                     k.boxes);
  hipLaunchKernelGGL(gigs::knn_super_kernel, dim3((ns * 64 + 255) / 256), dim3(256), 0, s, nb, k.boxes, k.supers);
  hipLaunchKernelGGL(gigs::knn_search_kernel, dim3((nb + 3) / 4), dim3(256), 0, s, P, nb, ns, k.sorted, k.boxes, k.supers,
                     mean_dists);
  gigs_internal_stage_end(tok);
  if (hipGetLastError() != hipSuccess) return gigs_internal_fail(GIGS_ERR_HIP, "dist2: launch failed");
  return 0;
}

}  // extern "C"
