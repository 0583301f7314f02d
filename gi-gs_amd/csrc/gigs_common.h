// gigs_common.h -- shared device/host helpers and scratch-buffer layouts (gfx950 only).
//
// Arithmetic rule for this library: every kernel whose outputs feed an INTEGER result that
// must be bit-exact (radii, tile rectangles, depth sort keys, point lists, n_contrib) is
// written as plain fp32 mul/add/div/sqrt in the reference's operation order and the whole
// library is compiled with -ffp-contract=off, so hipcc forms no FMAs on its own.  Division
// and sqrt are the correctly rounded forms (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt); fp32 denormals are kept (gfx9 default).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GIGS_BLOCK_X 16  // reference tile: R/cuda_rasterizer/config.h:15-17
#define GIGS_BLOCK_Y 16
#define GIGS_TILE (GIGS_BLOCK_X * GIGS_BLOCK_Y)
#define GIGS_BREC_F4 5   // packed blend record: 5 x float4 = 80 B per Gaussian
#define GIGS_GREC 20     // packed gradient record: 20 floats = 80 B per Gaussian

namespace gigs {

struct v3 { float x, y, z; };
struct v4 { float x, y, z, w; };

__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 cross3(v3 a, v3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// sutil normalize: v * (1 / sqrtf(dot(v, v))) -> NaN for the zero vector
// (R/cuda_rasterizer/vec_math.h:538-541)
__device__ __forceinline__ v3 normalize3(v3 v) {
  float inv = 1.0f / sqrtf(dot3(v, v));
  return v * inv;
}

// Column-major 3x3 (m[c][r]) with the product order of GLM's mat3 * mat3
// (R/third_party/glm/glm/detail/type_mat3x3.inl:486-519).
struct m3 { float m[3][3]; };
__device__ __forceinline__ m3 make_m3(float a, float b, float c, float d, float e, float f,
                                       float g, float h, float i) {
  m3 r;
  r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
  r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
  r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
  return r;
}
__device__ __forceinline__ m3 mul3(const m3& A, const m3& B) {
  m3 R;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++)
      R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
  return R;
}
__device__ __forceinline__ m3 transpose3(const m3& A) {
  m3 R;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
  return R;
}

// R/cuda_rasterizer/auxiliary.h:58-108 -- matrices are indexed m[0],m[4],m[8],m[12] per row
__device__ __forceinline__ v3 xform_point_4x3(v3 p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
          m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
__device__ __forceinline__ v4 xform_point_4x4(v3 p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
          m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
          m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}
__device__ __forceinline__ v3 xform_vec_4x3(v3 p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z,
          m[1] * p.x + m[5] * p.y + m[9] * p.z,
          m[2] * p.x + m[6] * p.y + m[10] * p.z};
}
__device__ __forceinline__ v3 xform_vec_4x3_T(v3 p, const float* m) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z,
          m[4] * p.x + m[5] * p.y + m[6] * p.z,
          m[8] * p.x + m[9] * p.y + m[10] * p.z};
}

// float -> int: v_cvt_i32_f32 saturates and maps NaN to 0, the same as CUDA's cvt.rzi.
__device__ __forceinline__ int f2i(float f) { return (int)f; }

// auxiliary.h:41-44 (double arithmetic because of the 1.0 / 0.5 literals)
__device__ __forceinline__ float ndc2pix(float v, int S) {
  return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

// auxiliary.h:46-56
__device__ __forceinline__ void tile_rect(float px, float py, int max_radius, unsigned gx,
                                          unsigned gy, unsigned& minx, unsigned& miny,
                                          unsigned& maxx, unsigned& maxy) {
  minx = min(gx, (unsigned)max(0, f2i((px - max_radius) / GIGS_BLOCK_X)));
  miny = min(gy, (unsigned)max(0, f2i((py - max_radius) / GIGS_BLOCK_Y)));
  maxx = min(gx, (unsigned)max(0, f2i((px + max_radius + GIGS_BLOCK_X - 1) / GIGS_BLOCK_X)));
  maxy = min(gy, (unsigned)max(0, f2i((py + max_radius + GIGS_BLOCK_Y - 1) / GIGS_BLOCK_Y)));
}

}  // namespace gigs

// ------------------------------------------------------------------------------------------
// Scratch layouts (host side).  Same role as GeometryState / ImageState / BinningState
// (R/cuda_rasterizer/rasterizer_impl.h:21-74) but laid out for this implementation: in
// addition to the reference's per-Gaussian arrays the geometry chunk holds
//   brec : one 80-byte packed record per Gaussian that the blend kernels gather with five
//          16-byte loads instead of nine scattered 4..16-byte gathers
//          f4[0] = (mean2D.x, mean2D.y, roughness, metallic)
//          f4[1] = (conic.x, conic.y, conic.z, opacity)
//          f4[2] = (rgb.r, rgb.g, rgb.b, pos_view.x)
//          f4[3] = (normal.x, normal.y, normal.z, pos_view.y)
//          f4[4] = (albedo.r, albedo.g, albedo.b, pos_view.z)      (pos_view.z == depth)
//   grec : one 80-byte packed gradient record per Gaussian, the atomic target of the blend
//          backward (one multi-lane atomic instruction writes one contiguous row)
//          [0..2] dL_dmean2D(x, y, |.|)  [3..5] dL_dconic(xx, xy, yy)  [6] dL_dopacity
//          [7..9] dL_dcolor  [10..12] dL_dnormal  [13..15] dL_dalbedo  [16] dL_droughness
//          [17] dL_dmetallic  [18] dL_ddepth  [19] unused
// ------------------------------------------------------------------------------------------
namespace gigs {

constexpr size_t kAlign = 256;

template <typename T>
inline void carve(char*& chunk, T*& ptr, size_t count) {
  size_t off = (reinterpret_cast<uintptr_t>(chunk) + kAlign - 1) & ~(kAlign - 1);
  ptr = reinterpret_cast<T*>(off);
  chunk = reinterpret_cast<char*>(ptr + count);
}

struct GeomState {
  float* depths;
  float* pos_view;
  float* means2D;
  float* cov3D;
  float* conic_opacity;
  float* rgb;
  uint8_t* clamped;
  uint32_t* tiles_touched;
  uint32_t* point_offsets;
  int* internal_radii;
  float4* brec;
  float* grec;
  char* scan_space;
  size_t scan_size;
  static GeomState fromChunk(char*& chunk, size_t P, size_t scan_size) {
    GeomState g;
    carve(chunk, g.depths, P);
    carve(chunk, g.pos_view, 3 * P);
    carve(chunk, g.means2D, 2 * P);
    carve(chunk, g.cov3D, 6 * P);
    carve(chunk, g.conic_opacity, 4 * P);
    carve(chunk, g.rgb, 3 * P);
    carve(chunk, g.clamped, 3 * P);
    carve(chunk, g.tiles_touched, P);
    carve(chunk, g.point_offsets, P);
    carve(chunk, g.internal_radii, P);
    carve(chunk, g.brec, GIGS_BREC_F4 * P);
    carve(chunk, g.grec, (size_t)GIGS_GREC * P);
    g.scan_size = scan_size;
    carve(chunk, g.scan_space, scan_size);
    return g;
  }
};

// Tile-bucketed binning (binning.hip): the Gaussians are cut into at most kBinGroups contiguous chunks, one workgroup
// each; bin_hist[g * T + t] holds first the number of instances chunk g contributes to tile t and then (in place)
// their exclusive prefix over the chunks.
#ifndef GIGS_BIN_GROUPS
// measured at C4 (3 M Gaussians, count / scatter in ms): 64 groups 0.52 / 1.32, 128: 0.27 / 0.79, 256: 0.14 / 0.51,
// 512: 0.11 / 0.44, 1024: 0.11 / 0.45 -- the walks are bound by the parallelism in flight, two 1024-lane groups fill a CU
#define GIGS_BIN_GROUPS 512
#endif
constexpr int kBinGroups = GIGS_BIN_GROUPS;
constexpr int kBinMaxTiles = 16384;
constexpr int kBucketMaxMeanList = 2500;  // mean instances per tile above which synchronous calls use the global radix sort  // the per-workgroup tile histogram lives in LDS (64 KB at this size)

struct ImageState {
  float* final_T;
  uint32_t* n_contrib;
  uint2* ranges;
  uint32_t* tile_order;  // workgroup b of the blend kernels handles tile tile_order[b]: longest lists first
  uint32_t* bin_hist;    // [kBinGroups + 2][T]
  uint32_t* bin_counters;  // [0] = R (instances of this forward), [1] = R if it exceeded the capacity (else 0)
  static ImageState fromChunk(char*& chunk, size_t N, size_t T) {
    ImageState s;
    carve(chunk, s.final_T, N);
    carve(chunk, s.n_contrib, N);
    carve(chunk, s.ranges, T);
    carve(chunk, s.tile_order, T);
    carve(chunk, s.bin_hist, (size_t)(kBinGroups + 2) * T);  // + the tile totals and the unclamped tile starts
    carve(chunk, s.bin_counters, 4);
    return s;
  }
};

struct BinningState {
  uint64_t* keys_unsorted;
  uint32_t* values_unsorted;
  uint64_t* keys;
  uint32_t* point_list;
  uint8_t* hit_mask;  // [4][R]: byte i of plane w is 1 iff quadrant (wave) w of instance i's tile had a pixel that blended it
  char* sort_space;
  size_t sort_size;
  static BinningState fromChunk(char*& chunk, size_t R, size_t sort_size) {
    BinningState b;
    carve(chunk, b.keys_unsorted, R);
    carve(chunk, b.values_unsorted, R);
    carve(chunk, b.keys, R);
    carve(chunk, b.point_list, R);
    carve(chunk, b.hit_mask, 4 * R);
    b.sort_size = sort_size;
    carve(chunk, b.sort_space, sort_size);
    return b;
  }
};

template <typename S, typename... A>
inline size_t required_bytes(A... a) {
  char* p = nullptr;
  S::fromChunk(p, a...);
  return reinterpret_cast<size_t>(p) + kAlign;
}

// ---- per-context state (include/gigs_hip.h: gigs_ctx / gigs_options) ------------------------
// Every switch a launch path reads lives here: the library itself keeps no mutable process-wide state besides
// immutable caches (ray tables, texel tables, rocPRIM temp sizes) and the diagnostic profile session.  The
// default options are the environment, parsed ONCE (first use); a context starts as a copy of them.
struct Options {
  int binning_legacy;    // GIGS_BINNING=legacy
  int bucket_max_mean;   // GIGS_BUCKET_MAX_MEAN
  int long_lists;        // GIGS_LONG_LISTS: -1 auto, 0 never, 1 always
  int bucket_target;     // GIGS_BUCKET_TARGET
  int bin_bands;         // GIGS_BIN_BANDS: passes of the by-tile scatter over bands of tile rows (0 = by density: 1 / 4)
  int blend_cull;        // GIGS_BLEND_CULL
  int pre_bwd_sh_skip;   // GIGS_PRE_BWD_SH_SKIP
  int gi_march;          // GIGS_GI_MARCH: 0 exact, 1 hoist, 2 hoist_fma, 3 proj_nr, 4 proj
  int gi_cert;           // GIGS_GI_CERT
  int gi_interleave;     // GIGS_GI_INTERLEAVE
  int gi_tile_log2w;     // GIGS_GI_TILE_LOG2W
  int gi_zero_rays;      // GIGS_GI_ZERO_RAYS: 1 = march the zero-weight rays too (diagnostic; same bits)
  int spec_max8, spec_max16;  // GIGS_SPEC_MAX8 / _MAX16
  int shade_lds_floats;  // GIGS_SHADE_LDS_FLOATS
  int shade_bwd_blocks;  // GIGS_SHADE_BWD_BLOCKS (0 = one workgroup per CU)
};
struct Ctx {
  Options opt;
  unsigned async_capacity;  // gigs_ctx_set_async_binning
  unsigned* async_counters;
  void* blend_begin_event;  // gigs_ctx_set_blend_begin_event
  int reuse_binning;        // gigs_ctx_set_reuse_binning
  unsigned* materials_only;  // gigs_ctx_set_materials_only: violation counter of the declared stage-2 gradient set
  const float* sh_rest;      // gigs_ctx_set_split_sh
};
const Options& default_options();  // api.hip
const Ctx& default_ctx();          // options = default_options(), no async binning, no event

// ---- kernel launchers implemented in the .hip files --------------------------------------
struct FwdArgs {
  int P, D, M, W, H;
  unsigned gx, gy;
  float focal_x, focal_y, tan_fovx, tan_fovy, scale_modifier;
  const float *means3D, *shs, *colors_precomp, *opacities, *normal, *albedo, *roughness,
      *metallic, *scales, *rotations, *cov3D_precomp, *viewmatrix, *projmatrix, *cam_pos,
      *background;
  int argmax_depth, inference;
  const float* shs_rest;  // gigs_ctx_set_split_sh: `shs` = coefficient 0 [P,1,3], coefficients 1..M-1 here [P,M-1,3]
};

void launch_preprocess_fwd(const FwdArgs& a, const GeomState& g, int* radii, hipStream_t s);
void launch_zero_words(uint32_t* p, size_t n, hipStream_t s);
// tile-bucketed binning: count -> prefix (ranges, R) -> scatter -> per-tile sort
void launch_bin_count(int P, const int* radii, unsigned gx, unsigned gy, const GeomState& g, const ImageState& img, hipStream_t s);
void launch_bin_prefix(int P, int T, unsigned capacity, const ImageState& img, unsigned* user_counters, hipStream_t s);
void launch_bin_scatter(int P, const int* radii, unsigned gx, unsigned gy, unsigned capacity, unsigned bands, const GeomState& g,
                        const BinningState& b, const ImageState& img, hipStream_t s);
// long_lists: tiles above 8192 keys are partitioned by sampled splitters and sorted bucket by bucket (dense scenes)
int launch_bin_sort(int T, int P, bool long_lists, unsigned bucket_target, unsigned capacity, const BinningState& b, const ImageState& img,
                    hipStream_t s);
size_t long_space_bytes(size_t T, size_t R);
void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present,
                         hipStream_t s);
hipError_t scan_tiles(const GeomState& g, int P, hipStream_t s);
size_t scan_temp_bytes(int P);
size_t sort_temp_bytes(int R);
void launch_duplicate(int P, const int* radii, unsigned gx, unsigned gy, const GeomState& g,
                      const BinningState& b, hipStream_t s);
hipError_t sort_pairs(const BinningState& b, int R, int end_bit, hipStream_t s);
void launch_tile_ranges(int R, const BinningState& b, uint2* ranges, hipStream_t s);
void launch_tile_order(int T, const uint2* ranges, uint32_t* tile_order, hipStream_t s);
void launch_blend_fwd(const FwdArgs& a, const GeomState& g, const BinningState& b,
                      const ImageState& im, float* out_color, float* out_opacity,
                      float* out_depth, float* out_normal, float* out_normal_view, float* out_pos,
                      float* out_albedo, float* out_roughness, float* out_metallic, int cull, size_t hit_stride,
                      hipStream_t s, bool reused_lists = false);  // hit_stride = the R the binning chunk was carved for

struct BwdArgs {
  int P, D, M, R, W, H;
  unsigned gx, gy;
  float focal_x, focal_y, tan_fovx, tan_fovy, scale_modifier;
  const float *means3D, *shs, *colors_precomp, *scales, *rotations, *cov3D_precomp, *viewmatrix,
      *projmatrix, *cam_pos, *background;
  const int* radii;
  const float *dL_dpix_depth, *dL_dpix, *dL_dpix_opacity, *dL_dpix_normal, *dL_dpix_albedo,
      *dL_dpix_roughness, *dL_dpix_metallic;
  float *dL_dmean2D, *dL_dconic, *dL_ddepth, *dL_dopacity, *dL_dnormal, *dL_dalbedo,
      *dL_droughness, *dL_dmetallic, *dL_dcolor, *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale,
      *dL_drot;
};
void launch_blend_bwd(const BwdArgs& a, const GeomState& g, const BinningState& b,
                      const ImageState& im, hipStream_t s);
void launch_preprocess_bwd(const BwdArgs& a, const GeomState& g, int sh_skip, hipStream_t s, unsigned* materials_only = nullptr);

void launch_depth_to_normal(int W, int H, float fx, float fy, const float* viewmatrix,
                            const float* depth, float* normal, float* depth_pos, hipStream_t s);
int launch_ssao(const Options& o, int W, int H, float fx, float fy, float radius, float bias, float thick,
                float delta, int step, int start, const float* normal, const float* pos,
                float* occlusion, void* scratch, hipStream_t s);
size_t gi_scratch_bytes(int W, int H);
int launch_ssr(const Options& o, int W, int H, float fx, float fy, float radius, float bias, float thick,
               float delta, int step, int start, const float* normal, const float* pos,
               const float* rgb, const float* albedo, const float* roughness,
               const float* metallic, const float* F0, float* color, float* abd, void* scratch, hipStream_t s,
               int hits_mode = 0, unsigned* hit_counts = nullptr, const unsigned* hit_offsets = nullptr,
               void* hit_entries = nullptr, unsigned hit_capacity = 0);
int launch_ssr_apply(int W, int H, float delta, const unsigned* offsets, const void* entries, const float* normal,
                     const float* pos, const float* rgb, const float* albedo, const float* metallic, const float* F0,
                     float* color, float* abd, hipStream_t s);
void launch_median3x3(int C, int H, int W, const float* in, float* out, hipStream_t s);
void launch_median3x3_bwd(int C, int H, int W, const float* in, const float* gout, float* gin,
                          hipStream_t s);
void launch_bilateral3x3(int C, int H, int W, float sigma_color, float sx, float sy,
                         const float* in, float* out, hipStream_t s);
void launch_derive_normal_fused(int W, int H, float fx, float fy, const float* viewmatrix, float sigma_color, float sx,
                                float sy, const float* depth_raw, float* normal_out, float* pos_filter_out, hipStream_t s);

}  // namespace gigs
