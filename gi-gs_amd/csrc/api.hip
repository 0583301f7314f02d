// api.hip -- the C ABI of libgigs_hip.so (declared in include/gigs_hip.h).
//
// Host-side orchestration only; it mirrors CudaRasterizer::Rasterizer::forward / backward
// (R/cuda_rasterizer/rasterizer_impl.cu:486-803): preprocess -> inclusive scan -> 4-byte D2H
// read of num_rendered -> duplicate -> radix sort -> tile ranges -> blend, with the scratch
// chunks obtained from the caller through allocation callbacks.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <algorithm>
#include <new>
#include <mutex>
#include <vector>

#include "../../include/gigs_hip.h"
#include "gigs_common.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return fail(GIGS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// CHECK_CUDA of the reference (auxiliary.h:178-185): in debug mode synchronise and report
// after every stage.
#define STAGE_CHECK(name)                                                                  \
  do {                                                                                     \
    hipError_t _e = hipGetLastError();                                                     \
    if (_e == hipSuccess && debug) _e = hipStreamSynchronize(s);                           \
    if (_e != hipSuccess)                                                                  \
      return fail(GIGS_ERR_HIP, "stage '%s' failed: %s", name, hipGetErrorString(_e));     \
  } while (0)

// temp-storage sizes of rocPRIM depend only on the element count: memoise them
std::mutex g_size_mu;
std::map<int, size_t> g_scan_sizes, g_sort_sizes;
size_t scan_size_cached(int P) {
  std::lock_guard<std::mutex> lk(g_size_mu);
  auto it = g_scan_sizes.find(P);
  if (it != g_scan_sizes.end()) return it->second;
  const size_t v = gigs::scan_temp_bytes(P);
  g_scan_sizes[P] = v;
  return v;
}
size_t sort_size_cached(int R) {
  // sizes are monotone in R within a rocPRIM config; cache by power-of-two bucket to keep the
  // map small and the binning chunk stable between iterations
  int bucket = 1024;
  while (bucket < R) bucket <<= 1;
  std::lock_guard<std::mutex> lk(g_size_mu);
  auto it = g_sort_sizes.find(bucket);
  if (it != g_sort_sizes.end()) return it->second;
  // the tile-bucketed path keeps the tables of its long-list partition in the same space instead (binning.hip)
  const size_t v = std::max(gigs::sort_temp_bytes(bucket), gigs::long_space_bytes((size_t)gigs::kBinMaxTiles, (size_t)bucket));
  g_sort_sizes[bucket] = v;
  return v;
}

// R/cuda_rasterizer/rasterizer_impl.cu:35-50
uint32_t higher_msb(uint32_t n) {
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

// ---- in-library per-stage timing --------------------------------------------------------
// A "profile session" records one hipEvent pair per kernel stage on the stream the stage is
// launched on, WITHOUT synchronising; gigs_profile_end() synchronises once and sums the
// elapsed times.  bench.py uses it to measure kernel durations live over its timed region.
enum Stage {
  kPreprocess = 0, kScan, kDuplicate, kSort, kRanges, kBlendFwd, kBlendBwd, kPreprocessBwd,
  kDepthToNormal, kSsao, kSsr, kMedian, kBilateral, kMedianBwd, kShadeFwd, kShadeBwd,
  kCubemapFwd, kCubemapBwd, kGbufferPost, kLossFwd, kLossBwd, kL1SsimFwd, kL1SsimBwd, kTvFwd, kTvBwd,
  kMaskedL1, kAdam, kDensifyStats, kGatherRows, kDist2, kActivateFwd, kActivateBwd, kNumStages
};
const char* kStageNames[kNumStages] = {
  "preprocess_fwd", "scan", "duplicate", "sort", "tile_ranges", "blend_fwd", "blend_bwd",
  "preprocess_bwd", "depth_to_normal", "ssao", "ssr", "median3x3", "bilateral3x3",
  "median3x3_bwd", "shade_fwd", "shade_bwd", "cubemap_fwd", "cubemap_bwd", "gbuffer_post",
  "stage2_loss_fwd", "stage2_loss_bwd", "l1_ssim_fwd", "l1_ssim_bwd", "tv_loss_fwd", "tv_loss_bwd", "masked_l1",
  "adam_step", "densify_stats", "gather_rows", "dist2", "activate_fwd", "activate_bwd"};

// ---- contexts (include/gigs_hip.h) -------------------------------------------------------------------------------
int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && e[0]) ? atoi(e) : dflt;
}
gigs::Options options_from_env() {
  gigs::Options o;
  const char* b = getenv("GIGS_BINNING");
  o.binning_legacy = (b && strcmp(b, "legacy") == 0) ? 1 : 0;
  o.bucket_max_mean = env_int("GIGS_BUCKET_MAX_MEAN", gigs::kBucketMaxMeanList);
  const char* ll = getenv("GIGS_LONG_LISTS");
  o.long_lists = (ll && ll[0]) ? (ll[0] == '1' ? 1 : 0) : -1;
  o.bucket_target = std::max(256, env_int("GIGS_BUCKET_TARGET", 1536));
  o.bin_bands = std::max(0, env_int("GIGS_BIN_BANDS", 0));
  o.blend_cull = env_int("GIGS_BLEND_CULL", 1) != 0;
  o.pre_bwd_sh_skip = env_int("GIGS_PRE_BWD_SH_SKIP", 1) != 0;
  o.gi_march = 4;
  if (const char* e = getenv("GIGS_GI_MARCH")) {
    static const char* names[] = {"exact", "hoist", "hoist_fma", "proj_nr", "proj"};
    for (int i = 0; i < 5; i++)
      if (strcmp(e, names[i]) == 0) o.gi_march = i;
  }
  o.gi_cert = env_int("GIGS_GI_CERT", 1) != 0;
  o.gi_interleave = env_int("GIGS_GI_INTERLEAVE", 1) != 0;
  o.gi_tile_log2w = env_int("GIGS_GI_TILE_LOG2W", 3);
  if (o.gi_tile_log2w < 0 || o.gi_tile_log2w > 6) o.gi_tile_log2w = 3;
  o.gi_zero_rays = env_int("GIGS_GI_ZERO_RAYS", 0) != 0;
  o.spec_max8 = env_int("GIGS_SPEC_MAX8", 128);
  o.spec_max16 = env_int("GIGS_SPEC_MAX16", 1500);
  o.shade_lds_floats = env_int("GIGS_SHADE_LDS_FLOATS", 30 * 1024);
  o.shade_bwd_blocks = std::max(0, env_int("GIGS_SHADE_BWD_BLOCKS", 0));
  return o;
}
const gigs::Ctx& ctx_of(const gigs_ctx* c) { return c ? *reinterpret_cast<const gigs::Ctx*>(c) : gigs::default_ctx(); }

struct ProfRec { int stage; hipEvent_t a, b; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;

hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

// RAII: records start at construction and stop at destruction when a session is active.
struct StageScope {
  hipStream_t s;
  int stage;
  bool on;
  hipEvent_t a, b;
  StageScope(int stage_, hipStream_t s_) : s(s_), stage(stage_), on(false) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on) return;
    on = true;
    a = prof_event();
    b = prof_event();
    hipEventRecord(a, s);
  }
  ~StageScope() {
    if (!on) return;
    hipEventRecord(b, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_recs.push_back({stage, a, b});
  }
};

}  // namespace

namespace gigs {
const Options& default_options() {
  static const Options o = options_from_env();  // the environment is read here, once
  return o;
}
const Ctx& default_ctx() {
  static const Ctx c = {default_options(), 0u, nullptr, nullptr, 0, nullptr, nullptr};
  return c;
}
}  // namespace gigs

// helpers for the other translation units of the library (not part of the ABI)
extern "C" {
__attribute__((visibility("hidden"))) const gigs::Options* gigs_internal_options(const gigs_ctx* ctx) { return &ctx_of(ctx).opt; }
__attribute__((visibility("hidden"))) int gigs_internal_fail(int code, const char* msg) { return fail(code, "%s", msg); }
__attribute__((visibility("hidden"))) void gigs_internal_stage_begin(int stage, void* stream, void** token) {
  *token = new StageScope(stage, (hipStream_t)stream);
}
__attribute__((visibility("hidden"))) void gigs_internal_stage_end(void* token) { delete (StageScope*)token; }
}

extern "C" {

const char* gigs_last_error(void) { return g_err; }
const char* gigs_build_arch(void) { return "gfx950"; }

size_t gigs_required_geom(int P) {
  if (P < 0) { fail(GIGS_ERR_INVALID, "P < 0"); return 0; }
  return gigs::required_bytes<gigs::GeomState>((size_t)P, scan_size_cached(P));
}
size_t gigs_required_image(int width, int height) {
  const size_t N = (size_t)width * height;
  const size_t T = (size_t)((width + GIGS_BLOCK_X - 1) / GIGS_BLOCK_X) * ((height + GIGS_BLOCK_Y - 1) / GIGS_BLOCK_Y);
  return gigs::required_bytes<gigs::ImageState>(N, T);
}
size_t gigs_required_binning(int R) {
  if (R < 0) { fail(GIGS_ERR_INVALID, "R < 0"); return 0; }
  return gigs::required_bytes<gigs::BinningState>((size_t)R, sort_size_cached(R));
}

long long gigs_geom_offset(int P, int which) {
  char* base = nullptr;
  char* p = base;
  gigs::GeomState g = gigs::GeomState::fromChunk(p, (size_t)P, 0);
  const void* ptrs[] = {g.depths, g.pos_view, g.means2D, g.cov3D, g.conic_opacity, g.rgb,
                        g.clamped, g.tiles_touched, g.point_offsets, g.internal_radii, g.brec, g.grec};
  if (which < 0 || which >= (int)(sizeof(ptrs) / sizeof(ptrs[0]))) return -1;
  return (long long)((const char*)ptrs[which] - base);
}
long long gigs_binning_offset(int R, int which) {
  char* base = nullptr;
  char* p = base;
  gigs::BinningState b = gigs::BinningState::fromChunk(p, (size_t)R, 0);
  const void* ptrs[] = {b.keys_unsorted, b.values_unsorted, b.keys, b.point_list, b.hit_mask};
  if (which < 0 || which >= 5) return -1;
  return (long long)((const char*)ptrs[which] - base);
}
long long gigs_image_offset(int width, int height, int which) {
  const size_t N = (size_t)width * height;
  const size_t T = (size_t)((width + GIGS_BLOCK_X - 1) / GIGS_BLOCK_X) * ((height + GIGS_BLOCK_Y - 1) / GIGS_BLOCK_Y);
  char* base = nullptr;
  char* p = base;
  gigs::ImageState s = gigs::ImageState::fromChunk(p, N, T);
  const void* ptrs[] = {s.final_T, s.n_contrib, s.ranges, s.tile_order};
  if (which < 0 || which >= 4) return -1;
  return (long long)((const char*)ptrs[which] - base);
}

gigs_ctx* gigs_ctx_create(void) {
  gigs::Ctx* c = new (std::nothrow) gigs::Ctx(gigs::default_ctx());
  if (!c) fail(GIGS_ERR_ALLOC, "gigs_ctx_create: out of host memory");
  return reinterpret_cast<gigs_ctx*>(c);
}
void gigs_ctx_destroy(gigs_ctx* ctx) { delete reinterpret_cast<gigs::Ctx*>(ctx); }

int gigs_ctx_get_options(const gigs_ctx* ctx, gigs_options* out) {
  if (!out || out->struct_bytes < (int)sizeof(gigs_options)) return fail(GIGS_ERR_INVALID, "gigs_ctx_get_options: set struct_bytes = sizeof(gigs_options)");
  const gigs::Options& o = ctx_of(ctx).opt;
  out->struct_bytes = (int)sizeof(gigs_options);
  out->binning_legacy = o.binning_legacy; out->bucket_max_mean = o.bucket_max_mean; out->long_lists = o.long_lists;
  out->bucket_target = o.bucket_target; out->bin_bands = o.bin_bands; out->blend_cull = o.blend_cull; out->pre_bwd_sh_skip = o.pre_bwd_sh_skip;
  out->gi_march = o.gi_march; out->gi_cert = o.gi_cert; out->gi_interleave = o.gi_interleave;
  out->gi_tile_log2w = o.gi_tile_log2w; out->gi_zero_rays = o.gi_zero_rays; out->spec_max8 = o.spec_max8;
  out->spec_max16 = o.spec_max16; out->shade_lds_floats = o.shade_lds_floats; out->shade_bwd_blocks = o.shade_bwd_blocks;
  return 0;
}
int gigs_ctx_set_options(gigs_ctx* ctx, const gigs_options* in) {
  if (!ctx) return fail(GIGS_ERR_INVALID, "gigs_ctx_set_options: the default context is immutable, create one");
  if (!in || in->struct_bytes < (int)sizeof(gigs_options)) return fail(GIGS_ERR_INVALID, "gigs_ctx_set_options: set struct_bytes = sizeof(gigs_options)");
  if (in->gi_march < 0 || in->gi_march > 4) return fail(GIGS_ERR_INVALID, "gi_march must be 0..4");
  if (in->gi_tile_log2w < 0 || in->gi_tile_log2w > 6) return fail(GIGS_ERR_INVALID, "gi_tile_log2w must be 0..6");
  if (in->long_lists < -1 || in->long_lists > 1) return fail(GIGS_ERR_INVALID, "long_lists must be -1, 0 or 1");
  if (in->bucket_target < 256 || in->bucket_max_mean < 0 || in->bin_bands < 0 || in->bin_bands > 64)
    return fail(GIGS_ERR_INVALID, "bucket_target >= 256, bucket_max_mean >= 0, 0 <= bin_bands <= 64");
  if (in->spec_max8 < 0 || in->spec_max16 < in->spec_max8) return fail(GIGS_ERR_INVALID, "0 <= spec_max8 <= spec_max16");
  if (in->shade_lds_floats < 0 || in->shade_bwd_blocks < 0) return fail(GIGS_ERR_INVALID, "shade_lds_floats, shade_bwd_blocks >= 0");
  gigs::Options& o = reinterpret_cast<gigs::Ctx*>(ctx)->opt;
  o.binning_legacy = in->binning_legacy != 0; o.bucket_max_mean = in->bucket_max_mean; o.long_lists = in->long_lists;
  o.bucket_target = in->bucket_target; o.bin_bands = in->bin_bands; o.blend_cull = in->blend_cull != 0; o.pre_bwd_sh_skip = in->pre_bwd_sh_skip != 0;
  o.gi_march = in->gi_march; o.gi_cert = in->gi_cert != 0; o.gi_interleave = in->gi_interleave != 0;
  o.gi_tile_log2w = in->gi_tile_log2w; o.gi_zero_rays = in->gi_zero_rays != 0; o.spec_max8 = in->spec_max8;
  o.spec_max16 = in->spec_max16; o.shade_lds_floats = in->shade_lds_floats; o.shade_bwd_blocks = in->shade_bwd_blocks;
  return 0;
}
int gigs_ctx_set_reuse_binning(gigs_ctx* ctx, int on) {
  if (!ctx) return fail(GIGS_ERR_INVALID, "gigs_ctx_set_reuse_binning: the default context is immutable, create one");
  reinterpret_cast<gigs::Ctx*>(ctx)->reuse_binning = on != 0;
  return 0;
}
int gigs_ctx_set_split_sh(gigs_ctx* ctx, const float* sh_rest) {
  if (!ctx) return fail(GIGS_ERR_INVALID, "gigs_ctx_set_split_sh: the default context is immutable, create one");
  reinterpret_cast<gigs::Ctx*>(ctx)->sh_rest = sh_rest;
  return 0;
}
int gigs_ctx_set_materials_only(gigs_ctx* ctx, void* violations) {
  if (!ctx) return fail(GIGS_ERR_INVALID, "gigs_ctx_set_materials_only: the default context is immutable, create one");
  reinterpret_cast<gigs::Ctx*>(ctx)->materials_only = reinterpret_cast<unsigned*>(violations);
  return 0;
}
int gigs_ctx_set_blend_begin_event(gigs_ctx* ctx, void* hip_event) {
  if (!ctx) return fail(GIGS_ERR_INVALID, "gigs_ctx_set_blend_begin_event: the default context is immutable, create one");
  reinterpret_cast<gigs::Ctx*>(ctx)->blend_begin_event = hip_event;
  return 0;
}

__global__ void __launch_bounds__(64) stream_delay_kernel(unsigned ticks) {
  // wall_clock64: constant 100 MHz counter; every wave reaches the exit after at most `ticks` (<= 100 000) ticks
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
}

int gigs_stream_delay(unsigned nanoseconds, void* stream) {
  if (nanoseconds == 0) return 0;
  const unsigned ns = nanoseconds > 1000000u ? 1000000u : nanoseconds;
  hipLaunchKernelGGL(stream_delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (ns + 9u) / 10u);
  return hipGetLastError() == hipSuccess ? 0 : gigs_internal_fail(GIGS_ERR_HIP, "stream_delay: launch failed");
}

int gigs_ctx_set_async_binning(gigs_ctx* ctx, int r_capacity, unsigned* device_counters) {
  if (!ctx) return fail(GIGS_ERR_INVALID, "gigs_ctx_set_async_binning: the default context is immutable, create one");
  gigs::Ctx* c = reinterpret_cast<gigs::Ctx*>(ctx);
  c->async_capacity = r_capacity > 0 ? (unsigned)r_capacity : 0u;
  c->async_counters = r_capacity > 0 ? device_counters : nullptr;
  return 0;
}

void gigs_profile_begin(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof_recs) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
  g_prof_recs.clear();
  g_prof_on = true;
}

int gigs_profile_end(float* total_ms, int* launches, int n) {
  std::vector<ProfRec> recs;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = false;
    recs.swap(g_prof_recs);
  }
  for (int i = 0; i < n; i++) { total_ms[i] = 0.0f; launches[i] = 0; }
  for (auto& r : recs) {
    hipEventSynchronize(r.b);
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.stage < n) {
      total_ms[r.stage] += ms;
      launches[r.stage] += 1;
    }
  }
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : recs) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
  return kNumStages;
}

const char* gigs_profile_stage_name(int stage) {
  return (stage >= 0 && stage < kNumStages) ? kStageNames[stage] : "";
}

int gigs_forward(gigs_ctx* ctx, gigs_alloc_fn geometryBuffer, void* geom_user, gigs_alloc_fn binningBuffer,
                 void* binning_user, gigs_alloc_fn imageBuffer, void* image_user, int P, int D, int M,
                 const float* background, int width, int height, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* opacities,
                 const float* normal, const float* albedo, const float* roughness,
                 const float* metallic, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                 const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                 int prefiltered, int argmax_depth, int inference, float* out_color,
                 float* out_opacity, float* out_depth, float* out_normal, float* out_normal_view,
                 float* out_pos, float* out_albedo, float* out_roughness, float* out_metallic,
                 int* radii, int debug, void* stream) {
  (void)prefiltered;  // the reference only uses it to trap on an impossible state (auxiliary.h:167-171)
  hipStream_t s = (hipStream_t)stream;
  const gigs::Ctx& cx = ctx_of(ctx);
  const gigs::Options& opt = cx.opt;
  if (P < 0 || width <= 0 || height <= 0) return fail(GIGS_ERR_INVALID, "bad P / image size");
  if (P == 0) return 0;  // rasterize_points.cu:190-191: outputs stay as the caller initialised them
  if (!geometryBuffer || !binningBuffer || !imageBuffer) return fail(GIGS_ERR_INVALID, "null allocation callback");
  if (!means3D || !opacities || !normal || !albedo || !roughness || !metallic || !viewmatrix ||
      !projmatrix || !cam_pos || !background)
    return fail(GIGS_ERR_INVALID, "null required input");
  if (!shs && !colors_precomp) return fail(GIGS_ERR_INVALID, "provide SHs or precomputed colors");
  if (!cov3D_precomp && (!scales || !rotations)) return fail(GIGS_ERR_INVALID, "provide scales+rotations or cov3D_precomp");
  if (!colors_precomp && (M <= 0 || (D + 1) * (D + 1) > M || D > 3)) return fail(GIGS_ERR_INVALID, "SH degree %d needs (D+1)^2 <= M=%d, D <= 3", D, M);

  gigs::FwdArgs a;
  a.P = P; a.D = D; a.M = M; a.W = width; a.H = height;
  a.gx = (width + GIGS_BLOCK_X - 1) / GIGS_BLOCK_X;
  a.gy = (height + GIGS_BLOCK_Y - 1) / GIGS_BLOCK_Y;
  a.focal_y = height / (2.0f * tan_fovy);
  a.focal_x = width / (2.0f * tan_fovx);
  a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.scale_modifier = scale_modifier;
  a.means3D = means3D; a.shs = shs; a.colors_precomp = colors_precomp; a.opacities = opacities;
  a.normal = normal; a.albedo = albedo; a.roughness = roughness; a.metallic = metallic;
  a.scales = scales; a.rotations = rotations; a.cov3D_precomp = cov3D_precomp;
  a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; a.cam_pos = cam_pos; a.background = background;
  a.argmax_depth = argmax_depth; a.inference = inference;
  a.shs_rest = (shs && !colors_precomp && M > 1) ? cx.sh_rest : nullptr;  // gigs_ctx_set_split_sh

  const size_t scan_sz = scan_size_cached(P);
  const size_t geom_bytes = gigs::required_bytes<gigs::GeomState>((size_t)P, scan_sz);
  char* geom_chunk = geometryBuffer(geom_bytes, geom_user);
  if (!geom_chunk) return fail(GIGS_ERR_ALLOC, "geometry buffer allocation of %zu bytes failed", geom_bytes);
  gigs::GeomState geom = gigs::GeomState::fromChunk(geom_chunk, (size_t)P, scan_sz);
  if (radii == nullptr) radii = geom.internal_radii;

  const size_t N = (size_t)width * height, T = (size_t)a.gx * a.gy;
  const size_t img_bytes = gigs::required_bytes<gigs::ImageState>(N, T);
  char* img_chunk = imageBuffer(img_bytes, image_user);
  if (!img_chunk) return fail(GIGS_ERR_ALLOC, "image buffer allocation of %zu bytes failed", img_bytes);
  gigs::ImageState img = gigs::ImageState::fromChunk(img_chunk, N, T);

  {
    StageScope sc(kPreprocess, s);
    gigs::launch_preprocess_fwd(a, geom, radii, s);
  }
  STAGE_CHECK("preprocess");
  int num_rendered = 0;
  gigs::BinningState bin;
  const unsigned async_cap = cx.async_capacity;
  const bool bucket = !opt.binning_legacy && T <= (size_t)gigs::kBinMaxTiles;
  if (async_cap > 0 && !bucket) return fail(GIGS_ERR_INVALID, "asynchronous binning needs the tile-bucketed path (options.binning_legacy = 0, <= %d tiles)", gigs::kBinMaxTiles);
  // dense scene (mean list above options.bucket_max_mean): lists beyond 8192 keys are partitioned by sampled splitters before
  // the LDS sorts (binning.hip).  options.long_lists = 1 / 0 forces / forbids it.
  auto long_lists = [&](size_t est_mean) {
    if (opt.long_lists >= 0) return opt.long_lists == 1;
    return est_mean > (size_t)opt.bucket_max_mean;
  };
  bool dense = false;
  if (cx.reuse_binning) {
    // Frozen geometry (gigs_ctx_set_reuse_binning): the caller's binning and image chunks still hold ranges / tile_order /
    // point_list of an earlier forward of this very view and geometry; only the per-Gaussian records (preprocess, above)
    // and the blend (below) run.  The layout is the asynchronous one, so the capacity names it.
    if (async_cap == 0 || !bucket)
      return fail(GIGS_ERR_INVALID, "reuse_binning needs asynchronous binning (gigs_ctx_set_async_binning) on the tile-bucketed path");
    num_rendered = (int)async_cap;
    const size_t sort_sz = sort_size_cached(num_rendered);
    const size_t bin_bytes = gigs::required_bytes<gigs::BinningState>((size_t)num_rendered, sort_sz);
    char* bin_chunk = binningBuffer(bin_bytes, binning_user);
    if (!bin_chunk) return fail(GIGS_ERR_ALLOC, "binning buffer allocation of %zu bytes failed", bin_bytes);
    bin = gigs::BinningState::fromChunk(bin_chunk, (size_t)num_rendered, sort_sz);
  } else if (bucket) {
    // Tile-bucketed binning (binning.hip): count -> prefix -> scatter -> per-tile sort, the instance count stays on
    // the device.  Synchronous calls (the reference's API returns num_rendered) read it back once, BEFORE the scatter,
    // to size the binning chunk exactly; with gigs_set_async_binning the chunk has the caller's capacity and nothing
    // is read back (the caller sizes it at twice a probed instance count: the density is judged from that).
    {
      StageScope sc(kDuplicate, s);
      gigs::launch_bin_count(P, radii, a.gx, a.gy, geom, img, s);
      gigs::launch_bin_prefix(P, (int)T, async_cap > 0 ? async_cap : 0x7fffffffu, img, cx.async_counters, s);
    }
    STAGE_CHECK("bin count / prefix");
    if (async_cap > 0) {
      num_rendered = (int)async_cap;
      dense = long_lists((size_t)async_cap / (2 * T));
    } else {
      uint32_t num_rendered_u = 0;
      HIP_TRY(hipMemcpyAsync(&num_rendered_u, img.bin_counters, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if (num_rendered_u > 0x7fffffffu) return fail(GIGS_ERR_INVALID, "num_rendered overflows int");
      num_rendered = (int)num_rendered_u;
      dense = long_lists((size_t)num_rendered / T);
    }
    const size_t sort_sz = sort_size_cached(num_rendered);
    const size_t bin_bytes = gigs::required_bytes<gigs::BinningState>((size_t)num_rendered, sort_sz);
    char* bin_chunk = binningBuffer(bin_bytes, binning_user);
    if (!bin_chunk) return fail(GIGS_ERR_ALLOC, "binning buffer allocation of %zu bytes failed", bin_bytes);
    bin = gigs::BinningState::fromChunk(bin_chunk, (size_t)num_rendered, sort_sz);
    {
      StageScope sc(kRanges, s);
      gigs::launch_tile_order((int)T, img.ranges, img.tile_order, s);
    }
    {
      StageScope sc(kSort, s);
      // dense scenes scatter in bands of tile rows (binning.hip::bin_scatter_kernel); options.bin_bands overrides
      const unsigned bands = opt.bin_bands > 0 ? (unsigned)opt.bin_bands : (dense ? 4u : 1u);
      gigs::launch_bin_scatter(P, radii, a.gx, a.gy, (unsigned)num_rendered, bands, geom, bin, img, s);
      if (gigs::launch_bin_sort((int)T, P, dense && num_rendered > 0, (unsigned)opt.bucket_target, (unsigned)num_rendered, bin, img, s) != 0)
        return fail(GIGS_ERR_HIP, "bin_sort: cannot fork the sort streams");
    }
    STAGE_CHECK("bin scatter / sort");
  } else {
  {
    StageScope sc(kScan, s);
    HIP_TRY(gigs::scan_tiles(geom, P, s));
  }
  STAGE_CHECK("scan");
  // the one blocking read of the forward (rasterizer_impl.cu:589): sizes the binning chunk
  uint32_t num_rendered_u = 0;
  HIP_TRY(hipMemcpyAsync(&num_rendered_u, geom.point_offsets + P - 1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (num_rendered_u > 0x7fffffffu) return fail(GIGS_ERR_INVALID, "num_rendered overflows int");
  num_rendered = (int)num_rendered_u;

  const size_t sort_sz = sort_size_cached(num_rendered);
  const size_t bin_bytes = gigs::required_bytes<gigs::BinningState>((size_t)num_rendered, sort_sz);
  char* bin_chunk = binningBuffer(bin_bytes, binning_user);
  if (!bin_chunk) return fail(GIGS_ERR_ALLOC, "binning buffer allocation of %zu bytes failed", bin_bytes);
  bin = gigs::BinningState::fromChunk(bin_chunk, (size_t)num_rendered, sort_sz);

  {
    StageScope sc(kDuplicate, s);
    gigs::launch_duplicate(P, radii, a.gx, a.gy, geom, bin, s);
  }
  STAGE_CHECK("duplicateWithKeys");
  const int bit = (int)higher_msb(a.gx * a.gy);
  if (num_rendered > 0) {
    StageScope sc(kSort, s);
    HIP_TRY(gigs::sort_pairs(bin, num_rendered, 32 + bit, s));
  }
  STAGE_CHECK("sort");
  {
    StageScope sc(kRanges, s);
    HIP_TRY(hipMemsetAsync(img.ranges, 0, T * sizeof(uint2), s));
    gigs::launch_tile_ranges(num_rendered, bin, img.ranges, s);
    gigs::launch_tile_order((int)T, img.ranges, img.tile_order, s);
  }
  STAGE_CHECK("identifyTileRanges");
  }
  if (void* ev = cx.blend_begin_event) HIP_TRY(hipEventRecord((hipEvent_t)ev, s));
  {
    StageScope sc(kBlendFwd, s);
    gigs::launch_blend_fwd(a, geom, bin, img, out_color, out_opacity, out_depth, out_normal,
                           out_normal_view, out_pos, out_albedo, out_roughness, out_metallic, opt.blend_cull,
                           (size_t)num_rendered, s, cx.reuse_binning != 0);
  }
  STAGE_CHECK("render");
  return num_rendered;
}

// Rasterizer::lite_forward (R/cuda_rasterizer/rasterizer.h:90-117, rasterizer_impl.cu:338-482; liteRenderCUDA forward.cu:279-418):
// colour / opacity / depth only.  The lite kernel composites exactly like the full one (same tests, same depth = view-space
// z), so this is the full forward over zero material attributes, with the planes the caller did not ask for in scratch: the
// geometry chunk is requested 8 floats per Gaussian larger (the zero attributes) and the image chunk 14 planes larger.
namespace {
struct LiteChunk { char* base; size_t bytes; };
char* lite_chunk_alloc(size_t n, void* user) {
  LiteChunk* c = static_cast<LiteChunk*>(user);
  return n <= c->bytes ? c->base : nullptr;
}
}  // namespace

int gigs_lite_forward(gigs_ctx* ctx, gigs_alloc_fn geometryBuffer, void* geom_user, gigs_alloc_fn binningBuffer, void* binning_user,
                      gigs_alloc_fn imageBuffer, void* image_user, int P, int D, int M, const float* background, int width,
                      int height, const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                      const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                      const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                      int prefiltered, int argmax_depth, float* out_color, float* out_opacity, float* out_depth, int* radii,
                      int debug, void* stream) {
  if (P < 0 || width <= 0 || height <= 0) return fail(GIGS_ERR_INVALID, "bad P / image size");
  if (P == 0) return 0;
  if (!geometryBuffer || !binningBuffer || !imageBuffer) return fail(GIGS_ERR_INVALID, "null allocation callback");
  if (!out_color || !out_opacity || !out_depth) return fail(GIGS_ERR_INVALID, "null output");
  const size_t N = (size_t)width * height;
  const size_t geom_bytes = gigs_required_geom(P), img_bytes = gigs_required_image(width, height);
  const size_t geom_extra = 8 * (size_t)P * sizeof(float) + gigs::kAlign, img_extra = 14 * N * sizeof(float) + gigs::kAlign;
  char* gb = geometryBuffer(geom_bytes + geom_extra, geom_user);
  char* ib = imageBuffer(img_bytes + img_extra, image_user);
  if (!gb || !ib) return fail(GIGS_ERR_ALLOC, "lite_forward: scratch allocation failed");
  auto up = [](char* p) { return reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + gigs::kAlign - 1) & ~(uintptr_t)(gigs::kAlign - 1)); };
  float* zeros = reinterpret_cast<float*>(up(gb + geom_bytes));   // normal 3P | albedo 3P | roughness P | metallic P
  float* spare = reinterpret_cast<float*>(up(ib + img_bytes));    // normal 3N | normal_view 3N | pos 3N | albedo 3N | roughness N | metallic N
  gigs::launch_zero_words(reinterpret_cast<uint32_t*>(zeros), 8 * (size_t)P, (hipStream_t)stream);
  LiteChunk gc{gb, geom_bytes}, ic{ib, img_bytes};
  return gigs_forward(ctx, lite_chunk_alloc, &gc, binningBuffer, binning_user, lite_chunk_alloc, &ic, P, D, M, background, width, height,
                      means3D, shs, colors_precomp, opacities, zeros, zeros + 3 * (size_t)P, zeros + 6 * (size_t)P,
                      zeros + 7 * (size_t)P, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos,
                      tan_fovx, tan_fovy, prefiltered, argmax_depth, 0, out_color, out_opacity, out_depth, spare, spare + 3 * N,
                      spare + 6 * N, spare + 9 * N, spare + 12 * N, spare + 13 * N, radii, debug, stream);
}

int gigs_backward(gigs_ctx* ctx, int P, int D, int M, int R, const float* background, int width, int height,
                  const float* means3D, const float* shs, const float* colors_precomp,
                  const float* normal, const float* albedo, const float* roughness,
                  const float* metallic, const float* scales, const float* rotations,
                  const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                  const float* cam_pos, const int* radii, float scale_modifier, float tan_fovx,
                  float tan_fovy, char* geom_buffer, char* binning_buffer, char* image_buffer,
                  const float* dL_dpix_depth, const float* dL_dpix, const float* dL_dpix_opacity,
                  const float* dL_dpix_normal, const float* dL_dpix_albedo,
                  const float* dL_dpix_roughness, const float* dL_dpix_metallic, float* dL_dmean2D,
                  float* dL_dconic, float* dL_ddepth, float* dL_dopacity, float* dL_dnormal,
                  float* dL_dalbedo, float* dL_droughness, float* dL_dmetallic, float* dL_dcolor,
                  float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                  float* dL_drot, int debug, void* stream) {
  (void)normal; (void)albedo; (void)roughness; (void)metallic;  // already packed in the blend record
  hipStream_t s = (hipStream_t)stream;
  if (P == 0) return 0;
  if (P < 0 || R < 0 || width <= 0 || height <= 0) return fail(GIGS_ERR_INVALID, "bad sizes");
  if (!geom_buffer || !binning_buffer || !image_buffer) return fail(GIGS_ERR_INVALID, "null scratch buffer");
  // gigs_ctx_set_materials_only: the declared stage-2 gradient set -- only these four outputs are written
  unsigned* const materials_only = ctx_of(ctx).materials_only;
  if (ctx_of(ctx).sh_rest && shs && M > 1 && !materials_only)
    return fail(GIGS_ERR_INVALID, "a split-SH forward (gigs_ctx_set_split_sh) has a materials-only backward only");
  if (!dL_dmean2D || !dL_dalbedo || !dL_droughness || !dL_dmetallic)
    return fail(GIGS_ERR_INVALID, "null gradient output");
  if (!materials_only && (!dL_dopacity || !dL_dnormal || !dL_dcolor || !dL_dmean3D || !dL_dcov3D || !dL_dscale || !dL_drot ||
                          (shs && !dL_dsh)))
    return fail(GIGS_ERR_INVALID, "null gradient output");

  gigs::BwdArgs a;
  a.P = P; a.D = D; a.M = M; a.R = R; a.W = width; a.H = height;
  a.gx = (width + GIGS_BLOCK_X - 1) / GIGS_BLOCK_X;
  a.gy = (height + GIGS_BLOCK_Y - 1) / GIGS_BLOCK_Y;
  a.focal_y = height / (2.0f * tan_fovy);
  a.focal_x = width / (2.0f * tan_fovx);
  a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.scale_modifier = scale_modifier;
  a.means3D = means3D; a.shs = shs; a.colors_precomp = colors_precomp; a.scales = scales;
  a.rotations = rotations; a.cov3D_precomp = cov3D_precomp; a.viewmatrix = viewmatrix;
  a.projmatrix = projmatrix; a.cam_pos = cam_pos; a.background = background;
  a.dL_dpix_depth = dL_dpix_depth; a.dL_dpix = dL_dpix; a.dL_dpix_opacity = dL_dpix_opacity;
  a.dL_dpix_normal = dL_dpix_normal; a.dL_dpix_albedo = dL_dpix_albedo;
  a.dL_dpix_roughness = dL_dpix_roughness; a.dL_dpix_metallic = dL_dpix_metallic;
  a.dL_dmean2D = dL_dmean2D; a.dL_dconic = dL_dconic; a.dL_ddepth = dL_ddepth;
  a.dL_dopacity = dL_dopacity; a.dL_dnormal = dL_dnormal; a.dL_dalbedo = dL_dalbedo;
  a.dL_droughness = dL_droughness; a.dL_dmetallic = dL_dmetallic; a.dL_dcolor = dL_dcolor;
  a.dL_dmean3D = dL_dmean3D; a.dL_dcov3D = dL_dcov3D; a.dL_dsh = dL_dsh; a.dL_dscale = dL_dscale;
  a.dL_drot = dL_drot;

  // re-derive the same pointers the forward carved (rasterizer_impl.cu:722-724)
  const size_t N = (size_t)width * height, T = (size_t)a.gx * a.gy;
  char* gp = geom_buffer;
  gigs::GeomState geom = gigs::GeomState::fromChunk(gp, (size_t)P, scan_size_cached(P));
  char* bp = binning_buffer;
  gigs::BinningState bin = gigs::BinningState::fromChunk(bp, (size_t)R, 0);
  char* ip = image_buffer;
  gigs::ImageState img = gigs::ImageState::fromChunk(ip, N, T);
  a.radii = radii ? radii : geom.internal_radii;

  {
    StageScope sc(kBlendBwd, s);
    // geom.grec is zero here: preprocess_fwd clears it and preprocess_bwd clears it again after reading (no fill launch)
    gigs::launch_blend_bwd(a, geom, bin, img, s);
  }
  STAGE_CHECK("render backward");
  {
    StageScope sc(kPreprocessBwd, s);
    gigs::launch_preprocess_bwd(a, geom, ctx_of(ctx).opt.pre_bwd_sh_skip, s, materials_only);
  }
  STAGE_CHECK("preprocess backward");
  return 0;
}

int gigs_mark_visible(int P, const float* means3D, const float* viewmatrix,
                      const float* projmatrix, uint8_t* present, void* stream) {
  (void)projmatrix;
  if (P == 0) return 0;
  if (P < 0 || !means3D || !viewmatrix || !present) return fail(GIGS_ERR_INVALID, "bad argument");
  gigs::launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

int gigs_depth_to_normal(int width, int height, float focal_x, float focal_y,
                         const float* viewmatrix, const float* depth, float* normal,
                         float* depth_pos, void* stream) {
  if (width <= 0 || height <= 0 || !viewmatrix || !depth || !normal || !depth_pos)
    return fail(GIGS_ERR_INVALID, "bad argument");
  StageScope sc(kDepthToNormal, (hipStream_t)stream);
  gigs::launch_depth_to_normal(width, height, focal_x, focal_y, viewmatrix, depth, normal, depth_pos, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

int gigs_derive_normal(int width, int height, float focal_x, float focal_y, const float* viewmatrix, const float* depth,
                       float sigma_color, float sigma_x, float sigma_y, float* normal_from_depth, float* depth_pos_filter,
                       void* stream) {
  if (width <= 1 || height <= 1 || !viewmatrix || !depth || !normal_from_depth || !depth_pos_filter)
    return fail(GIGS_ERR_INVALID, "derive_normal: bad argument (images must be larger than 1x1)");
  StageScope sc(kDepthToNormal, (hipStream_t)stream);
  gigs::launch_derive_normal_fused(width, height, focal_x, focal_y, viewmatrix, sigma_color, sigma_x, sigma_y, depth,
                                   normal_from_depth, depth_pos_filter, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

size_t gigs_gi_scratch_bytes(int width, int height) {
  if (width <= 0 || height <= 0) return 0;
  return gigs::gi_scratch_bytes(width, height);
}

int gigs_ssao_ex(gigs_ctx* ctx, int width, int height, float focal_x, float focal_y, float radius, float bias,
                 float thick, float delta, int step, int start, const float* normal_view,
                 const float* pos, float* occlusion, void* scratch, void* stream) {
  if (width <= 0 || height <= 0 || !normal_view || !pos || !occlusion) return fail(GIGS_ERR_INVALID, "bad argument");
  if (width >= (1 << 15) || height >= (1 << 15)) return fail(GIGS_ERR_INVALID, "image side above 32767 pixels");
  StageScope sc(kSsao, (hipStream_t)stream);
  const int rc = gigs::launch_ssao(ctx_of(ctx).opt, width, height, focal_x, focal_y, radius, bias, thick, delta, step, start,
                                   normal_view, pos, occlusion, scratch, (hipStream_t)stream);
  if (rc == -1) return fail(GIGS_ERR_INVALID, "delta=%g gives an unbounded or oversized ray set", (double)delta);
  if (rc) return fail(GIGS_ERR_HIP, "ray table upload failed");
  HIP_TRY(hipGetLastError());
  return 0;
}
int gigs_ssao(int width, int height, float focal_x, float focal_y, float radius, float bias,
              float thick, float delta, int step, int start, const float* normal_view,
              const float* pos, float* occlusion, void* stream) {
  return gigs_ssao_ex(nullptr, width, height, focal_x, focal_y, radius, bias, thick, delta, step, start, normal_view, pos,
                      occlusion, nullptr, stream);
}

int gigs_ssr_ex(gigs_ctx* ctx, int width, int height, float focal_x, float focal_y, float radius, float bias,
                float thick, float delta, int step, int start, const float* normal_view,
                const float* pos, const float* rgb, const float* albedo, const float* roughness,
                const float* metallic, const float* F0, float* color, float* abd, void* scratch, void* stream) {
  if (width <= 0 || height <= 0 || !normal_view || !pos || !rgb || !albedo || !metallic || !F0 || !color || !abd)
    return fail(GIGS_ERR_INVALID, "bad argument");
  if (width >= (1 << 15) || height >= (1 << 15)) return fail(GIGS_ERR_INVALID, "image side above 32767 pixels");
  StageScope sc(kSsr, (hipStream_t)stream);
  const int rc = gigs::launch_ssr(ctx_of(ctx).opt, width, height, focal_x, focal_y, radius, bias, thick, delta, step, start,
                                  normal_view, pos, rgb, albedo, roughness, metallic, F0, color, abd, scratch,
                                  (hipStream_t)stream);
  if (rc == -1) return fail(GIGS_ERR_INVALID, "delta=%g gives an unbounded or oversized ray set", (double)delta);
  if (rc) return fail(GIGS_ERR_HIP, "ray table upload failed");
  HIP_TRY(hipGetLastError());
  return 0;
}
int gigs_ssr_hits(gigs_ctx* ctx, int width, int height, float focal_x, float focal_y, float radius, float bias, float thick,
                  float delta, int step, int start, const float* normal_view, const float* pos, const float* rgb,
                  const float* albedo, const float* roughness, const float* metallic, const float* F0, float* color, float* abd,
                  int mode, unsigned* counts, const unsigned* offsets, void* entries, unsigned capacity, void* scratch,
                  void* stream) {
  if (width <= 0 || height <= 0 || !normal_view || !pos || !rgb || !albedo || !metallic || !F0 || !color || !abd)
    return fail(GIGS_ERR_INVALID, "bad argument");
  if (width >= (1 << 15) || height >= (1 << 15)) return fail(GIGS_ERR_INVALID, "image side above 32767 pixels");
  if ((mode != 1 && mode != 2) || (mode == 1 && !counts) || (mode == 2 && (!offsets || (!entries && capacity > 0))))
    return fail(GIGS_ERR_INVALID, "ssr_hits: mode 1 needs counts, mode 2 offsets and entries");
  StageScope sc(kSsr, (hipStream_t)stream);
  const int rc = gigs::launch_ssr(ctx_of(ctx).opt, width, height, focal_x, focal_y, radius, bias, thick, delta, step, start,
                                  normal_view, pos, rgb, albedo, roughness, metallic, F0, color, abd, scratch,
                                  (hipStream_t)stream, mode, counts, offsets, entries, capacity);
  if (rc == -1) return fail(GIGS_ERR_INVALID, "delta=%g gives an unbounded or oversized ray set", (double)delta);
  if (rc == -3) return fail(GIGS_ERR_INVALID, "ssr_hits: the hit list is recorded by the default march only (gi_march = proj, start < step)");
  if (rc) return fail(GIGS_ERR_HIP, "ray table upload failed");
  HIP_TRY(hipGetLastError());
  return 0;
}

int gigs_ssr_apply(int width, int height, float delta, const unsigned* offsets, const void* entries, const float* normal_view,
                   const float* pos, const float* rgb, const float* albedo, const float* metallic, const float* F0,
                   float* color, float* abd, void* stream) {
  if (width <= 0 || height <= 0 || !offsets || !normal_view || !pos || !rgb || !albedo || !metallic || !F0 || !color || !abd)
    return fail(GIGS_ERR_INVALID, "bad argument");
  StageScope sc(kSsr, (hipStream_t)stream);
  const int rc = gigs::launch_ssr_apply(width, height, delta, offsets, entries, normal_view, pos, rgb, albedo, metallic, F0, color,
                                        abd, (hipStream_t)stream);
  if (rc == -1) return fail(GIGS_ERR_INVALID, "delta=%g gives an unbounded or oversized ray set", (double)delta);
  if (rc) return fail(GIGS_ERR_HIP, "ray table upload failed");
  HIP_TRY(hipGetLastError());
  return 0;
}

int gigs_ssr(int width, int height, float focal_x, float focal_y, float radius, float bias,
             float thick, float delta, int step, int start, const float* normal_view,
             const float* pos, const float* rgb, const float* albedo, const float* roughness,
             const float* metallic, const float* F0, float* color, float* abd, void* stream) {
  return gigs_ssr_ex(nullptr, width, height, focal_x, focal_y, radius, bias, thick, delta, step, start, normal_view, pos, rgb,
                     albedo, roughness, metallic, F0, color, abd, nullptr, stream);
}

// Gaussian_SSR's backward as the reference's autograd function computes it (R/diff_gaussian_rasterization/__init__.py:671-693):
// grad_albedo = grad_color * abd, nothing to roughness / metallic / F0 (the CUDA `SSR_BACKWARD` the reference also exports is
// unreachable -- its call is commented out -- and is not what this entry restates).
namespace gigs {
__global__ void __launch_bounds__(256) ssr_backward_kernel(size_t n, const float* __restrict__ g, const float* __restrict__ abd,
                                                           float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = g[i] * abd[i];
}
}  // namespace gigs
int gigs_ssr_backward(int width, int height, const float* grad_color, const float* abd, float* grad_albedo, void* stream) {
  if (width <= 0 || height <= 0 || !grad_color || !abd || !grad_albedo) return fail(GIGS_ERR_INVALID, "bad argument");
  const size_t n = 3 * (size_t)width * height;
  hipLaunchKernelGGL(gigs::ssr_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, grad_color,
                     abd, grad_albedo);
  HIP_TRY(hipGetLastError());
  return 0;
}

int gigs_median3x3(int channels, int height, int width, const float* in, float* out, void* stream) {
  if (channels <= 0 || width <= 0 || height <= 0 || !in || !out) return fail(GIGS_ERR_INVALID, "bad argument");
  StageScope sc(kMedian, (hipStream_t)stream);
  gigs::launch_median3x3(channels, height, width, in, out, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

int gigs_median3x3_backward(int channels, int height, int width, const float* in,
                            const float* grad_out, float* grad_in, void* stream) {
  if (channels <= 0 || width <= 0 || height <= 0 || !in || !grad_out || !grad_in) return fail(GIGS_ERR_INVALID, "bad argument");
  StageScope sc(kMedianBwd, (hipStream_t)stream);
  gigs::launch_median3x3_bwd(channels, height, width, in, grad_out, grad_in, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

int gigs_bilateral3x3(int channels, int height, int width, float sigma_color, float sigma_x,
                      float sigma_y, const float* in, float* out, void* stream) {
  if ((channels != 1 && channels != 3) || width <= 1 || height <= 1 || !in || !out)
    return fail(GIGS_ERR_INVALID, "bilateral3x3 supports 1 or 3 channels and images larger than 1x1");
  StageScope sc(kBilateral, (hipStream_t)stream);
  gigs::launch_bilateral3x3(channels, height, width, sigma_color, sigma_x, sigma_y, in, out, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"
