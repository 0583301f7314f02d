/* gigs_hip.h -- C ABI of libgigs_hip.so, the MI355X (gfx950) implementation of the GI-GS
 * rasterizer hot path.
 *
 * Every entry point replaces one static method of `CudaRasterizer::Rasterizer`
 * (reference: submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:20-199,
 * "R/" below) or one kornia call the reference makes inside the operator.  All pointers
 * are DEVICE pointers owned by the caller unless a parameter says "host".  Nothing is
 * allocated or freed by the library; scratch memory is requested from the caller through
 * `gigs_alloc_fn` callbacks, the C form of the reference's
 * `std::function<char*(size_t)>` resize functors (R/rasterize_points.cu:31-37).
 *
 * Conventions
 *   - return value: >= 0 on success (gigs_forward returns num_rendered), < 0 on error;
 *     `gigs_last_error()` then returns a message (thread-local, host memory).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  The reference
 *     launches on the legacy default stream; the stream argument is the only addition.
 *   - image planes are planar CHW fp32 (R/cuda_rasterizer/forward.cu:608-610);
 *     `viewmatrix` / `projmatrix` are the 16 floats of the row-vector matrices the
 *     reference passes (scene/cameras.py:75-85).
 *   - optional inputs (shs / colors_precomp, scales+rotations / cov3D_precomp) are NULL
 *     when absent, like the null data_ptr of the reference's empty tensors
 *     (R/diff_gaussian_rasterization/__init__.py:435-445).
 */
#ifndef GIGS_HIP_H_
#define GIGS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIGS_OK 0
#define GIGS_ERR_INVALID (-1)  /* bad argument (shape, null pointer, non-RGB without colours) */
#define GIGS_ERR_HIP (-2)      /* a HIP call or kernel failed; see gigs_last_error()           */
#define GIGS_ERR_ALLOC (-3)    /* an allocation callback returned NULL                        */

/* Replaces std::function<char*(size_t)> (R/rasterize_points.cu:31-37): must return a device
 * pointer to at least `nbytes` bytes that stays valid until the matching backward has run. */
typedef char* (*gigs_alloc_fn)(size_t nbytes, void* user);

const char* gigs_last_error(void);
/* "gfx950" -- the only architecture the code objects are built for. */
const char* gigs_build_arch(void);

/* ---- contexts: per-instance state -----------------------------------------------------------------
 * The reference's rasterizer is a set of stateless static methods (R/cuda_rasterizer/rasterizer.h:20-199) that
 * all run on the legacy default stream.  This library adds streams, an asynchronous binning mode, a
 * scheduling event and a number of tuning / diagnostic switches -- and keeps ALL of that in a context the
 * caller owns, so that two rasterizer instances (or two streams) of one process do not see each other:
 * every entry point that reads such state takes a `gigs_ctx*` first.  NULL = the default context: the
 * options below as the environment gave them when the library was first used (parsed ONCE, immutable
 * afterwards), no asynchronous binning, no event.  Calls only READ their context, so any number of threads
 * and streams may use one concurrently; the gigs_ctx_set_* functions must not race with calls that use it.
 * Different contexts are independent (what the library itself caches process-wide -- ray tables, texel
 * tables, temp-storage sizes -- is keyed by its inputs and never modified once built).
 * The in-library profile session (gigs_profile_begin/end, a bench.py diagnostic) is the one process-wide
 * facility left. */
typedef struct gigs_ctx gigs_ctx;
typedef struct gigs_options {
  int struct_bytes;     /* sizeof(gigs_options): set by the caller (ABI growth) */
  int binning_legacy;   /* 0 (default) tile-bucketed binning; 1 scan / duplicate / global radix sort / tile ranges, the
                           reference's structure (identical keys, point_list, ranges).        env GIGS_BINNING=legacy */
  int bucket_max_mean;  /* mean instances per tile above which a scene counts as dense (2500). env GIGS_BUCKET_MAX_MEAN */
  int long_lists;       /* -1 (default) by density, 0 / 1 forbid / force the long-list partition.  env GIGS_LONG_LISTS */
  int bucket_target;    /* keys per bucket of that partition (1536).                            env GIGS_BUCKET_TARGET */
  int bin_bands;        /* passes of the by-tile scatter over bands of tile rows: 0 (default) = 4 for dense scenes, else 1;
                           same keys, point_list, ranges.                                          env GIGS_BIN_BANDS */
  int blend_cull;       /* 1 (default); 0 = blend forward without the quadrant cull (same bits). env GIGS_BLEND_CULL */
  int pre_bwd_sh_skip;  /* 1 (default); 0 = the preprocess backward evaluates every chain.   env GIGS_PRE_BWD_SH_SKIP */
  int gi_march;         /* SSAO / SSR march: 0 exact (the oracle's pixel choices bit for bit), 1 hoist, 2 hoist_fma,
                           3 proj_nr, 4 proj (default).              env GIGS_GI_MARCH=exact|hoist|hoist_fma|proj_nr|proj */
  int gi_cert;          /* 1 (default) coarse-depth certification in front of marches 3 / 4 (same bits). env GIGS_GI_CERT */
  int gi_interleave;    /* 1 (default) interleaved ray pairs per wave, 0 contiguous quarters.     env GIGS_GI_INTERLEAVE */
  int gi_tile_log2w;    /* pixel rectangle of a march workgroup, 2^k x 64/2^k, k = 0..6 (3).      env GIGS_GI_TILE_LOG2W */
  int gi_zero_rays;     /* 0 (default) the zero-weight rays (theta = 0: 32 of 512 at delta 0.0625) are not marched -- their
                           contributions are exact zeros; 1 = marched all the same (same bits).   env GIGS_GI_ZERO_RAYS */
  int spec_max8;        /* largest mean GGX window served by 8-lane groups (128).                   env GIGS_SPEC_MAX8 */
  int spec_max16;       /* ... by 16-lane groups (1500).                                            env GIGS_SPEC_MAX16 */
  int shade_lds_floats; /* LDS accumulator budget of the shade backward in floats (30720).    env GIGS_SHADE_LDS_FLOATS */
  int shade_bwd_blocks; /* its persistent workgroups, 0 (default) = one per CU.               env GIGS_SHADE_BWD_BLOCKS */
} gigs_options;
/* A new context holds a copy of the default options.  Destroying a context frees host memory only; work queued
 * with it may still be running. */
gigs_ctx* gigs_ctx_create(void);
void gigs_ctx_destroy(gigs_ctx* ctx);
/* out->struct_bytes must be set by the caller; ctx == NULL reads the defaults.  set: values outside their range are
 * rejected (GIGS_ERR_INVALID) and nothing changes; ctx must not be NULL (the default context is immutable). */
int gigs_ctx_get_options(const gigs_ctx* ctx, gigs_options* out);
int gigs_ctx_set_options(gigs_ctx* ctx, const gigs_options* in);

/* Scratch sizes: required<GeometryState>(P), required<ImageState>(N), required<BinningState>(R)
 * (R/cuda_rasterizer/rasterizer_impl.h:67-73).  Need a visible GPU (rocPRIM temp-storage
 * queries). Return 0 and set the error string on failure. */
size_t gigs_required_geom(int P);
size_t gigs_required_image(int width, int height);
size_t gigs_required_binning(int num_rendered);

/* Rasterizer::forward (R/cuda_rasterizer/rasterizer.h:24-63, rasterizer_impl.cu:486-672).
 * Runs preprocess -> scan -> (one 4-byte D2H read of num_rendered) -> duplicate -> radix sort
 * -> tile ranges -> G-buffer blend.  `background` is 3 floats.  `radii` may be NULL.
 * Every pixel of every output plane and every radii[i] is written when P > 0; when P == 0 nothing
 * is launched and the caller's (zero) initialisation stays. */
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
                 int* radii, int debug, void* stream);

/* Rasterizer::backward (rasterizer.h:103-150, rasterizer_impl.cu:676-803).  Every element of every
 * dL_d* output is written (culled Gaussians get zeros), so the caller need not zero-initialise them
 * as the reference binding does (R/rasterize_points.cu:299-312); dL_dconic [P,2,2] and dL_ddepth [P]
 * are the two scratch gradients the reference allocates but does not return: either may be NULL (not
 * written).  Any of the seven incoming dL_dpix_* planes may be NULL = an all-zero gradient (an output
 * the loss does not use). */
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
                  float* dL_drot, int debug, void* stream);

/* Rasterizer::markVisible (rasterizer.h:24-29 / rasterizer_impl.cu:141-153); present = bool[P]. */
int gigs_mark_visible(int P, const float* means3D, const float* viewmatrix,
                      const float* projmatrix, uint8_t* present, void* stream);

/* Rasterizer::depthToNormal (rasterizer_impl.cu:201-220 -> forward.cu:914-1032).  Every pixel of
 * `normal` and `depth_pos` ([3,H,W]) is written (zeros where the reference leaves its zero fill). */
int gigs_depth_to_normal(int width, int height, float focal_x, float focal_y,
                         const float* viewmatrix, const float* depth, float* normal,
                         float* depth_pos, void* stream);

/* Rasterizer::SSAO (rasterizer_impl.cu:222-253 -> forward.cu:635-724). occlusion = [1,H,W], fully written. */
/* The chain GaussianRasterizer.forward runs between the blend and SSAO (R/diff_gaussian_rasterization/__init__.py:475-517)
 * as ONE launch: normal_from_depth = bilateral3x3(depth_to_normal(median3x3(depth)).normal, sigma_color, sigma_x, sigma_y),
 * depth_pos_filter = median3x3(depth_to_normal(median3x3(depth)).pos).  Bit-identical to calling gigs_median3x3,
 * gigs_depth_to_normal, gigs_bilateral3x3 and gigs_median3x3 in turn (gigs-hip extension; depth [1,H,W], outputs [3,H,W]). */
int gigs_derive_normal(int width, int height, float focal_x, float focal_y, const float* viewmatrix, const float* depth,
                       float sigma_color, float sigma_x, float sigma_y, float* normal_from_depth, float* depth_pos_filter,
                       void* stream);
int gigs_ssao(int width, int height, float focal_x, float focal_y, float radius, float bias,
              float thick, float delta, int step, int start, const float* normal_view,
              const float* pos, float* occlusion, void* stream);

/* Rasterizer::SSR (rasterizer_impl.cu:255-298 -> forward.cu:726-909). color, abd = [3,H,W]. */
int gigs_ssr(int width, int height, float focal_x, float focal_y, float radius, float bias,
             float thick, float delta, int step, int start, const float* normal_view,
             const float* pos, const float* rgb, const float* albedo, const float* roughness,
             const float* metallic, const float* F0, float* color, float* abd, void* stream);

/* The same two passes with a caller-owned scratch buffer of gigs_gi_scratch_bytes(width, height) bytes (gigs-hip
 * extension; may be NULL = identical to the plain entries).  With it the default march first builds a min/max table
 * of the position plane's z over 16..64-pixel blocks there and skips, conservatively, the z-plane lookups of samples
 * that cannot hit: bit-identical outputs, ~92 % fewer gathers on the bench view.  The buffer is
 * only used during the call's kernels (stream-ordered).  `ctx` selects the march (gigs_options.gi_*; NULL = defaults);
 * gigs_ssao / gigs_ssr are these with ctx = NULL and scratch = NULL. */
size_t gigs_gi_scratch_bytes(int width, int height);
int gigs_ssao_ex(gigs_ctx* ctx, int width, int height, float focal_x, float focal_y, float radius, float bias,
                 float thick, float delta, int step, int start, const float* normal_view,
                 const float* pos, float* occlusion, void* scratch, void* stream);
int gigs_ssr_ex(gigs_ctx* ctx, int width, int height, float focal_x, float focal_y, float radius, float bias,
                float thick, float delta, int step, int start, const float* normal_view,
                const float* pos, const float* rgb, const float* albedo, const float* roughness,
                const float* metallic, const float* F0, float* color, float* abd, void* scratch, void* stream);

/* The hit list of the indirect-light march (gigs-hip extension; frozen-geometry reuse).  WHICH pixel each ray of each pixel
 * hits depends on normals and positions only (forward.cu:796-829), not on the radiance gathered there, so a view whose geometry
 * does not change marches once and gathers afterwards.  gigs_ssr_hits is gigs_ssr_ex (same outputs) that additionally,
 *   mode 1: writes counts[4 * pixel + w] = the number of hits of pixel's rays handled by wave w of its workgroup (u32 [4 N]);
 *   mode 2: given offsets = the exclusive prefix of those counts (u32 [4 N + 1]), writes the hits in march order as
 *           {hit pixel, ray index} pairs (2 x u32 each) at offsets[...]; pairs beyond `capacity` are dropped -- compare
 *           offsets[4 N] with it.
 * gigs_ssr_apply evaluates color / abd from such a list: per pixel the four sequences are summed in order, combined as the
 * march combines its four waves, and the same tail applied -- the march's outputs bit for bit while normals / positions are the
 * recorded ones.  Recorded by the default march only (gigs_options.gi_march = 4). */
int gigs_ssr_hits(gigs_ctx* ctx, int width, int height, float focal_x, float focal_y, float radius, float bias, float thick,
                  float delta, int step, int start, const float* normal_view, const float* pos, const float* rgb,
                  const float* albedo, const float* roughness, const float* metallic, const float* F0, float* color, float* abd,
                  int mode, unsigned* counts, const unsigned* offsets, void* entries, unsigned capacity, void* scratch,
                  void* stream);
int gigs_ssr_apply(int width, int height, float delta, const unsigned* offsets, const void* entries, const float* normal_view,
                   const float* pos, const float* rgb, const float* albedo, const float* metallic, const float* F0,
                   float* color, float* abd, void* stream);

/* kornia.filters.median_blur(x[None], (3,3))[0] as called at
 * R/diff_gaussian_rasterization/__init__.py:478, 504 (zero padding, NaN-propagating). */
int gigs_median3x3(int channels, int height, int width, const float* in, float* out, void* stream);
/* Backward of the above: routes each output gradient to the tap that was selected. */
int gigs_median3x3_backward(int channels, int height, int width, const float* in,
                            const float* grad_out, float* grad_in, void* stream);

/* kornia.filters.bilateral_blur(x[None], (3,3), sigma_color, (sigma_y, sigma_x))[0]
 * (…/__init__.py:491; reflect border, L1 colour distance). */
int gigs_bilateral3x3(int channels, int height, int width, float sigma_color, float sigma_x,
                      float sigma_y, const float* in, float* out, void* stream);

/* ---- deferred shade and cubemap light (pbr/shade.py, pbr/light.py, pbr/renderutils) ----------
 * Cubemaps are [6, res, res, 3] fp32 (NHWC as in the reference); HWC image tensors as passed to
 * pbr_shading.  Sampling rule of the texture lookups (the reference uses nvdiffrast dr.texture,
 * third party, parity unpinned): cube face/(u,v) selection by the dominant axis with the face
 * orientations of pbr/light.py:40-52; bilinear taps at (uv * size - 0.5); a tap that leaves a face
 * is taken from the neighbouring face, the single missing tap at a cube corner is dropped and the
 * weights renormalised; 2-D lookups clamp tap indices; the mip blend uses
 * level = clamp(mip_level_bias, 0, L-1) between floor(level) and floor(level)+1. */

/* diffuse_cubemap_fwd / _bwd (pbr/renderutils/c_src/torch_bindings.cpp:740-795 -> cubemap.cu:110-169).
 * The backward is a gather over the same weights; grad_cubemap is fully overwritten. */
int gigs_diffuse_cubemap_fwd(int res, const float* cubemap, float* out, void* stream);
int gigs_diffuse_cubemap_bwd(int res, const float* grad_out, float* grad_cubemap, void* stream);
/* specular_bounds (torch_bindings.cpp:797-822 -> cubemap.cu:181-244): bounds = [6,res,res,24]. */
int gigs_specular_bounds(int res, float costheta_cutoff, float* bounds, void* stream);
/* specular_cubemap_fwd / _bwd (torch_bindings.cpp:824-890 -> cubemap.cu:246-350).  out and grad_out
 * are [6,res,res,4] (rgb, weight sum); grad_cubemap [6,res,res,3] is fully overwritten. */
int gigs_specular_cubemap_fwd(int res, const float* cubemap, const float* bounds, float roughness,
                              float costheta_cutoff, float* out, void* stream);
int gigs_specular_cubemap_bwd(int res, const float* bounds, const float* grad_out, float roughness,
                              float costheta_cutoff, float* grad_cubemap, void* stream);
/* The same filter through a cached weight table (an MI355X-specific addition, no reference
 * counterpart): the pair weights depend only on (res, roughness, cutoff), so they are computed once
 * into `weights` and the per-step filter becomes a streaming pass.  offsets = uint32 [6*res*res*6]
 * exclusive prefix sums of the per-(texel, face) AABB areas of `bounds`; weights holds that many floats
 * (+ the last area).  swap_roles = 0 builds the table the forward reads, 1 the one the backward reads
 * (the NDF argument V.H is evaluated with the reference's operand roles in both, so results are
 * bit-identical to the table-free entry points up to summation order). */
int gigs_specular_weights(int res, const float* bounds, const uint32_t* offsets, float roughness,
                          float costheta_cutoff, int swap_roles, float* weights, void* stream);
/* out_weights[o][i] = weights[o][i] / texel_divisor[texel i of o's window] (negative markers kept), texel_divisor
 * [6,res,res].  Applied to the role-swapped table with the forward's weight sums it yields a table whose gather of the
 * incoming gradient IS the backward of the normalised filter (no separate division by wsum). */
int gigs_specular_weights_divide(int res, const float* bounds, const uint32_t* offsets, const float* weights,
                                 const float* texel_divisor, float* out_weights, void* stream);
/* wsum_out == NULL: out = [6,res,res,4] (rgb, weight sum) like the table-free entry point;
 * wsum_out != NULL: out = [6,res,res,3] = rgb / wsum (the division of ops.py:458 folded in) and the
 * weight sums go to wsum_out [6,res,res].  grad_is_rgb: grad_out is [6,res,res,3] (already divided
 * by wsum by the caller) instead of [6,res,res,4].  avg_window = table length / (6 res^2), the mean
 * number of candidates per texel: a scheduling hint only (lanes per texel), 0 = unknown. */
int gigs_specular_cubemap_fwd_w(gigs_ctx* ctx, int res, const float* cubemap, const float* bounds, const uint32_t* offsets,
                                const float* weights, int avg_window, float* out, float* wsum_out, void* stream);
int gigs_specular_cubemap_bwd_w(gigs_ctx* ctx, int res, const float* bounds, const uint32_t* offsets,
                                const float* weights_swapped, int avg_window, const float* grad_out,
                                int grad_is_rgb, float* grad_cubemap, void* stream);

/* The table-driven GGX filter of ALL levels of a light in one launch (gigs-hip extension): the levels of
 * CubemapLight.build_mips (pbr/light.py:166-170) are independent of each other, so their texels share one grid instead of
 * one dependent launch per level.  `levels` is a HOST array; per level the arguments of gigs_specular_cubemap_fwd_w
 * (backward = 0: src = the level's mip, dst = rgb / wsum [6,res,res,3], wsum [6,res,res] required) or of
 * gigs_specular_cubemap_bwd_w with grad_is_rgb = 1 (backward = 1: src = the incoming gradient, weights = the role-swapped
 * table already divided by the forward's weight sums, dst = the gradient w.r.t. the mip, wsum unused). */
typedef struct gigs_spec_level {
  int res, avg_window;
  const float* src;
  const float* bounds;
  const uint32_t* offsets;
  const float* weights;
  float* dst;
  float* wsum;
} gigs_spec_level;
int gigs_specular_cubemap_multi_w(gigs_ctx* ctx, int n_levels, const gigs_spec_level* levels, int backward, void* stream);
/* cubemap_mip (pbr/light.py:54-79): forward 2x2 average pool [6,2r,2r,C] -> [6,r,r,C]; backward =
 * bilinear cube lookup of 0.25*dout at every fine texel direction, dout [6,r,r,3] -> din [6,2r,2r,3]. */
int gigs_cubemap_mip_fwd(int res_out, int channels, const float* in, float* out, void* stream);
int gigs_cubemap_mip_bwd(int res_out, const float* dout, float* din, void* stream);
/* din = add + the above (add [6,2r,2r,3]: the gradient the fine level receives from its other consumer, so that the
 * chain base <- mip1 <- ... <- mipN is walked back without separate accumulation passes). */
int gigs_cubemap_mip_bwd_add(int res_out, const float* dout, const float* add, float* din, void* stream);
/* the same with a second gradient of the coarse level: din = add + lookup(0.25 * (dout + dout2)); dout2 and add may be
 * NULL.  (The coarsest level feeds the GGX filter and the diffuse filter, the base the GGX filter and the chain.) */
int gigs_cubemap_mip_bwd_add2(int res_out, const float* dout, const float* dout2, const float* add, float* din, void* stream);

/* pbr_shading (pbr/shade.py:108-241) fused into one kernel.  normals/view_dirs/albedo [H,W,3],
 * roughness/occlusion/metallic [H,W,1] (occlusion, metallic, background may be NULL), mask = bool
 * [H,W,1]; diffuse = light.diffuse [6,dres,dres,3]; spec = HOST array of n_levels device pointers
 * (light.specular), spec_res their resolutions (host); lut = BRDF LUT [lut_h, lut_w, 2].
 * Outputs [H,W,3]: render_rgb, diffuse_rgb, specular_rgb, diffuse_light (the result dict). */
int gigs_shade_fwd(int H, int W, const float* normals, const float* view_dirs, const float* albedo,
                   const float* roughness, const uint8_t* mask, const float* occlusion,
                   const float* metallic, const float* background, const float* diffuse, int diffuse_res,
                   int n_levels, const float* const* spec, const int* spec_res, const float* lut,
                   int lut_w, int lut_h, int tone, int gamma, float* render_rgb, float* diffuse_rgb,
                   float* specular_rgb, float* diffuse_light, void* stream);
/* Backward of the above.  g_* are gradients w.r.t. the four outputs (any may be NULL = zero).
 * d_albedo [H,W,3], d_roughness [H,W,1], d_metallic [H,W,1] (NULL when metallic is NULL) are
 * overwritten; d_diffuse and d_spec[i] (host array of device pointers, entries may be NULL) are
 * ACCUMULATED with float atomics and must be zero-initialised by the caller. */
int gigs_shade_bwd(int H, int W, const float* normals, const float* view_dirs, const float* albedo,
                   const float* roughness, const uint8_t* mask, const float* occlusion,
                   const float* metallic, const float* diffuse, int diffuse_res, int n_levels,
                   const float* const* spec, const int* spec_res, const float* lut, int lut_w, int lut_h,
                   int tone, int gamma, const float* g_render, const float* g_diffuse_rgb,
                   const float* g_specular_rgb, const float* g_diffuse_light, float* d_albedo,
                   float* d_roughness, float* d_metallic, float* d_diffuse, float* const* d_spec,
                   void* stream);

/* Stage-2 fusion hooks for the two shade entry points (gigs-hip extension; NULL = the plain operator).
 * They fold the tensor glue train.py wraps around pbr_shading into the same kernels:
 *   planar         1: normals, albedo and every [H,W,3] output / gradient are [3,H,W] planes, i.e. the
 *                  rasterizer's own layout (no permute + copy);  view_dirs stays [H,W,3]
 *   rough_scale/bias  roughness = raw * scale + bias (train.py:293-295); d_roughness is w.r.t. raw
 *   forward extras (may be NULL, output layout): out_F0 = (1-m)*0.04 + albedo*m (train.py:352-356),
 *                  out_linear = srgb_to_linear(render_rgb) (train.py:70-81), out_roughness [H,W]
 *   backward extras (may be NULL): d_albedo += g_albedo_mul_a * g_albedo_mul_b (Gaussian_SSR's closed-form
 *                  backward grad_out * abd, R/diff_gaussian_rasterization/__init__.py:671-673);
 *                  g_roughness_add / g_metallic_add [H,W] are added to the roughness (remapped) / metallic
 *                  gradients (the lamb regulariser, train.py:401-402);
 *                  g_scale (device scalar, NULL = 1): g_render and g_albedo_mul_a are gradients for a UNIT upstream
 *                  gradient (gigs_stage2_loss_fwd_grad writes them in the forward) and are multiplied by it here;
 *                  lamb_mask [H,W] + lamb_acc4 (gigs_stage2_loss_fwd's acc4): the lamb regulariser's gradients
 *                  -/+ mask / acc4[3] * 0.001 * g_scale are formed here instead of being read from g_*_add;
 *                  part: see the member.
 * With ext != NULL diffuse_rgb / specular_rgb / diffuse_light may be NULL (not written). */
typedef struct gigs_shade_ext {
  int planar;
  float rough_scale, rough_bias;
  float *out_F0, *out_linear, *out_roughness;
  const float *g_albedo_mul_a, *g_albedo_mul_b, *g_roughness_add, *g_metallic_add;
  const float *g_scale, *lamb_mask, *lamb_acc4;
  int part; /* gigs_shade_bwd_ex: 0 = all gradients; 1 = d_albedo / d_roughness / d_metallic only (d_diffuse / d_spec are not
               touched); 2 = d_diffuse / d_spec only (the material outputs may be NULL).  The two parts of one backward may run
               on different streams: the material gradients feed the rasterizer's backward, the light's feed its filters'. */
} gigs_shade_ext;
int gigs_shade_fwd_ex(gigs_ctx* ctx, int H, int W, const float* normals, const float* view_dirs, const float* albedo,
                      const float* roughness, const uint8_t* mask, const float* occlusion,
                      const float* metallic, const float* background, const float* diffuse, int diffuse_res,
                      int n_levels, const float* const* spec, const int* spec_res, const float* lut,
                      int lut_w, int lut_h, int tone, int gamma, float* render_rgb, float* diffuse_rgb,
                      float* specular_rgb, float* diffuse_light, const gigs_shade_ext* ext, void* stream);
int gigs_shade_bwd_ex(gigs_ctx* ctx, int H, int W, const float* normals, const float* view_dirs, const float* albedo,
                      const float* roughness, const uint8_t* mask, const float* occlusion,
                      const float* metallic, const float* diffuse, int diffuse_res, int n_levels,
                      const float* const* spec, const int* spec_res, const float* lut, int lut_w, int lut_h,
                      int tone, int gamma, const float* g_render, const float* g_diffuse_rgb,
                      const float* g_specular_rgb, const float* g_diffuse_light, float* d_albedo,
                      float* d_roughness, float* d_metallic, float* d_diffuse, float* const* d_spec,
                      const gigs_shade_ext* ext, void* stream);

/* The rest of the stage-2 tensor glue as single passes (gigs-hip extension; all planes [C,H,W] fp32).
 * gigs_gbuffer_post = gaussian_renderer/__init__.py:157-199 for normal_map and out_normal_view:
 *   normal_mask = (normal_map != 0).all(0)  (u8 and/or f32 copies, either may be NULL);
 *   normals_view = -(median3x3(normalize_where(normal_map)) @ viewmatrix[:3,:3]);
 *   out_normal_view_filtered = median3x3(normalize_where(out_normal_view)).  No gradient (detached in stage 2).
 * gigs_stage2_loss_fwd = train.py:382-402: render_rgb = render_direct + median3x3(linear_to_srgb(irr));
 *   loss = mean|render_rgb - gt| + 0.001 * (mean_mask(1 - roughness) + mean_mask(metallic)); acc4 is a
 *   scratch of GIGS_STAGE2_ACC_FLOATS floats whose first four receive {sum|.|, sum(1-r)m, sum(metallic m),
 *   sum m} (kept for the backward; the rest holds per-workgroup partial sums), render_rgb may be NULL.
 * gigs_stage2_loss_bwd: gradients of that loss (times *g_loss, NULL = 1) w.r.t. render_direct, irr (through the
 *   median's tap selection and the sRGB curve; overwritten), roughness and metallic [H,W]. */
#define GIGS_STAGE2_ACC_FLOATS (4 + 4 * 256)
int gigs_gbuffer_post(int height, int width, const float* normal_map, const float* out_normal_view,
                      const float* viewmatrix, float* normals_view, uint8_t* normal_mask, float* normal_mask_f,
                      float* out_normal_view_filtered, void* stream);
/* Stage 1 differentiates through the normal post-processing (train.py:327-328 use render()'s normal_map):
 * gigs_gbuffer_post_bwd = gradient of gigs_gbuffer_post's normals_view w.r.t. normal_map (rotation, the median's tap
 *   selection -- first tap in row-major order equal to the median, as gigs_median3x3_backward --, normalize_where);
 *   scratch3 is [3,H,W] floats of scratch (zeroed inside), g_normal_map [3,H,W] is overwritten.
 * gigs_normalize_mask = gaussian_renderer/__init__.py:157-163 for normal_map_from_depth: mask = (in != 0).all(0)
 *   (u8 [H,W], may be NULL), out = normalize_where(in). */
int gigs_gbuffer_post_bwd(int height, int width, const float* normal_map, const float* viewmatrix,
                          const float* g_normals_view, float* scratch3, float* g_normal_map, void* stream);
int gigs_normalize_mask(int height, int width, const float* in, float* out, uint8_t* mask, void* stream);
/* mask [H,W] (floats 0 / 1) = (in != 0).all(0) of in [3,H,W]: normal_mask of gaussian_renderer/__init__.py:158 as the
 * masked TV loss (train.py:116-142) weighs with it. */
int gigs_nonzero_mask(int height, int width, const float* in, float* mask, void* stream);
int gigs_stage2_loss_fwd(int height, int width, const float* render_direct, const float* irr_linear,
                         const float* gt_image, const float* normal_mask_f, const float* roughness,
                         const float* metallic, float* render_rgb, float* acc4, float* loss, void* stream);
int gigs_stage2_loss_bwd(int height, int width, const float* render_direct, const float* irr_linear,
                         const float* gt_image, const float* normal_mask_f, const float* acc4, const float* g_loss,
                         float* d_render_direct, float* d_irr_linear, float* d_roughness, float* d_metallic,
                         void* stream);
/* gigs_stage2_loss_fwd that also writes, in the same pass over the image (the backward kernel would convert the same
 * halo tiles and select the same medians again), the loss gradients w.r.t. render_direct and irr_linear for a unit
 * upstream gradient ([3,H,W] each; d_irr_linear_unit is zeroed inside and accumulated with atomics through the median's
 * tap selection).  gigs_shade_bwd_ex scales them (ext.g_scale) and forms the lamb terms (ext.lamb_*), so the stage-2
 * backward needs no loss kernel. */
int gigs_stage2_loss_fwd_grad(int height, int width, const float* render_direct, const float* irr_linear,
                              const float* gt_image, const float* normal_mask_f, const float* roughness,
                              const float* metallic, float* render_rgb, float* acc4, float* loss,
                              float* d_render_direct_unit, float* d_irr_linear_unit, void* stream);

/* dr.texture(cubemap[None], dirs[None], filter_mode="linear", boundary_mode="cube") (train.py:409-417, render.py:80,
 * relight.py:108) for n directions [n,3], sampled as the shade kernel samples light.diffuse (face by largest |axis|,
 * bilinear taps, taps beyond an edge come from the neighbouring face, the missing corner tap is dropped; the rule
 * documented at gigs_shade_fwd): out is [n,3], or three planes [3,n] if planar.
 * The backward accumulates into d_cubemap [6,res,res,3] (caller zeroes); directions get no gradient. */
int gigs_cube_texture_fwd(int res, const float* cubemap, int n, const float* dirs, float* out, int planar,
                          void* stream);
/* latlong_to_cubemap (relight.py:92-111): cubemap [6,res_y,res_x,C] from an equirectangular map [lat_h,lat_w,C]:
 * texel direction = normalize(cube_to_dir(face, linspace(-1+1/res, 1-1/res))), tu = atan2(x,-z)/(2pi)+0.5,
 * tv = acos(clamp(y))/pi, then the 2D lookup dr.texture(latlong, (tu,tv), filter_mode="linear") -- nvdiffrast
 * (third party, absent): bilinear, texel centres at (i+0.5)/size, boundary_mode "wrap".  PARITY UNPINNED. */
int gigs_latlong_to_cubemap(int res_y, int res_x, int lat_h, int lat_w, int channels, const float* latlong,
                            float* cubemap, void* stream);
int gigs_cube_texture_bwd(int res, int n, const float* dirs, const float* g_out, float* d_cubemap, int planar,
                          void* stream);
/* The lookup's backward as a gather, for a direction set that does not change between calls (the envmap TV's panorama grid).
 * gigs_cube_taps exports the four taps of every direction (idx [n,4] texel index or -1, w [n,4]); the caller sorts the valid
 * ones by texel into a CSR list -- offsets [6 res^2 + 1], ent_sample / ent_w in sample order within a texel -- and
 * gigs_cube_texture_bwd_gather writes EVERY texel of d_cubemap (no zero-fill needed) as the sum of its entries: one lane per
 * texel, one wave for each of the n_heavy texels (heavy_ids) that hold more than `heavy` entries.  No atomics: reproducible. */
int gigs_cube_taps(int res, int n, const float* dirs, int* idx, float* w, void* stream);
int gigs_cube_texture_bwd_gather(int res, int n, int planar, const int* offsets, const int* ent_sample, const float* ent_w,
                                 int heavy, int n_heavy, const int* heavy_ids, const float* g_out, float* d_cubemap,
                                 void* stream);

/* Training-loop glue (gigs-hip extension): the image losses of train.py and the Adam step, one
 * pass each.  Planes are [C,H,W] fp32; `scratch` holds at least gigs_loss_scratch_floats(C,H,W) floats (per-workgroup
 * partial sums, added in a fixed order -> reproducible losses); g_loss is a device scalar (NULL = 1).
 * gigs_l1_ssim_fwd = train.py:318-320 with utils/loss_utils.py:19-20, 55-98:
 *   out3 = {(1-lambda)*mean|image-gt| + lambda*(1-mean(ssim_map)), mean|image-gt|, mean(ssim_map)}; window 11,
 *   sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2.  d_mu1/d_e11/d_e12 [C,H,W] (all or none) receive the
 *   per-pixel derivatives of ssim_map that gigs_l1_ssim_bwd spreads back through the window into g_image.
 * gigs_tv_loss_fwd/bwd = get_tv_loss(gt, prediction, pad=1, step) (train.py:83-113) or, with mask_f [H,W] != NULL,
 *   get_masked_tv_loss without erosion (train.py:116-142); gt is [3,H,W] (NULL = no guide image, all pair weights 1:
 *   the plain TV of train.py:419-421), prediction [C,H,W]; gradient to prediction.
 * gigs_masked_l1_fwd/bwd = F.l1_loss(a[:, mask], b[:, mask]) (train.py:327), mask u8 [H,W];
 *   loss_count = {loss, number of mask pixels}; g_a / g_b may each be NULL (g_b = -g_a).
 * gigs_adam_step = torch.optim.Adam(eps=..., betas=...) without weight decay / amsgrad, as the reference configures it
 *   (scene/gaussian_model.py:325-346), for all parameter groups in one launch per 16 groups; `step` is the 1-based
 *   update count of the group's state, `lr` its current learning rate; zero_grad != 0 also clears the gradients.
 *   grad == NULL: the group's gradient is an exact zero that was not materialised -- the update runs with g = 0 (the
 *   moments decay, the parameter follows its momentum), which is NOT torch's "grad is None: skip the parameter". */
typedef struct gigs_adam_group {
  float* param;
  float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  long long n;
  double lr;
  int step;
} gigs_adam_group;
size_t gigs_loss_scratch_floats(int channels, int height, int width);
int gigs_l1_ssim_fwd(int channels, int height, int width, const float* image, const float* gt, float lambda_dssim,
                     float* d_mu1, float* d_e11, float* d_e12, float* scratch, float* out3, void* stream);
int gigs_l1_ssim_bwd(int channels, int height, int width, const float* image, const float* gt, float lambda_dssim,
                     const float* d_mu1, const float* d_e11, const float* d_e12, const float* g_loss, float* g_image,
                     void* stream);
int gigs_tv_loss_fwd(int channels, int height, int width, int step, const float* gt, const float* prediction,
                     const float* mask_f, float* scratch, float* loss, void* stream);
int gigs_tv_loss_bwd(int channels, int height, int width, int step, const float* gt, const float* prediction,
                     const float* mask_f, const float* g_loss, float* g_prediction, void* stream);
int gigs_masked_l1_fwd(int channels, int height, int width, const float* a, const float* b, const uint8_t* mask,
                       float* scratch, float* loss_count, void* stream);
int gigs_masked_l1_bwd(int channels, int height, int width, const float* a, const float* b, const uint8_t* mask,
                       const float* loss_count, const float* g_loss, float* g_a, float* g_b, void* stream);
int gigs_adam_step(int n_groups, const gigs_adam_group* groups, double beta1, double beta2, double eps, int zero_grad,
                   void* stream);
/* hipGraph-capturable form (gigs-hip extension): the per-step scalars of group k -- {lr / (1 - beta1^t), sqrt(1 - beta2^t)}
 * -- are read by the kernel from DEVICE memory, dyn[2k], dyn[2k+1] (k = index into `groups`; the groups' lr / step members
 * are ignored), so a captured launch stays valid while the host refreshes the table before every replay.
 * gigs_adam_scalars computes one group's pair exactly as gigs_adam_step does (double arithmetic, torch's order). */
int gigs_adam_step_dyn(int n_groups, const gigs_adam_group* groups, double beta1, double beta2, double eps, int zero_grad,
                       const float* dyn, void* stream);
void gigs_adam_scalars(double lr, int step, double beta1, double beta2, float* out2);
/* gigs_adam_step_dyn that also reports whether the update CHANGED a bit of any watched group: `watch` is a HOST array of
 * n_groups flags (non-zero = watched), `changed` a DEVICE word that is OR-ed with 1 when a parameter of a watched group
 * differs from its value before the step (it is never cleared here).  A stage-2 trainer watches the geometry groups: as
 * long as the word stays 0, per-view tile lists, occlusion planes and indirect-light hit lists remain valid
 * (gigs_ctx_set_reuse_binning). */
int gigs_adam_step_watch(int n_groups, const gigs_adam_group* groups, double beta1, double beta2, double eps, int zero_grad,
                         const float* dyn, const unsigned char* watch, unsigned* changed, void* stream);
/* gigs_adam_step_watch behind a guard: `guard` is a DEVICE word (NULL = none); while it is non-zero the launch changes nothing
 * -- parameters, moments and gradients stay as they are.  With the violation counter of gigs_ctx_set_materials_only as the
 * guard, an update is never computed from gradients that were taken for zero and were not. */
int gigs_adam_step_guarded(int n_groups, const gigs_adam_group* groups, double beta1, double beta2, double eps, int zero_grad,
                           const float* dyn, const unsigned char* watch, unsigned* changed, const unsigned* guard, void* stream);

/* The parameter getters of GaussianModel (scene/gaussian_model.py:48-58, 178-263) as one pass each way:
 * shs = cat(f_dc [P,1,3], f_rest [P,K-1,3]) -> [P,K,3]; opacities / albedo / roughness / metallic = sigmoid(raw);
 * scales = exp(scaling); rotations = F.normalize(rotation), normal = F.normalize(normal) (dim -1, eps 1e-12).
 * gigs_activate_fwd: out->shs may be NULL (the caller hands the two SH tensors to the rasterizer as they are:
 * gigs_ctx_set_split_sh) -- no concatenation then.
 * gigs_activate_bwd: grad_out members may be NULL (= zero gradient); every non-NULL grad_raw tensor is overwritten, a NULL
 * one is neither computed nor written (a gradient the caller knows to be zero and keeps no tensor for). */
typedef struct gigs_activation_raw {
  const float *f_dc, *f_rest, *opacity, *normal, *albedo, *roughness, *metallic, *scaling, *rotation;
} gigs_activation_raw;
typedef struct gigs_activation_out { /* forward: outputs; backward: their incoming gradients */
  float *shs, *opacities, *normal, *albedo, *roughness, *metallic, *scales, *rotations;
} gigs_activation_out;
typedef struct gigs_activation_raw_grad {
  float *f_dc, *f_rest, *opacity, *normal, *albedo, *roughness, *metallic, *scaling, *rotation;
} gigs_activation_raw_grad;
int gigs_activate_fwd(int P, int K, const gigs_activation_raw* raw, const gigs_activation_out* out, void* stream);
int gigs_activate_bwd(int P, int K, const gigs_activation_raw* raw, const gigs_activation_out* grad_out,
                      const gigs_activation_raw_grad* grad_raw, void* stream);

/* Densification bookkeeping (gigs-hip extension).
 * gigs_densify_stats = train.py:494-498 + GaussianModel.add_densification_stats (scene/gaussian_model.py:933-945) in one
 *   pass over the P Gaussians, for those with radii > 0: max_radii2D = max(., radii); xyz_gradient_accum += |(gx,gy)|;
 *   xyz_gradient_accum_abs += |gx|+|gy|; xyz_gradient_accum_abs_max = max(., |gx|+|gy|); denom += 1.
 *   viewspace_grad is means2D.grad [P,3]; the five statistics are fp32 [P].
 * gigs_gather_rows rebuilds any number of row-major fp32 tensors after a densify / prune decision in one launch
 *   (replaces the per-tensor boolean-mask indexing and torch.cat of scene/gaussian_model.py:595-706):
 *   dst_t[r, :] = (zero_row && zero_row[r] && t.zero_new) ? 0 : src_t[src_index[r], :] for r < n_rows_out;
 *   src_index values must lie in [0, n_rows_in): the kernel does not check them (gi-gs_amd/densify.py builds them from
 *   arange(n_rows_in) selections only). */
typedef struct gigs_gather_tensor {
  const float* src;
  float* dst;
  int row_floats; /* floats per row */
  int zero_new;   /* 1: rows flagged in zero_row become zero (optimizer moments of new Gaussians) */
} gigs_gather_tensor;
int gigs_densify_stats(int P, const float* viewspace_grad, const int* radii, float* xyz_gradient_accum,
                       float* xyz_gradient_accum_abs, float* xyz_gradient_accum_abs_max, float* denom,
                       float* max_radii2D, void* stream);
int gigs_gather_rows(int n_tensors, const gigs_gather_tensor* tensors, long long n_rows_out, long long n_rows_in,
                     const int* src_index, const uint8_t* zero_row, void* stream);

/* simple-knn's distCUDA2 (submodules/simple-knn/simple_knn.cu:165-224, ext.cpp / spatial.cu):
 * mean_dists[i] = mean of the squared distances from points[i] to its three nearest other points (by index: duplicates
 * count, at distance 0); with fewer than four points the reference's FLT_MAX placeholders stay in the sum (> 1e38 or +inf).  points [P,3] fp32.
 * scratch: gigs_dist2_scratch_bytes(P) bytes of device memory.  No host read-back; everything is queued on `stream`. */
size_t gigs_dist2_scratch_bytes(int P);
int gigs_dist2(int P, const float* points, float* mean_dists, void* scratch, size_t scratch_bytes, void* stream);

/* Test/diagnostic views into the opaque scratch buffers (byte offsets from the buffer
 * start, or -1).  `which`: geometry 0 depths f32[P], 1 pos_view f32[3P], 2 means2D f32[2P],
 * 3 cov3D f32[6P], 4 conic_opacity f32[4P], 5 rgb f32[3P], 6 clamped u8[3P],
 * 7 tiles_touched u32[P], 8 point_offsets u32[P];  binning 0 keys_unsorted u64[R],
 * 1 values_unsorted u32[R], 2 keys u64[R], 3 point_list u32[R], 4 hit_mask u8[4][R] (byte i of plane w: quadrant
 * w of instance i's tile blended it);  image 0 final_T f32[N],
 * 1 n_contrib u32[N], 2 ranges u32[2T], 3 tile_order u32[T] (the blend kernels' workgroup -> tile map, longest lists first). */
long long gigs_geom_offset(int P, int which);
long long gigs_binning_offset(int num_rendered, int which);
long long gigs_image_offset(int width, int height, int which);

/* Self-test of two exact rewrites used by the SSAO/SSR march: out_fast[2i..] = the shared-reciprocal
 * division (nx[i]/d[i], ny[i]/d[i]), out_ref = the same with the compiler's IEEE division;
 * out_round[2i] = the add-and-truncate rounding of nx[i], out_round[2i+1] = (int)roundf(nx[i]). */
int gigs_selftest_div2(int n, const float* nx, const float* ny, const float* d, float* out_fast,
                       float* out_ref, int* out_round, void* stream);
/* Exhaustive self-test of the march's pixel rounding: *mismatches (device u64) receives the number of fp32 bit
 * patterns t (all 2^32 are tried) for which floor(t + (0.5 - 2^-25)) and (int)roundf(t) would select a different
 * pixel or decide "inside the image" differently for some image side < 2^15.  Expected: 0. */
int gigs_selftest_round(unsigned long long* mismatches, void* stream);

/* Rasterizer::lite_forward (R/cuda_rasterizer/rasterizer.h:90-117; bound as _C.lite_rasterize_gaussians,
 * R/rasterize_points.cu:39-127): colour [3,H,W], opacity [1,H,W] and depth [1,H,W] only, "for baking".  Composites exactly
 * like gigs_forward; the geometry / image callbacks are asked for somewhat more than gigs_required_geom / _image
 * (zero material attributes and the planes that are not returned live there).  Returns num_rendered. */
int gigs_lite_forward(gigs_ctx* ctx, gigs_alloc_fn geometryBuffer, void* geom_user, gigs_alloc_fn binningBuffer, void* binning_user,
                      gigs_alloc_fn imageBuffer, void* image_user, int P, int D, int M, const float* background, int width,
                      int height, const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                      const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                      const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                      int prefiltered, int argmax_depth, float* out_color, float* out_opacity, float* out_depth, int* radii,
                      int debug, void* stream);

/* The backward of Gaussian_SSR as the reference's autograd function defines it (R/diff_gaussian_rasterization/__init__.py:
 * 671-693): grad_albedo [3,H,W] = grad_color * abd; roughness / metallic / F0 receive nothing.  (The reference also exports
 * a CUDA SSR_BACKWARD whose only call is commented out; this entry is the live arithmetic, not that kernel.) */
int gigs_ssr_backward(int width, int height, const float* grad_color, const float* abd, float* grad_albedo, void* stream);

/* Frozen geometry (gigs-hip extension).  A stage-2 iteration of train.py updates materials and light only
 * (train.py:330-420): positions, covariances and opacities -- hence every tile list -- are the same from one visit of a
 * view to the next.  After gigs_ctx_set_reuse_binning(ctx, 1) a gigs_forward(ctx, ...) runs the preprocess (the per-Gaussian
 * records carry the material attributes) and the blend, and SKIPS count / prefix / scatter / sort: the binning and image
 * chunks its callbacks return must still hold what an earlier forward of the SAME view and geometry left there (ranges,
 * tile order, sorted point list; asynchronous layout: needs gigs_ctx_set_async_binning with the same capacity).  Outputs
 * are then those of a complete forward bit for bit.  Whether the geometry is unchanged is the caller's knowledge
 * (gigs_adam_step_watch reports it from the optimizer step). */
int gigs_ctx_set_reuse_binning(gigs_ctx* ctx, int on);

/* Declared stage-2 gradient set (gigs-hip extension).  The loss of a stage-2 iteration reaches the material planes and
 * the light only (train.py:330-420), and the material planes' blend gradients do not feed dL/dalpha (backward.cu:580-590):
 * every gradient of gigs_backward other than dL_dalbedo / dL_droughness / dL_dmetallic (and the densification slot of
 * dL_dmean2D) is an exact zero, 59 of 67 floats per Gaussian at SH degree 3 that the reference writes, activates and feeds
 * to Adam all the same.  After gigs_ctx_set_materials_only(ctx, violations) -- `violations` a DEVICE uint32 the caller
 * zeroes -- gigs_backward(ctx, ...) writes those four outputs only; every other gradient pointer may be NULL and is never
 * written.  The premise is checked on the device: each wave that finds a live Gaussian whose blend-backward record holds a
 * non-zero (or NaN) value in any other slot adds 1 to *violations; a caller that reads a non-zero count has used zeros that
 * were not zeros and must not continue.  The consumers take the absent gradients as what they are: gigs_activate_bwd skips a
 * NULL grad_raw member, a gigs_adam_group with grad == NULL is updated with g = 0 (same arithmetic, nothing read).
 * violations == NULL restores the complete backward. */
int gigs_ctx_set_materials_only(gigs_ctx* ctx, void* violations);

/* Split SH input (gigs-hip extension).  The optimizer keeps the SH coefficients as two tensors -- _features_dc [P,1,3] and
 * _features_rest [P,M-1,3] (scene/gaussian_model.py:48-58) -- and get_features concatenates them for every render: a pure copy
 * of 12 M bytes per Gaussian each way.  After gigs_ctx_set_split_sh(ctx, sh_rest), gigs_forward(ctx, ...) takes `shs` as the
 * degree-0 tensor and reads coefficients 1..M-1 from sh_rest (M is still the total count): same colours bit for bit, no
 * concatenated copy.  The backward of such a forward is the materials-only one (gigs_ctx_set_materials_only, which does not
 * touch SH); gigs_backward refuses anything else.  sh_rest == NULL restores the single tensor. */
int gigs_ctx_set_split_sh(gigs_ctx* ctx, const float* sh_rest);

/* Optional scheduling hook: a hipEvent_t (caller-owned, NULL = none) that gigs_forward records on its stream right
 * before it launches the alpha-blend kernel, so that a caller can start independent work on another stream next to
 * that kernel (it is latency-bound and leaves most CUs idle) rather than next to the bandwidth-bound binning kernels.
 * Per context: forwards issued with `ctx` record it, others do not. */
int gigs_ctx_set_blend_begin_event(gigs_ctx* ctx, void* hip_event);

/* One-wave kernel that occupies `stream` for about `nanoseconds` (it spins on the constant-rate wall clock; capped at
 * 1 ms; capturable into a hipGraph).  Inside a graph, kernel nodes that become ready together start in an order the
 * caller cannot choose; a short head start for the branch that carries a latency-bound kernel (the blend backward's
 * long tiles) ahead of a branch that floods every CU (the light's GGX backward) is expressed as a delay node at the
 * head of the latter (pipeline.WholeStepGraph).  gigs-hip extension. */
int gigs_stream_delay(unsigned nanoseconds, void* stream);

/* Asynchronous binning (gigs-hip extension).  The reference's forward reads the instance count back in the middle
 * (rasterizer_impl.cu:589-594) to size the binning buffer, which serialises host and device and keeps the forward out
 * of a hipGraph.  After gigs_ctx_set_async_binning(ctx, r_capacity > 0, counters) every gigs_forward(ctx, ...) asks the
 * binning callback for gigs_required_binning(r_capacity) bytes, bins at most r_capacity instances, reads nothing back
 * and RETURNS r_capacity (pass it to gigs_backward as R: both carve the same layout).  `counters` (device, 2 x u32, may
 * be NULL) receives {actual instance count, the count again if it exceeded the capacity else 0}; on overflow the surplus
 * instances are dropped (memory-safe, wrong image): the caller checks the flag when convenient, grows the capacity
 * and repeats the step.  r_capacity = 0 restores the synchronous behaviour.  Needs the tile-bucketed binning path
 * (options.binning_legacy = 0) and <= 16384 tiles.  Per context: two contexts may bin with different capacities and
 * counters on two streams at the same time. */
int gigs_ctx_set_async_binning(gigs_ctx* ctx, int r_capacity, unsigned* device_counters);

/* In-library stage timing for bench.py.  Between gigs_profile_begin() and gigs_profile_end()
 * every kernel stage launched by this library records a hipEvent pair on its own stream (no
 * synchronisation is added).  gigs_profile_end() waits for the recorded events, writes the
 * summed milliseconds and launch counts per stage id into the two arrays of length n and
 * returns the number of stage ids; gigs_profile_stage_name(i) names stage i. */
void gigs_profile_begin(void);
int gigs_profile_end(float* total_ms, int* launches, int n);
const char* gigs_profile_stage_name(int stage);

#ifdef __cplusplus
}
#endif
#endif /* GIGS_HIP_H_ */
