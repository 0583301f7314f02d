// pbr_oracle.cpp -- CPU restatement of the deferred-shade side of the GI-GS hot path.
//
// *** TEST INFRASTRUCTURE ONLY *** (see gigs_oracle.cpp).  Built into libgigs_oracle.so.
//
// Two groups of functions:
//  (1) cubemap filters, restated from the reference's own CUDA
//      (pbr/renderutils/c_src/cubemap.cu, "RU/cubemap.cu" below): diffuse irradiance
//      convolution :110-169, GGX bounds :181-244, GGX specular pre-filter :246-350,
//      helpers pixel_area :17-31, cube_to_dir :33-47, safeNormalize RU/vec3f.h:90-94.
//  (2) texture lookups + split-sum shade, restated from pbr/shade.py:108-241 and
//      pbr/light.py:54-79, 142-152.  The lookups themselves are nvdiffrast `dr.texture`
//      calls in the reference; nvdiffrast is a third-party dependency that is absent from the
//      reference tree and from this container and is unpinned (cloned from GitHub HEAD,
//      README.md:34-35) -> PARITY UNPINNED for the lookups.  The definition used here follows
//      nvdiffrast's published behaviour: cube face selection / (u,v) as in its
//      `indexCubeMap`, bilinear taps at (uv * size - 0.5), taps that leave a face are taken
//      from the neighbouring face (seamless), the one missing tap at a cube corner is dropped
//      and the other weights renormalised, 2-D 'clamp' boundary clamps tap indices, and
//      'linear-mipmap-linear' with an explicit mip stack and mip_level_bias blends the two
//      nearest levels of level = clamp(bias, 0, L-1).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 safeNormalize(V3 v) {  // RU/vec3f.h:90-94
  float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
  return l > 0.0f ? V3{v.x / l, v.y / l, v.z / l} : V3{0, 0, 0};
}
inline float pixel_area(int x, int y, int N) {  // RU/cubemap.cu:17-31
  if (N > 1) {
    int H = N / 2;
    x = std::abs(x - H);
    y = std::abs(y - H);
    float dx = atanf((float)(x + 1) / (float)H) - atanf((float)x / (float)H);
    float dy = atanf((float)(y + 1) / (float)H) - atanf((float)y / (float)H);
    return dx * dy;
  }
  return 1;
}
inline V3 cube_dir_raw(float fx, float fy, int side) {
  switch (side) {
    case 0: return {1, -fy, -fx};
    case 1: return {-1, -fy, fx};
    case 2: return {fx, 1, fy};
    case 3: return {fx, -1, -fy};
    case 4: return {fx, -fy, 1};
    case 5: return {-fx, -fy, -1};
  }
  return {0, 0, 0};
}
inline V3 cube_to_dir(int x, int y, int side, int N) {  // RU/cubemap.cu:33-47
  float fx = 2.0f * (((float)x + 0.5f) / (float)N) - 1.0f;
  float fy = 2.0f * (((float)y + 0.5f) / (float)N) - 1.0f;
  return safeNormalize(cube_dir_raw(fx, fy, side));
}
inline float ndfGGX(float alphaSqr, float cosTheta) {  // RU/cubemap.cu:174-179 (M_PI is double)
  float c = fminf(fmaxf(cosTheta, 0.0f), 1.0f);
  float d = (c * alphaSqr - c) * c + 1.0f;
  return (float)((double)alphaSqr / ((double)(d * d) * M_PI));
}

// ---- texture sampling ----------------------------------------------------------------
// face / (u, v) in [0,1] of a direction; -1 for non-finite input
inline int cube_face_uv(float x, float y, float z, float& u, float& v) {
  float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
  int idx;
  float c;
  if (az > fmaxf(ax, ay)) { idx = 4; c = z; }
  else if (ay > ax) { idx = 2; c = y; y = z; }
  else { idx = 0; c = x; x = z; }
  if (c < 0.f) idx += 1;
  float m = (1.0f / fabsf(c)) * 0.5f;
  float m0 = (idx == 0 || idx == 5) ? -m : m;
  float m1 = (idx != 2) ? -m : m;
  u = x * m0 + 0.5f;
  v = y * m1 + 0.5f;
  if (!std::isfinite(u) || !std::isfinite(v)) return -1;
  u = fminf(fmaxf(u, 0.f), 1.f);
  v = fminf(fmaxf(v, 0.f), 1.f);
  return idx;
}
struct Taps { int idx[4]; float w[4]; };
// bilinear taps of a cube level of resolution `res`; idx = (face*res + y)*res + x or -1
inline bool cube_taps(int res, float dx, float dy, float dz, Taps& t) {
  float u, v;
  const int face = cube_face_uv(dx, dy, dz, u, v);
  if (face < 0) return false;
  const float fu = u * (float)res - 0.5f, fv = v * (float)res - 0.5f;
  const float flu = floorf(fu), flv = floorf(fv);
  const int iu0 = (int)flu, iv0 = (int)flv;
  const float tu = fu - flu, tv = fv - flv;
  float wsum = 0.0f;
  bool dropped = false;
  for (int k = 0; k < 4; k++) {
    const int ox = k & 1, oy = k >> 1;
    const int ix = iu0 + ox, iy = iv0 + oy;
    const float w = (ox ? tu : 1.0f - tu) * (oy ? tv : 1.0f - tv);
    const bool out_x = ix < 0 || ix >= res, out_y = iy < 0 || iy >= res;
    int idx;
    if (!out_x && !out_y) idx = (face * res + iy) * res + ix;
    else if (out_x && out_y) { idx = -1; dropped = true; }
    else {
      // texel centre beyond the face edge, seen from the cube centre, lands in the
      // neighbouring face's border texel
      const float a = 2.0f * (((float)ix + 0.5f) / (float)res) - 1.0f;
      const float b = 2.0f * (((float)iy + 0.5f) / (float)res) - 1.0f;
      const V3 d = cube_dir_raw(a, b, face);
      float u2, v2;
      const int f2 = cube_face_uv(d.x, d.y, d.z, u2, v2);
      const int x2 = std::min(res - 1, std::max(0, (int)floorf(u2 * (float)res)));
      const int y2 = std::min(res - 1, std::max(0, (int)floorf(v2 * (float)res)));
      idx = (f2 * res + y2) * res + x2;
    }
    t.idx[k] = idx;
    t.w[k] = w;
    if (idx >= 0) wsum += w;
  }
  if (dropped) {
    for (int k = 0; k < 4; k++) t.w[k] = t.idx[k] >= 0 ? t.w[k] / wsum : 0.0f;
  }
  return true;
}
inline V3 cube_sample(const float* tex, int res, const Taps& t) {
  V3 r = {0, 0, 0};
  for (int k = 0; k < 4; k++)
    if (t.idx[k] >= 0) {
      const float* p = tex + 3 * (size_t)t.idx[k];
      r.x += p[0] * t.w[k];
      r.y += p[1] * t.w[k];
      r.z += p[2] * t.w[k];
    }
  return r;
}

// pbr/light.py:142-152
inline float get_mip(float r, int L, float& dmip_dr) {
  const float MINR = 0.08f, MAXR = 0.5f;
  if (r < MAXR) {
    const float c = fminf(fmaxf(r, MINR), MAXR);
    dmip_dr = (r >= MINR && r <= MAXR) ? (1.0f / (MAXR - MINR)) * (float)(L - 2) : 0.0f;
    return (c - MINR) / (MAXR - MINR) * (float)(L - 2);
  }
  const float c = fminf(fmaxf(r, MAXR), 1.0f);
  dmip_dr = (r >= MAXR && r <= 1.0f) ? 1.0f / (1.0f - MAXR) : 0.0f;
  return (c - MAXR) / (1.0f - MAXR) + (float)L - 2.0f;
}

inline float lin2srgb(float x, float& d) {  // pbr/shade.py:50-63
  const float eps = 1.1920929e-07f;
  if (x <= 0.0031308f) { d = 323.0f / 25.0f; return 323.0f / 25.0f * x; }
  const float c = fmaxf(x, eps);
  const float p = powf(c, 5.0f / 12.0f);
  d = x >= eps ? 211.0f * (5.0f / 12.0f) * p / c / 200.0f : 0.0f;
  return (211.0f * p - 11.0f) / 200.0f;
}
inline float aces(float x, float& d) {  // pbr/shade.py:33-47 (before the clamp)
  const float a = 2.51f, b = 0.03f, c = 2.43f, dd = 0.59f, e = 0.14f;
  const float num = x * (a * x + b), den = x * (c * x + dd) + e;
  d = ((2 * a * x + b) * den - num * (2 * c * x + dd)) / (den * den);
  return num / den;
}

struct ShadeIn {
  int H, W;
  const float *normals, *view_dirs, *albedo, *roughness;
  const uint8_t* mask;
  const float *occlusion, *metallic, *background;
  const float* diffuse; int diffuse_res;
  int L; const float* spec[8]; int spec_res[8];
  const float* lut; int lut_w, lut_h;
  int tone, gamma;
};

}  // namespace

extern "C" {

// ---- cubemap filters ------------------------------------------------------------------
void orc_diffuse_cubemap_fwd(int N, const float* cubemap, float* out) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int o = 0; o < 6 * N * N; o++) {
    const int pz = o / (N * N), py = (o / N) % N, px = o % N;
    const V3 Nn = cube_to_dir(px, py, pz, N);
    V3 col = {0, 0, 0};
    for (int s = 0; s < 6; ++s)
      for (int y = 0; y < N; ++y)
        for (int x = 0; x < N; ++x) {
          const V3 L = cube_to_dir(x, y, s, N);
          const float costheta = fminf(fmaxf(dot(Nn, L), 0.0f), 0.999f);
          const float w = costheta * pixel_area(x, y, N) / 3.141592f;
          const float* t = cubemap + 3 * (size_t)((s * N + y) * N + x);
          col.x += t[0] * w; col.y += t[1] * w; col.z += t[2] * w;
        }
    out[3 * (size_t)o] = col.x; out[3 * (size_t)o + 1] = col.y; out[3 * (size_t)o + 2] = col.z;
  }
}

void orc_diffuse_cubemap_bwd(int N, const float* grad_out, float* grad_cubemap) {
  std::vector<double> acc(18 * (size_t)N * N, 0.0);
  for (int o = 0; o < 6 * N * N; o++) {
    const int pz = o / (N * N), py = (o / N) % N, px = o % N;
    const V3 Nn = cube_to_dir(px, py, pz, N);
    const float* g = grad_out + 3 * (size_t)o;
    for (int s = 0; s < 6; ++s)
      for (int y = 0; y < N; ++y)
        for (int x = 0; x < N; ++x) {
          const V3 L = cube_to_dir(x, y, s, N);
          const float costheta = fminf(fmaxf(dot(Nn, L), 0.0f), 0.999f);
          const float w = costheta * pixel_area(x, y, N) / 3.141592f;
          const size_t i = 3 * (size_t)((s * N + y) * N + x);
          acc[i] += g[0] * w; acc[i + 1] += g[1] * w; acc[i + 2] += g[2] * w;
        }
  }
  for (size_t i = 0; i < acc.size(); i++) grad_cubemap[i] = (float)acc[i];
}

// SpecularBoundsKernel RU/cubemap.cu:181-244 -> bounds [6,N,N,24] (ints stored as fp32)
void orc_specular_bounds(int N, float costheta_cutoff, float* bounds) {
  const int TILE = 16;
#pragma omp parallel for schedule(dynamic, 16)
  for (int o = 0; o < 6 * N * N; o++) {
    const int pz = o / (N * N), py = (o / N) % N, px = o % N;
    const V3 VNR = cube_to_dir(px, py, pz, N);
    for (int s = 0; s < 6; ++s) {
      int minx = N - 1, maxx = 0, miny = N - 1, maxy = 0;
      for (int tx = 0; tx < (N + TILE - 1) / TILE; tx++)
        for (int ty = 0; ty < (N + TILE - 1) / TILE; ty++) {
          const int tsx = tx * TILE, tsy = ty * TILE;
          const int tex = std::min((tx + 1) * TILE, N), tey = std::min((ty + 1) * TILE, N);
          const V3 L0 = cube_to_dir(tsx, tsy, s, N), L1 = cube_to_dir(tex, tsy, s, N);
          const V3 L2 = cube_to_dir(tsx, tey, s, N), L3 = cube_to_dir(tex, tey, s, N);
          const float mnx = fminf(fminf(L0.x, L1.x), fminf(L2.x, L3.x)), mxx = fmaxf(fmaxf(L0.x, L1.x), fmaxf(L2.x, L3.x));
          const float mny = fminf(fminf(L0.y, L1.y), fminf(L2.y, L3.y)), mxy = fmaxf(fmaxf(L0.y, L1.y), fmaxf(L2.y, L3.y));
          const float mnz = fminf(fminf(L0.z, L1.z), fminf(L2.z, L3.z)), mxz = fmaxf(fmaxf(L0.z, L1.z), fmaxf(L2.z, L3.z));
          const float maxdp = fmaxf(mnx * VNR.x, mxx * VNR.x) + fmaxf(mny * VNR.y, mxy * VNR.y) + fmaxf(mnz * VNR.z, mxz * VNR.z);
          if (maxdp >= costheta_cutoff) {
            for (int y = tsy; y < tey; ++y)
              for (int x = tsx; x < tex; ++x) {
                const V3 L = cube_to_dir(x, y, s, N);
                if (dot(L, VNR) >= costheta_cutoff) {
                  minx = std::min(minx, x); maxx = std::max(maxx, x);
                  miny = std::min(miny, y); maxy = std::max(maxy, y);
                }
              }
          }
        }
      float* b = bounds + 24 * (size_t)o + s * 4;
      b[0] = (float)minx; b[1] = (float)maxx; b[2] = (float)miny; b[3] = (float)maxy;
    }
  }
}

// SpecularCubemapFwdKernel RU/cubemap.cu:246-298 -> out [6,N,N,4] (rgb, wsum)
void orc_specular_cubemap_fwd(int N, const float* cubemap, const float* bounds, float roughness,
                              float costheta_cutoff, float* out) {
  const float alpha = roughness * roughness, alphaSqr = alpha * alpha;
#pragma omp parallel for schedule(dynamic, 16)
  for (int o = 0; o < 6 * N * N; o++) {
    const int pz = o / (N * N), py = (o / N) % N, px = o % N;
    const V3 VNR = cube_to_dir(px, py, pz, N);
    float wsum = 0.0f;
    V3 col = {0, 0, 0};
    for (int s = 0; s < 6; ++s) {
      const float* b = bounds + 24 * (size_t)o + s * 4;
      const int xmin = (int)b[0], xmax = (int)b[1], ymin = (int)b[2], ymax = (int)b[3];
      if (xmin <= xmax)
        for (int y = ymin; y <= ymax; ++y)
          for (int x = xmin; x <= xmax; ++x) {
            const V3 L = cube_to_dir(x, y, s, N);
            if (dot(L, VNR) >= costheta_cutoff) {
              const V3 Hh = safeNormalize(L + VNR);
              const float wiDotN = fmaxf(dot(L, VNR), 0.0f);
              const float VNRDotH = fmaxf(dot(VNR, Hh), 0.0f);
              const float w = wiDotN * ndfGGX(alphaSqr, VNRDotH) * pixel_area(x, y, N) / 4.0f;
              const float* t = cubemap + 3 * (size_t)((s * N + y) * N + x);
              col.x += t[0] * w; col.y += t[1] * w; col.z += t[2] * w;
              wsum += w;
            }
          }
    }
    float* q = out + 4 * (size_t)o;
    q[0] = col.x; q[1] = col.y; q[2] = col.z; q[3] = wsum;
  }
}

// SpecularCubemapBwdKernel RU/cubemap.cu:300-350; grad_out is [6,N,N,4], channel 3 unused
void orc_specular_cubemap_bwd(int N, const float* bounds, const float* grad_out, float roughness,
                              float costheta_cutoff, float* grad_cubemap) {
  const float alpha = roughness * roughness, alphaSqr = alpha * alpha;
  std::vector<double> acc(18 * (size_t)N * N, 0.0);
  for (int o = 0; o < 6 * N * N; o++) {
    const int pz = o / (N * N), py = (o / N) % N, px = o % N;
    const V3 VNR = cube_to_dir(px, py, pz, N);
    const float* g = grad_out + 4 * (size_t)o;
    for (int s = 0; s < 6; ++s) {
      const float* b = bounds + 24 * (size_t)o + s * 4;
      const int xmin = (int)b[0], xmax = (int)b[1], ymin = (int)b[2], ymax = (int)b[3];
      if (xmin <= xmax)
        for (int y = ymin; y <= ymax; ++y)
          for (int x = xmin; x <= xmax; ++x) {
            const V3 L = cube_to_dir(x, y, s, N);
            if (dot(L, VNR) >= costheta_cutoff) {
              const V3 Hh = safeNormalize(L + VNR);
              const float wiDotN = fmaxf(dot(L, VNR), 0.0f);
              const float VNRDotH = fmaxf(dot(VNR, Hh), 0.0f);
              const float w = wiDotN * ndfGGX(alphaSqr, VNRDotH) * pixel_area(x, y, N) / 4.0f;
              const size_t i = 3 * (size_t)((s * N + y) * N + x);
              acc[i] += g[0] * w; acc[i + 1] += g[1] * w; acc[i + 2] += g[2] * w;
            }
          }
    }
  }
  for (size_t i = 0; i < acc.size(); i++) grad_cubemap[i] = (float)acc[i];
}

// cubemap_mip forward: 2x2 average pool per face (pbr/light.py:56-60). in [6,2r,2r,C] -> out [6,r,r,C]
void orc_cubemap_mip_fwd(int r, int C, const float* in, float* out) {
  const int R2 = 2 * r;
  for (int f = 0; f < 6; f++)
    for (int y = 0; y < r; y++)
      for (int x = 0; x < r; x++)
        for (int c = 0; c < C; c++) {
          auto at = [&](int yy, int xx) { return in[((size_t)(f * R2 + yy) * R2 + xx) * C + c]; };
          out[((size_t)(f * r + y) * r + x) * C + c] =
              (at(2 * y, 2 * x) + at(2 * y, 2 * x + 1) + at(2 * y + 1, 2 * x) + at(2 * y + 1, 2 * x + 1)) * 0.25f;
        }
}

// cubemap_mip backward (pbr/light.py:62-79): every fine texel bilinearly looks up 0.25*dout
// along its own direction (dr.texture, 'linear', 'cube').  dout [6,r,r,3] -> din [6,2r,2r,3].
void orc_cubemap_mip_bwd(int r, const float* dout, float* din) {
  const int res = 2 * r;
  std::vector<float> q(18 * (size_t)r * r);
  for (size_t i = 0; i < q.size(); i++) q[i] = dout[i] * 0.25f;
  for (int s = 0; s < 6; s++)
    for (int y = 0; y < res; y++)
      for (int x = 0; x < res; x++) {
        // torch.linspace(-1 + 1/res, 1 - 1/res, res)
        const float start = -1.0f + 1.0f / (float)res, end = 1.0f - 1.0f / (float)res;
        const float stepv = (end - start) / (float)(res - 1);
        auto lin = [&](int i) { return i < res / 2 ? start + stepv * (float)i : end - stepv * (float)(res - 1 - i); };
        const float gx = lin(x), gy = lin(y);
        V3 d = cube_dir_raw(gx, gy, s);
        const float n = fmaxf(sqrtf(d.x * d.x + d.y * d.y + d.z * d.z), 1e-12f);  // F.normalize
        d = {d.x / n, d.y / n, d.z / n};
        Taps t;
        V3 v = {0, 0, 0};
        if (cube_taps(r, d.x, d.y, d.z, t)) v = cube_sample(q.data(), r, t);
        float* o = din + 3 * (size_t)((s * res + y) * res + x);
        o[0] = v.x; o[1] = v.y; o[2] = v.z;
      }
}

// ---- deferred shade -------------------------------------------------------------------
// pbr_shading forward (pbr/shade.py:108-241).  HWC inputs; spec mips as separate arrays.
void orc_shade_fwd(int H, int W, const float* normals, const float* view_dirs, const float* albedo,
                   const float* roughness, const uint8_t* mask, const float* occlusion,
                   const float* metallic, const float* background, const float* diffuse, int diffuse_res,
                   int L, const float* const* spec, const int* spec_res, const float* lut, int lut_w,
                   int lut_h, int tone, int gamma, float* render_rgb, float* diffuse_rgb,
                   float* specular_rgb, float* diffuse_light) {
#pragma omp parallel for schedule(static)
  for (int p = 0; p < H * W; p++) {
    const V3 n = {normals[3 * p], normals[3 * p + 1], normals[3 * p + 2]};
    const V3 v = {view_dirs[3 * p], view_dirs[3 * p + 1], view_dirs[3 * p + 2]};
    const V3 a = {albedo[3 * p], albedo[3 * p + 1], albedo[3 * p + 2]};
    const float r = roughness[p];
    const float ndv = n.x * v.x + n.y * v.y + n.z * v.z;
    const float c2 = 2.0f * fmaxf(ndv, 0.0f);
    const V3 ref = {c2 * n.x - v.x, c2 * n.y - v.y, c2 * n.z - v.z};
    // dir @ transform.T with transform = [[0,-1,0],[0,0,1],[-1,0,0]]  ->  (-y, z, -x)
    const V3 nt = {-n.y, n.z, -n.x}, vt = {-v.y, v.z, -v.x}, rt = {-ref.y, ref.z, -ref.x};
    Taps t;
    V3 dl = {0, 0, 0};
    if (cube_taps(diffuse_res, nt.x, nt.y, nt.z, t)) dl = cube_sample(diffuse, diffuse_res, t);
    if (occlusion) dl = dl * occlusion[p];
    V3 drgb = {dl.x * a.x, dl.y * a.y, dl.z * a.z};
    // NoV = clamp(sum(nt * vt), 1e-4, 1)
    const float nov = fminf(fmaxf(nt.x * vt.x + nt.y * vt.y + nt.z * vt.z, 1e-4f), 1.0f);
    // fg = texture2D(lut, (NoV, roughness)), linear, clamp
    float fgx, fgy;
    {
      const float fu = nov * (float)lut_w - 0.5f, fv = r * (float)lut_h - 0.5f;
      const float flu = floorf(fu), flv = floorf(fv);
      const float tu = fu - flu, tv = fv - flv;
      auto cl = [](int i, int n) { return std::min(n - 1, std::max(0, i)); };
      const int x0 = cl((int)flu, lut_w), x1 = cl((int)flu + 1, lut_w), y0 = cl((int)flv, lut_h), y1 = cl((int)flv + 1, lut_h);
      auto T = [&](int y, int x, int c) { return lut[((size_t)y * lut_w + x) * 2 + c]; };
      const float w00 = (1 - tu) * (1 - tv), w10 = tu * (1 - tv), w01 = (1 - tu) * tv, w11 = tu * tv;
      fgx = T(y0, x0, 0) * w00 + T(y0, x1, 0) * w10 + T(y1, x0, 0) * w01 + T(y1, x1, 0) * w11;
      fgy = T(y0, x0, 1) * w00 + T(y0, x1, 1) * w10 + T(y1, x0, 1) * w01 + T(y1, x1, 1) * w11;
    }
    float dmdr;
    float lvl = get_mip(r, L, dmdr);
    lvl = fminf(fmaxf(lvl, 0.0f), (float)(L - 1));
    const int l0 = std::min((int)floorf(lvl), L - 1), l1 = std::min(l0 + 1, L - 1);
    const float lf = lvl - (float)l0;
    V3 s0 = {0, 0, 0}, s1 = {0, 0, 0};
    if (cube_taps(spec_res[l0], rt.x, rt.y, rt.z, t)) s0 = cube_sample(spec[l0], spec_res[l0], t);
    if (l1 != l0 && cube_taps(spec_res[l1], rt.x, rt.y, rt.z, t)) s1 = cube_sample(spec[l1], spec_res[l1], t);
    const V3 sp = l1 != l0 ? V3{s0.x * (1 - lf) + s1.x * lf, s0.y * (1 - lf) + s1.y * lf, s0.z * (1 - lf) + s1.z * lf} : s0;
    V3 F0;
    if (metallic) {
      const float m = metallic[p];
      F0 = {(1.0f - m) * 0.04f + a.x * m, (1.0f - m) * 0.04f + a.y * m, (1.0f - m) * 0.04f + a.z * m};
    } else F0 = {0.04f, 0.04f, 0.04f};
    const V3 refl = {F0.x * fgx + fgy, F0.y * fgx + fgy, F0.z * fgx + fgy};
    V3 srgb = {sp.x * refl.x, sp.y * refl.y, sp.z * refl.z};
    float rr[3] = {drgb.x + srgb.x, drgb.y + srgb.y, drgb.z + srgb.z};
    float dd;
    for (int c = 0; c < 3; c++) {
      float x = rr[c];
      if (tone) x = aces(x, dd);
      x = fminf(fmaxf(x, 0.0f), 1.0f);
      if (gamma) x = lin2srgb(x, dd);
      rr[c] = x;
    }
    if (gamma) {
      drgb = {lin2srgb(drgb.x, dd), lin2srgb(drgb.y, dd), lin2srgb(drgb.z, dd)};
      srgb = {lin2srgb(srgb.x, dd), lin2srgb(srgb.y, dd), lin2srgb(srgb.z, dd)};
    }
    const bool mk = mask[p] != 0;
    for (int c = 0; c < 3; c++) render_rgb[3 * p + c] = mk ? rr[c] : (background ? background[3 * p + c] : 0.0f);
    diffuse_rgb[3 * p] = drgb.x; diffuse_rgb[3 * p + 1] = drgb.y; diffuse_rgb[3 * p + 2] = drgb.z;
    specular_rgb[3 * p] = srgb.x; specular_rgb[3 * p + 1] = srgb.y; specular_rgb[3 * p + 2] = srgb.z;
    diffuse_light[3 * p] = dl.x; diffuse_light[3 * p + 1] = dl.y; diffuse_light[3 * p + 2] = dl.z;
  }
}

}  // extern "C"
