"""CPU composition of the reference's per-view operator sequences from the C oracle's pieces.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg): numpy glue around
oracle/gigs_oracle.cpp and oracle/pbr_oracle.cpp that follows, line by line, what the reference's Python does
between its CUDA calls:

    operator_forward    R/diff_gaussian_rasterization/__init__.py:448-537 (GaussianRasterizer.forward)
    gbuffer_post        gaussian_renderer/__init__.py:157-199 (incl. pad_normal :159-173)
    build_mips          pbr/light.py:154-170 with ndf_cutoff = pbr/renderutils/ops.py:428-443
    stage2_forward      train.py:293-385 (stage-2 image: direct shade + indirect diffuse)
    latlong_to_cubemap  relight.py:92-111
    relight_view        relight.py:153-251 (inference image under a new environment light)
    psnr                utils/image_utils.py:31-33

Third-party arithmetic restated from its documented behaviour (absent here, unpinned in the reference's
environment.yml -> PARITY UNPINNED, as for the shade lookups): nvdiffrast `dr.texture(latlong, uv,
filter_mode="linear")` = bilinear, texel centres at (i + 0.5) / size, boundary_mode "wrap" (its default);
kornia median_blur = the oracle's median3x3.
"""
from __future__ import annotations

import math
import os
import time
from typing import Dict, Optional

import numpy as np

KEYS = ["means3D", "opacities", "normal", "albedo", "roughness", "metallic", "shs", "scales", "rotations"]
_f32 = np.float32


# --------------------------------------------------------------------------------------------------
def linear_to_srgb(x):  # train.py:54-68
    x = np.asarray(x, _f32)
    eps = np.finfo(np.float32).eps
    s0 = _f32(323.0 / 25.0) * x
    s1 = (_f32(211.0) * np.maximum(x, eps) ** _f32(5.0 / 12.0) - _f32(11.0)) / _f32(200.0)
    return np.where(x <= _f32(0.0031308), s0, s1).astype(_f32)


def srgb_to_linear(x):  # train.py:70-81
    x = np.asarray(x, _f32)
    l0 = _f32(25.0 / 323.0) * x
    with np.errstate(invalid="ignore"):
        l1 = ((x + _f32(0.055)) / _f32(1.055)) ** _f32(2.4)
    return np.where(x <= _f32(0.04045), l0, l1).astype(_f32)


def psnr(img1, img2):
    """utils/image_utils.py:31-33 as the callers use it (`psnr(a, b).mean()`, train.py:786, render.py:379): one
    PSNR per leading index (channel of a [3,H,W] image), then their mean.  Pinned by tests/golden/ref_psnr.npz."""
    a, b = np.asarray(img1, np.float64), np.asarray(img2, np.float64)
    mse = ((a - b) ** 2).reshape(a.shape[0], -1).mean(1)
    return float(np.mean(20.0 * np.log10(1.0 / np.sqrt(np.maximum(mse, 1e-30)))))


def focal(cam):
    return cam["image_width"] / (2.0 * cam["tanfovx"]), cam["image_height"] / (2.0 * cam["tanfovy"])


# --------------------------------------------------------------------------------------------------
def _tick(timings, name, t0):
    if timings is not None:
        timings[name] = timings.get(name, 0.0) + (time.perf_counter() - t0)
    return time.perf_counter()


def operator_forward(orc, sc, cam, gi, sh_degree, bg=(0.0, 0.0, 0.0), inference=False, keep_state=False, timings=None,
                     derive_normal=True, scale_modifier=1.0):
    """GaussianRasterizer.forward: rasterizer, median(depth), depth->normal, bilateral, median(pos), SSAO on the RAW
    view-space normal (R/.../__init__.py:475-517).  derive_normal=False: zeros in place of the derived normal and
    positions, the bilateral / median still applied to them (:486-504).  Returns the 12-tuple's planes by name (+ the
    Rasterizer)."""
    H, W = cam["image_height"], cam["image_width"]
    fx, fy = focal(cam)
    t0 = time.perf_counter()
    r = orc.Rasterizer()
    out = r.forward(bg=np.asarray(bg, _f32), **{k: sc[k] for k in KEYS}, sh_degree=sh_degree,
                    viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], campos=cam["campos"],
                    tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], image_height=H, image_width=W,
                    inference=inference, scale_modifier=scale_modifier)
    t0 = _tick(timings, "rasterizer_fwd", t0)
    if derive_normal:
        depth_f = orc.median3x3(out["depth"])
        nd, pos = orc.depth_to_normal(W, H, fx, fy, cam["viewmatrix"], depth_f)
    else:  # torch.zeros_like(out_normal) twice (:488-489)
        nd, pos = np.zeros((3, H, W), _f32), np.zeros((3, H, W), _f32)
    nd = orc.bilateral3x3(nd)
    posf = orc.median3x3(pos)
    occ = orc.ssao(W, H, fx, fy, gi["radius"], gi["bias"], gi["thick"], gi["delta"], gi["step"], gi["start"],
                   out["normal_view"], posf)
    _tick(timings, "filters_ssao", t0)
    res = dict(render=out["color"], radii=out["radii"], opacity_map=out["opacity"], depth_map=out["depth"],
               normal_map_from_depth=nd, normal_map=out["normal"], occlusion_map=occ, albedo_map=out["albedo"],
               roughness_map=out["roughness"], metallic_map=out["metallic"], out_normal_view=out["normal_view"],
               depth_pos=posf, pos_raw=out["pos"], num_rendered=out["num_rendered"])
    if keep_state:
        res["rasterizer"] = r
    return res


def _normalize_where(v):
    """torch.where(norm > 0, F.normalize(v, dim=0), v) with F.normalize's eps = 1e-12."""
    n = np.sqrt((v.astype(_f32) ** 2).sum(0, keepdims=True, dtype=_f32)).astype(_f32)
    with np.errstate(invalid="ignore", divide="ignore"):
        u = (v / np.maximum(n, _f32(1e-12))).astype(_f32)
    return np.where(n > 0, u, v).astype(_f32)


def gbuffer_post(orc, r: Dict, viewmatrix, pad_normal=False) -> Dict:
    """gaussian_renderer.render after the operator call (:157-199)."""
    nfd, nm, onv = r["normal_map_from_depth"], r["normal_map"], r["out_normal_view"]
    opacity = r["opacity_map"]
    normal_from_depth_mask = (nfd != 0).all(0)
    normal_mask = (nm != 0).all(0, keepdims=True)
    if pad_normal:  # :159-173
        opacity = np.where(opacity < _f32(0.004), _f32(0), opacity)
        opacity = np.where(opacity > _f32(1.0 - 0.004), _f32(1), opacity).astype(_f32)
        bgn = np.array([0.0, 0.0, 1.0], _f32)[:, None, None]
        nm = (nm * opacity + (_f32(1.0) - opacity) * bgn).astype(_f32)
        m = (nfd == 0.0).all(0, keepdims=True).astype(_f32)
        nfd = (nfd * (_f32(1.0) - m) + m * bgn).astype(_f32)
    nfd = _normalize_where(nfd)
    nm = _normalize_where(nm)
    nm = orc.median3x3(nm)
    R = np.asarray(viewmatrix, _f32)[:3, :3]
    normals_view = -(np.einsum("chw,cd->dhw", nm, R).astype(_f32))
    onv = orc.median3x3(_normalize_where(onv))
    out = dict(r)
    out.update(opacity_map=opacity, normal_map_from_depth=nfd, normal_from_depth_mask=normal_from_depth_mask,
               normal_map=normals_view, normal_mask=normal_mask, out_normal_view=onv)
    return out


def canonical_view_dirs(cam):
    """scene/__init__.py:137-169 + train.py:299-308: -(normalize(ray) . c2w rows)."""
    H, W = cam["image_height"], cam["image_width"]
    fx, fy = focal(cam)
    x, y = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    rays = np.stack([(x - W / 2 + 0.5) / fx, (y - H / 2 + 0.5) / fy, np.ones_like(x, dtype=np.float64)], -1).astype(_f32)
    rays = rays / np.maximum(np.linalg.norm(rays, axis=-1, keepdims=True), 1e-12)
    c2w = np.linalg.inv(np.asarray(cam["viewmatrix"], np.float64).T)
    return (-(rays[..., None, :].astype(np.float64) * c2w[None, None, :3, :3]).sum(-1)).astype(_f32)


# --------------------------------------------------------------------------------------------------
def ndf_cutoff(roughness: float, cutoff: float = 0.99) -> float:
    """renderutils/ops.py:428-443 (__ndfBounds): cos(theta) below which the GGX lobe holds < 1 - cutoff of its energy,
    from a 1M-sample numerical CDF."""
    def ndf_ggx(alpha_sqr, costheta):
        costheta = np.clip(costheta, 0.0, 1.0)
        d = (costheta * alpha_sqr - costheta) * costheta + 1.0
        return alpha_sqr / (d * d * np.pi)

    nsamples = 1_000_000
    alpha_sqr = roughness ** 4  # (roughness^2)^2
    costheta = np.cos(np.linspace(0, np.pi / 2.0, nsamples))
    d = np.cumsum(ndf_ggx(alpha_sqr, costheta))
    idx = np.argmax(d >= d[..., -1] * cutoff)
    return float(costheta[idx])


def build_mips(orc, base, cutoff=0.99, min_res=16, rmin=0.08, rmax=0.5):
    """CubemapLight.build_mips (pbr/light.py:154-170): (diffuse, [specular levels]) from the base cubemap."""
    spec = [np.ascontiguousarray(base, _f32)]
    while spec[-1].shape[1] > min_res:
        spec.append(orc.cubemap_mip_fwd(spec[-1]))
    diffuse = orc.diffuse_cubemap_fwd(spec[-1])
    L = len(spec)
    out = []
    for idx in range(L):
        rough = (idx / (L - 2)) * (rmax - rmin) + rmin if idx < L - 1 else 1.0
        cc = ndf_cutoff(rough, cutoff)
        b = orc.specular_bounds(spec[idx].shape[1], cc)
        rgbw = orc.specular_cubemap_fwd(spec[idx], b, rough, cc)
        out.append((rgbw[..., :3] / rgbw[..., 3:]).astype(_f32))
    return diffuse, out


def brdf_lut():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gi-gs_amd", "pbr", "brdf_256_256.bin")
    return np.fromfile(path, dtype=np.float32).reshape(256, 256, 2)  # the reference's data file (pbr/brdf_256_256.bin)


def _hwc(x):
    return np.ascontiguousarray(np.transpose(x, (1, 2, 0)))


def _chw(x):
    return np.ascontiguousarray(np.transpose(x, (2, 0, 1)))


def shade_direct(orc, post, view_dirs, albedo_chw, roughness_chw, metallic_chw, occlusion_chw, diffuse, spec, lut,
                 tone=False, gamma=False):
    """pbr_shading(...)['render_rgb'] then torch.where(normal_mask, ., background = 0) (train.py:340-368)."""
    mask = _hwc(post["normal_mask"])
    res = orc.shade_fwd(_hwc(post["normal_map"]), view_dirs, _hwc(albedo_chw), _hwc(roughness_chw), mask,
                        None if occlusion_chw is None else _hwc(occlusion_chw),
                        None if metallic_chw is None else _hwc(metallic_chw), None, diffuse, spec, lut, tone=tone, gamma=gamma)
    direct = _chw(res["render_rgb"])
    return np.where(post["normal_mask"], direct, _f32(0)).astype(_f32)


def stage2_forward(orc, sc, cam, gi, sh_degree, light_base, metallic=True, indirect=True, tone=False, gamma=False,
                   keep_state=False, timings=None):
    """train.py:266-385 for one view: returns render_rgb (the stage-2 image the L1 loss sees) and the intermediates."""
    H, W = cam["image_height"], cam["image_width"]
    fx, fy = focal(cam)
    raw = operator_forward(orc, sc, cam, gi, sh_degree, keep_state=keep_state, timings=timings)
    t0 = time.perf_counter()
    post = gbuffer_post(orc, raw, cam["viewmatrix"])
    rough = (post["roughness_map"] * _f32(1.0 - 0.04) + _f32(0.04)).astype(_f32)  # :297-298
    occ = post["occlusion_map"] if indirect else np.ones_like(rough)
    t0 = _tick(timings, "gbuffer_post", t0)
    diffuse, spec = build_mips(orc, light_base)
    t0 = _tick(timings, "light_prefilter_fwd", t0)
    lut = brdf_lut()
    vd = canonical_view_dirs(cam)
    direct = shade_direct(orc, post, vd, post["albedo_map"], rough, post["metallic_map"] if metallic else None, occ, diffuse,
                          spec, lut, tone=tone, gamma=gamma)
    if metallic:
        metal = post["metallic_map"]
        F0 = ((_f32(1.0) - metal) * _f32(0.04) + post["albedo_map"] * metal).astype(_f32)
    else:
        F0 = np.full_like(post["albedo_map"], 0.04)
        metal = np.zeros_like(rough)
    lin = srgb_to_linear(direct)
    t0 = _tick(timings, "shade_fwd", t0)
    a = (gi["radius"], gi["bias"], gi["thick"], gi["delta"], gi["step"], gi["start"])
    irr, abd = orc.ssr(W, H, fx, fy, *a, post["out_normal_view"], post["depth_pos"], lin, post["albedo_map"], rough, metal, F0)
    t0 = _tick(timings, "ssr", t0)
    irr_s = orc.median3x3(linear_to_srgb(irr))
    render_rgb = (direct + irr_s).astype(_f32)
    _tick(timings, "srgb_median_sum", t0)
    return dict(render_rgb=render_rgb, render_direct=direct, IRR=irr, IRR_srgb=irr_s, abd=abd, raw=raw, post=post,
                roughness=rough, F0=F0, diffuse=diffuse, specular=spec)


# --------------------------------------------------------------------------------------------------
def cube_to_dir(s, x, y):  # relight.py:75-89
    one = np.ones_like(x)
    return np.stack({0: (one, -y, -x), 1: (-one, -y, x), 2: (x, one, y), 3: (x, -one, -y), 4: (x, -y, one),
                     5: (-x, -y, -one)}[s], axis=-1)


def texture2d_linear_wrap(tex, uv):
    """nvdiffrast dr.texture(tex[None], uv[None], filter_mode='linear') with its default boundary_mode='wrap':
    tex [H, W, C], uv [..., 2] in texture units (u along W).  Texel centres at (i + 0.5) / size."""
    Ht, Wt, _ = tex.shape
    u = uv[..., 0].astype(_f32) * _f32(Wt) - _f32(0.5)
    v = uv[..., 1].astype(_f32) * _f32(Ht) - _f32(0.5)
    iu0, iv0 = np.floor(u), np.floor(v)
    fu, fv = (u - iu0).astype(_f32), (v - iv0).astype(_f32)
    iu0, iv0 = iu0.astype(np.int64), iv0.astype(np.int64)
    iu1, iv1 = iu0 + 1, iv0 + 1
    iu0, iu1, iv0, iv1 = iu0 % Wt, iu1 % Wt, iv0 % Ht, iv1 % Ht
    fu, fv = fu[..., None], fv[..., None]
    a = tex[iv0, iu0] * (1 - fu) + tex[iv0, iu1] * fu
    b = tex[iv1, iu0] * (1 - fu) + tex[iv1, iu1] * fu
    return (a * (1 - fv) + b * fv).astype(_f32)


def latlong_to_cubemap(latlong, res):
    """relight.py:92-111 (float32 throughout, like the torch code)."""
    latlong = np.ascontiguousarray(latlong, _f32)
    cube = np.zeros((6, res[0], res[1], latlong.shape[-1]), _f32)
    ly = np.linspace(-1.0 + 1.0 / res[0], 1.0 - 1.0 / res[0], res[0], dtype=_f32)
    lx = np.linspace(-1.0 + 1.0 / res[1], 1.0 - 1.0 / res[1], res[1], dtype=_f32)
    gy, gx = np.meshgrid(ly, lx, indexing="ij")
    for s in range(6):
        v = cube_to_dir(s, gx, gy).astype(_f32)
        v = v / np.maximum(np.sqrt((v * v).sum(-1, keepdims=True, dtype=_f32)), _f32(1e-12))
        tu = np.arctan2(v[..., 0:1], -v[..., 2:3]).astype(_f32) / _f32(2 * np.pi) + _f32(0.5)
        tv = np.arccos(np.clip(v[..., 1:2], -1, 1)).astype(_f32) / _f32(np.pi)
        cube[s] = texture2d_linear_wrap(latlong, np.concatenate([tu, tv], -1))
    return cube


def relight_view(orc, sc, cam, gi, sh_degree, diffuse, spec, alpha_mask=None, albedo_ratio=(1.0, 1.0, 1.0),
                 metallic=False, tone=False, gamma=False, pad_normal=False):
    """relight.py:153-251 for one view with pre-built light levels (build_mips runs once per run, :141).
    Quirks kept: roughness is NOT remapped here; `metallic=True` shades with the metallic map but feeds SSR
    F0 = 0.04 and a zero metallic plane, `metallic=False` feeds F0 = (1 - 0) * 0.04 + albedo * metallic_map (:236-240);
    the albedo ratio scales the shade's albedo only (:216), SSR sees the unscaled albedo (:243)."""
    H, W = cam["image_height"], cam["image_width"]
    fx, fy = focal(cam)
    raw = operator_forward(orc, sc, cam, gi, sh_degree, inference=True)
    post = gbuffer_post(orc, raw, cam["viewmatrix"], pad_normal=pad_normal)
    vd = canonical_view_dirs(cam)
    ratio = np.asarray(albedo_ratio, _f32)[:, None, None]
    direct = shade_direct(orc, post, vd, (post["albedo_map"] * ratio).astype(_f32), post["roughness_map"],
                          post["metallic_map"] if metallic else None, post["occlusion_map"], diffuse, spec, brdf_lut(),
                          tone=tone, gamma=gamma)
    if metallic:
        F0 = np.full_like(post["albedo_map"], 0.04)
        metal = np.zeros_like(post["roughness_map"])
    else:
        metal = post["metallic_map"]
        F0 = (_f32(1.0 - 0.0) * _f32(0.04) + post["albedo_map"] * metal).astype(_f32)
    lin = srgb_to_linear(direct)
    a = (gi["radius"], gi["bias"], gi["thick"], gi["delta"], gi["step"], gi["start"])
    irr, _ = orc.ssr(W, H, fx, fy, *a, post["out_normal_view"], post["depth_pos"], lin, post["albedo_map"],
                     post["roughness_map"], metal, F0)
    irr_s = orc.median3x3(linear_to_srgb(irr))
    render_rgb = (direct + irr_s).astype(_f32)
    if alpha_mask is not None:
        render_rgb = (render_rgb * np.asarray(alpha_mask, _f32)).astype(_f32)
    return dict(render_rgb=render_rgb, render_direct=direct, IRR=irr, occlusion=post["occlusion_map"], post=post)
