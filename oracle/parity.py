"""Full-size parity harness: one view of the stage-2 path on the GPU (product code) against the CPU oracle
composition (oracle/stage2_ref.py), as one report.

TEST INFRASTRUCTURE ONLY: used by bench.py's cpu_baseline leg (`parity_c2` in the JSON line) and by
tests/test_gpu_fullsize.py.  The product never imports this; this module imports the product lazily to drive it.

Report fields (north_star: bit-exact tile/point indices, fp planes within 1e-4 mean per-pixel L1):
  radii_equal / point_list_equal / ranges_equal / keys_equal   integer state of A1-A4
  n_contrib_flips            pixels whose last-contributor index differs (exp ulp at the alpha thresholds)
  planes[name].mean_l1/max   the 9 G-buffer planes, normal_from_depth, depth_pos, occlusion, IRR, render_direct, render_rgb
  planes[name].frac_over_1e-4 / frac_over_1e-5   share of the plane's ELEMENTS whose |a-b| exceeds the threshold: where the
                             isolated "max" outliers of the GI planes sit (a march sample that rounds to the neighbouring
                             pixel at a depth edge flips a ray's hit) -- to be read beside march_noise() below
  K_pairs_evaluated / K_pairs_contributing       SURVEY 8(d)'s K from the oracle's blend (pairs walked / pairs blended)
  covered_px_frac            share of pixels with at least one contributor (GI work scales with it)
  grads_rel_l1[name]         rasterizer backward for fixed pixel gradients: mean |a-b| / mean |b|
  psnr_render_rgb            utils/image_utils.py:31 of the final stage-2 image, GPU vs oracle
"""
from __future__ import annotations

import time
from typing import Dict

import numpy as np

from . import stage2_ref

PLANES9 = ["color", "opacity", "depth", "normal", "normal_view", "pos", "albedo", "roughness", "metallic"]
GRAD_PLANES = (("color", 3), ("opacity", 1), ("depth", 1), ("normal", 3), ("albedo", 3), ("roughness", 1), ("metallic", 1))


def pixel_grads(H, W, seed=7, only=None):
    """Deterministic incoming image gradients; `only` = planes that are non-zero (stage 2 feeds albedo/roughness/metallic)."""
    rng = np.random.default_rng(seed)
    g = {}
    for k, c in GRAD_PLANES:
        a = (rng.normal(size=(c, H, W)) / (H * W)).astype(np.float32)
        g[k] = a if (only is None or k in only) else np.zeros_like(a)
    return g


def gpu_capture(sc, cam, gi, sh_degree, light=None, brdf_lut=None, stepper=None, grads_only=None, dev="cuda:0") -> Dict:
    """Runs the product on one view and returns everything as numpy."""
    import torch

    import diff_gaussian_rasterization as dgr
    import gigs_lib
    import pipeline

    lib = gigs_lib.lib()
    keys = stage2_ref.KEYS
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    g = {k: tt(sc[k]) for k in keys}
    camt = {k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    H, W, P = cam["image_height"], cam["image_width"], sc["means3D"].shape[0]
    N, T = H * W, ((W + 15) // 16) * ((H + 15) // 16)
    bg = torch.zeros(3, device=dev)
    e = torch.Tensor([])
    out = {}
    with torch.no_grad():
        res = dgr._C.rasterize_gaussians(bg, g["means3D"], e, g["opacities"], g["normal"], g["albedo"], g["roughness"],
                                         g["metallic"], g["scales"], g["rotations"], e, g["shs"], camt["campos"],
                                         camt["viewmatrix"], camt["projmatrix"], 1.0, cam["tanfovx"], cam["tanfovy"], H, W,
                                         sh_degree, False, False, False, False)
        torch.cuda.synchronize()
        (R, color, radii, geom, binning, img, opacity, depth, normal, normal_view, pos, albedo, rough, metal) = res
        for k, v in zip(PLANES9, (color, opacity, depth, normal, normal_view, pos, albedo, rough, metal)):
            out[k] = v.cpu().numpy()
        out["radii"], out["num_rendered"] = radii.cpu().numpy(), int(R)

        def view(buf, off, dtype, count):
            return buf[off:off + count * np.dtype(dtype).itemsize].cpu().numpy().view(dtype)

        if R > 0:
            out["keys"] = view(binning, lib.gigs_binning_offset(R, 2), np.uint64, R)
            out["point_list"] = view(binning, lib.gigs_binning_offset(R, 3), np.uint32, R)
        out["n_contrib"] = view(img, lib.gigs_image_offset(W, H, 1), np.uint32, N)
        out["ranges"] = view(img, lib.gigs_image_offset(W, H, 2), np.uint32, 2 * T)
        # rasterizer backward on the state of this forward, for fixed pixel gradients
        pg = pixel_grads(H, W, only=grads_only)
        gt = {k: tt(v) for k, v in pg.items()}
        bw = dgr._C.rasterize_gaussians_backward(
            bg, g["means3D"], radii, e, g["normal"], g["albedo"], g["roughness"], g["metallic"], g["scales"], g["rotations"],
            e, g["shs"], camt["campos"], camt["viewmatrix"], camt["projmatrix"], 1.0, cam["tanfovx"], cam["tanfovy"],
            sh_degree, gt["depth"], gt["color"], gt["opacity"], gt["normal"], gt["albedo"], gt["roughness"], gt["metallic"],
            geom, binning, img, R, False)
        names = ["means2D", "colors", "opacity", "normal", "albedo", "roughness", "metallic", "means3D", "cov3D", "sh", "scales",
                 "rotations"]
        out["grads"] = {k: v.cpu().numpy() for k, v in zip(names, bw)}
        # the operator (filters + SSAO)
        op, _, _ = pipeline.rasterize(camt, g, sh_degree, bg, gi)
        out["normal_from_depth"], out["occlusion"], out["depth_pos"] = (op[4].cpu().numpy(), op[6].cpu().numpy(),
                                                                        op[11].cpu().numpy())
    if light is not None:
        gp = {k: v.clone().requires_grad_(True) for k, v in g.items()}
        rays = pipeline.canonical_rays(cam, dev)
        vd = pipeline.view_dirs_for(camt, rays, dev)
        if stepper is None:
            stepper = pipeline.Stage2Step(light, brdf_lut, gi, sh_degree, graphs=False, fused=True)
        gt_image = torch.zeros(3, H, W, device=dev)
        so = stepper(camt, gp, gt_image, vd)
        torch.cuda.synchronize()
        for k in ("render_rgb", "render_direct", "IRR"):
            out[k] = so[k].detach().cpu().numpy()
        out["light_base"] = light.base.detach().cpu().numpy()
    return out


def oracle_capture(orc, sc, cam, gi, sh_degree, light_base=None, grads_only=None):
    """The same quantities from the oracle, with wall-clock sections (the cpu_baseline leg)."""
    t = {}
    t0 = time.perf_counter()
    H, W = cam["image_height"], cam["image_width"]
    if light_base is not None:
        s2 = stage2_ref.stage2_forward(orc, sc, cam, gi, sh_degree, light_base, keep_state=True, timings=t)
        raw = s2["raw"]
    else:
        s2 = None
        raw = stage2_ref.operator_forward(orc, sc, cam, gi, sh_degree, keep_state=True, timings=t)
    r = raw["rasterizer"]
    t1 = time.perf_counter()
    pg = pixel_grads(H, W, only=grads_only)
    grads = r.backward(**{"grad_" + k: v for k, v in pg.items()})
    t["backward"] = time.perf_counter() - t1
    t["total"] = time.perf_counter() - t0
    out = dict(color=raw["render"], opacity=raw["opacity_map"], depth=raw["depth_map"], normal=raw["normal_map"],
               normal_view=raw["out_normal_view"], pos=raw["pos_raw"], albedo=raw["albedo_map"], roughness=raw["roughness_map"],
               metallic=raw["metallic_map"], radii=raw["radii"], num_rendered=raw["num_rendered"],
               normal_from_depth=raw["normal_map_from_depth"], occlusion=raw["occlusion_map"], depth_pos=raw["depth_pos"],
               grads=grads)
    for k in ("keys", "point_list", "n_contrib", "ranges"):
        out[k] = r.state(k)
    c = r.counters()
    out["K_pairs_evaluated"], out["K_pairs_contributing"] = c["pairs_evaluated"], c["pairs_contributing"]
    if s2 is not None:
        out.update(render_rgb=s2["render_rgb"], render_direct=s2["render_direct"], IRR=s2["IRR"])
    return out, t


GI_PLANES = ("occlusion", "IRR", "render_direct", "render_rgb")


def plane_stats(d: np.ndarray, nan_equal: bool = True) -> Dict:
    """d = |a - b| (NaNs already zeroed)."""
    return {"mean_l1": float(d.mean()), "max": float(d.max()), "nan_pattern_equal": bool(nan_equal),
            "frac_over_1e-4": float((d > 1e-4).mean()), "frac_over_1e-5": float((d > 1e-5).mean())}


def diff_planes(a: Dict, b: Dict, keys=GI_PLANES) -> Dict:
    out = {}
    for k in keys:
        if k in a and k in b:
            x, y = a[k], b[k]
            d = np.abs(np.nan_to_num(x.astype(np.float64)) - np.nan_to_num(y.astype(np.float64)))
            out[k] = plane_stats(d, np.array_equal(np.isnan(x), np.isnan(y)))
    return out


def gpu_exact_march(sc, cam, gi, sh_degree, light, brdf_lut, dev="cuda:0") -> Dict:
    """The GI planes of the same view from the product with the EXACT march (gi_march = exact: the oracle's sample
    arithmetic bit for bit -- the checker of the default, projective march), through an eager fused step."""
    import torch

    import gigs_lib
    import pipeline

    with gigs_lib.options(gi_march="exact"):  # gigs_options.gi_march of the library context the operators run with
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        g = {k: tt(sc[k]) for k in stage2_ref.KEYS}
        camt = {k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
        H, W = cam["image_height"], cam["image_width"]
        out = {}
        with torch.no_grad():
            op, _, _ = pipeline.rasterize(camt, g, sh_degree, torch.zeros(3, device=dev), gi)
            out["occlusion"] = op[6].cpu().numpy()
        gp = {k: v.clone().requires_grad_(True) for k, v in g.items()}
        vd = pipeline.view_dirs_for(camt, pipeline.canonical_rays(cam, dev), dev)
        so = pipeline.Stage2Step(light, brdf_lut, gi, sh_degree, graphs=False, fused=True)(
            camt, gp, torch.zeros(3, H, W, device=dev), vd)
        torch.cuda.synchronize()
        for k in ("render_rgb", "render_direct", "IRR"):
            out[k] = so[k].detach().cpu().numpy()
        return out


def march_noise(orc, sc, cam, gi, sh_degree, gpu: Dict, ref: Dict, light=None, brdf_lut=None, dev="cuda:0") -> Dict:
    """The per-pixel evidence for the GI planes, three comparisons on ONE view, same fields each (plane_stats):
      default_vs_oracle   the product's default (projective, certified) march against the oracle
      exact_vs_oracle     the product's exact march against the oracle (what the bit-exactness claims refer to)
      oracle_vs_fma_twin  the oracle against its FMA-contracted twin (libgigs_oracle_fma.so, nvcc's default contraction):
                          two legitimate fp32 compilations of the cited reference lines -- the noise floor of
                          "parity with the CUDA binary"
    A reader sees whether the default march's isolated outliers (max ~1e-3) are as rare as the reference's own compile
    noise."""
    rep = {"default_vs_oracle": diff_planes(gpu, ref)}
    if light is not None:
        rep["exact_vs_oracle"] = diff_planes(gpu_exact_march(sc, cam, gi, sh_degree, light, brdf_lut, dev), ref)
    twin = orc.variant("fma")
    twin.set_threads(orc.max_threads())
    if "light_base" in gpu:
        s2 = stage2_ref.stage2_forward(twin, sc, cam, gi, sh_degree, gpu["light_base"], keep_state=False)
        tw = dict(occlusion=s2["raw"]["occlusion_map"], IRR=s2["IRR"], render_direct=s2["render_direct"],
                  render_rgb=s2["render_rgb"])
    else:
        raw = stage2_ref.operator_forward(twin, sc, cam, gi, sh_degree, keep_state=False)
        tw = dict(occlusion=raw["occlusion_map"])
    rep["oracle_vs_fma_twin"] = diff_planes(ref, tw)
    return rep


def compare(gpu: Dict, ref: Dict) -> Dict:
    rep = {"num_rendered": [int(gpu["num_rendered"]), int(ref["num_rendered"])]}
    rep["radii_equal"] = bool(np.array_equal(gpu["radii"], ref["radii"]))
    for k in ("keys", "point_list", "ranges"):
        rep[k + "_equal"] = bool(k in gpu and gpu[k].shape == ref[k].shape and np.array_equal(gpu[k], ref[k]))
    rep["n_contrib_flips"] = int((gpu["n_contrib"] != ref["n_contrib"]).sum())
    planes = {}
    for k in PLANES9 + ["normal_from_depth", "depth_pos", "occlusion", "IRR", "render_direct", "render_rgb"]:
        if k not in gpu or k not in ref:
            continue
        a, b = gpu[k], ref[k]
        nan_equal = bool(np.array_equal(np.isnan(a), np.isnan(b)))
        d = np.abs(np.nan_to_num(a.astype(np.float64)) - np.nan_to_num(b.astype(np.float64)))
        planes[k] = plane_stats(d, nan_equal)
    rep["planes"] = planes
    rep["covered_px_frac"] = float((ref["n_contrib"] > 0).mean())
    for k in ("K_pairs_evaluated", "K_pairs_contributing"):
        if k in ref:
            rep[k] = int(ref[k])
    rep["worst_plane_mean_l1"] = max(v["mean_l1"] for v in planes.values())
    gr = {}
    gmap = {"means2D": "means2D", "colors": "colors", "opacity": "opacity", "normal": "normal", "albedo": "albedo",
            "roughness": "roughness", "metallic": "metallic", "means3D": "means3D", "cov3D": "cov3D", "sh": "sh",
            "scales": "scales", "rotations": "rotations"}
    for k, rk in gmap.items():
        a = gpu["grads"][k].astype(np.float64)
        b = np.asarray(ref["grads"][rk], np.float64).reshape(a.shape)
        if k == "cov3D" and np.abs(b).max() == 0:
            continue
        gr[k] = float(np.abs(a - b).mean() / max(np.abs(b).mean(), 1e-30))
    rep["grads_rel_l1"] = gr
    rep["worst_grad_rel_l1"] = max(gr.values()) if gr else None
    if "render_rgb" in gpu and "render_rgb" in ref:
        rep["psnr_render_rgb"] = round(stage2_ref.psnr(np.nan_to_num(gpu["render_rgb"]), np.nan_to_num(ref["render_rgb"])), 2)
    rep["psnr_color"] = round(stage2_ref.psnr(gpu["color"], ref["color"]), 2)
    return rep
