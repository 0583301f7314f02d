"""Relighting a trained scene under a new environment map: the per-view operator sequence of the reference's
relight.py, on the HIP library (BASELINE config C3: inference-only PBR + indirect).

    latlong_to_cubemap(latlong_map, res)      relight.py:92-111   (gigs_latlong_to_cubemap)
    make_light(hdri, res=256)                 relight.py:278-285  (CubemapLight + latlong_to_cubemap, eval mode)
    Relighter(...)(cam, gaussians, ...)       relight.py:153-251  per view: render(inference=True) -> pbr_shading ->
                                              Gaussian_SSR -> linear_to_srgb -> 3x3 median -> + direct -> * alpha_mask
                                              with light.build_mips() ONCE per run (:141), not per view

Two formulations with identical results (tests/test_gpu_relight.py): `fused=False` is the reference's op sequence
spelled with this package's drop-in operators (pipeline.render, pbr.pbr_shading, Gaussian_SSR, torch elementwise
ops); `fused=True` (default) runs the same arithmetic as five library launches after the rasterizer
(gigs_gbuffer_post, gigs_shade_fwd_ex in planar layout with the sRGB->linear epilogue, gigs_ssr,
gigs_stage2_loss_fwd for linear_to_srgb + median + sum) on the rasterizer's [C,H,W] planes.

Reference quirks kept on purpose: the `metallic` branches for F0 are swapped (relight.py:236-240):
with metallic=True the shade uses the metallic map but SSR receives F0 = 0.04 and a zero metallic plane; with
metallic=False (`metallic` is the Python bool) SSR receives F0 = (1 - False) * 0.04 + albedo * metallic_map.  The
per-channel albedo ratio read from albedo_ratio.json (:203-220) scales the shade's albedo only.  Image I/O
(read_hdr, save_image, the JSON) stays with the caller.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

import gigs_lib
import pipeline
from diff_gaussian_rasterization import Gaussian_SSR, _C as _ops, filters
from pbr import CubemapLight, get_brdf_lut, pbr_shading
from pbr.shade import _ptr_array

_lib = gigs_lib.lib()


def latlong_to_cubemap(latlong_map: torch.Tensor, res: List[int]) -> torch.Tensor:
    """relight.py:92-111: [H, W, C] equirectangular map -> [6, res[0], res[1], C] cubemap."""
    if not latlong_map.is_cuda:
        raise RuntimeError("latlong_map must be a CUDA/HIP tensor: gigs-hip has no CPU path")
    lat = latlong_map.contiguous().float()
    Hl, Wl, Cn = lat.shape
    cube = torch.empty((6, int(res[0]), int(res[1]), Cn), dtype=torch.float32, device=lat.device)
    with torch.cuda.device(lat.device):
        gigs_lib.check(_lib.gigs_latlong_to_cubemap(int(res[0]), int(res[1]), Hl, Wl, Cn, lat.data_ptr(), cube.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream), "latlong_to_cubemap")
    return cube


def make_light(hdri: torch.Tensor, res: int = 256) -> CubemapLight:
    """relight.py:278-285."""
    light = CubemapLight(base_res=res, device=hdri.device)
    light.base.data = latlong_to_cubemap(hdri, [res, res])
    light.eval()
    return light


class Relighter:
    """render_set (relight.py:113-251) without the file I/O: build_mips once, then one call per view."""

    def __init__(self, light: CubemapLight, gi: Dict, sh_degree: int, metallic: bool = False, tone: bool = False,
                 gamma: bool = False, fused: bool = True, pad_normal: bool = False, brdf_lut: Optional[torch.Tensor] = None,
                 graphs: bool = False):
        """graphs=True (with fused): the whole view -- rasterizer under asynchronous binning, filters, SSAO, shade, SSR,
        sRGB / median / sum -- is captured once into ONE hipGraph and replayed per view (camera pose and view
        directions are its inputs; image size, field of view, GI settings and the Gaussian tensors are baked in).  The
        tensors it returns are the graph's static outputs: consume them before the next call."""
        self.light, self.gi, self.sh_degree = light, gi, sh_degree
        self.metallic, self.tone, self.gamma = bool(metallic), bool(tone), bool(gamma)
        self.fused, self.pad_normal = bool(fused) and not pad_normal, bool(pad_normal)
        self.graphs = bool(graphs) and self.fused
        self._graph = self._graph_key = self._bin = None
        self._capacity = 0
        dev = light.base.device
        self.brdf_lut = (brdf_lut if brdf_lut is not None else get_brdf_lut()).to(dev)
        with torch.no_grad():
            light.build_mips()  # relight.py:141: once per run
        self._scratch = {}

    def _buf(self, name, shape, dtype, dev):
        t = self._scratch.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != dev:
            t = self._scratch[name] = torch.empty(shape, dtype=dtype, device=dev)
        return t

    @torch.no_grad()
    def __call__(self, cam: Dict, g: Dict[str, torch.Tensor], view_dirs: torch.Tensor,
                 alpha_mask: Optional[torch.Tensor] = None, albedo_ratio: Optional[Sequence[float]] = None) -> Dict:
        if self.graphs:
            try:
                return self._graphed(cam, g, view_dirs, alpha_mask, albedo_ratio)
            except pipeline.DenseScene:
                self.graphs = False
        if self.fused:
            return self._fused(cam, g, view_dirs, alpha_mask, albedo_ratio)
        return self._unfused(cam, g, view_dirs, alpha_mask, albedo_ratio)

    # -- the fused sequence replayed from one hipGraph --------------------------------------------------------------
    def _graphed(self, cam, g, view_dirs, alpha_mask, albedo_ratio):
        from diff_gaussian_rasterization import AsyncBinning, BinningOverflow
        names = ("render_rgb", "render_direct", "IRR", "occlusion", "depth_map", "normal_map", "normal_mask", "radii")
        key = (int(cam["image_height"]), int(cam["image_width"]), float(cam["tanfovx"]), float(cam["tanfovy"]),
               tuple(sorted((k, v.data_ptr()) for k, v in g.items())), None if albedo_ratio is None else tuple(albedo_ratio))
        for _ in range(4):
            if self._graph is None or self._graph_key != key:
                if self._capacity <= 0:
                    probe = pipeline.GraphedRaster(cam, g, self.gi, self.sh_degree, inference=True)
                    r = probe._probe(cam, g, torch.zeros(3, device=g["means3D"].device))
                    tiles = ((int(cam["image_height"]) + 15) // 16) * ((int(cam["image_width"]) + 15) // 16)
                    if pipeline._declined_as_dense(r, tiles):
                        raise pipeline.DenseScene(f"{r} instances over {tiles} tiles")
                    self._capacity = max(65536, -(-2 * r // 65536) * 65536)
                self._bin = AsyncBinning(self._capacity, g["means3D"].device)
                scalars = {k: v for k, v in cam.items() if not isinstance(v, torch.Tensor)}

                def core(viewmatrix, projmatrix, campos, vd):
                    c = dict(scalars, viewmatrix=viewmatrix, projmatrix=projmatrix, campos=campos)
                    o = self._fused(c, g, vd, None, albedo_ratio)
                    return tuple(o[n] for n in names)

                with self._bin:
                    self._graph = pipeline._graphed_inference(core, (cam["viewmatrix"], cam["projmatrix"], cam["campos"], view_dirs))
                self._graph_key = key
            out = dict(zip(names, self._graph(cam["viewmatrix"], cam["projmatrix"], cam["campos"], view_dirs)))
            self._bin.snapshot()
            if alpha_mask is not None:
                out["render_rgb"] = out["render_rgb"] * alpha_mask
            try:
                out["num_rendered"] = self._bin.check()
                return out
            except BinningOverflow as ex:
                self._capacity = -(-int(1.5 * ex.needed) // 65536) * 65536
                self.close()
        raise RuntimeError("Relighter: the binning capacity kept overflowing")

    def close(self) -> None:
        """Release the view graph with the device idle before and after (see pipeline.WholeStepGraph._drop_graphs)."""
        if self._graph is not None:
            torch.cuda.synchronize()
            self._graph = self._graph_key = None
            torch.cuda.synchronize()

    # -- the reference's op sequence, operator by operator ------------------------------------------------------
    def _unfused(self, cam, g, view_dirs, alpha_mask, albedo_ratio):
        dev = g["means3D"].device
        gi = self.gi
        background = torch.zeros(3, device=dev)
        r = pipeline.render(cam, g, self.sh_degree, background, gi, inference=True, derive_normal=True,
                            pad_normal=self.pad_normal)
        H, W = cam["image_height"], cam["image_width"]
        normal_mask = r["normal_mask"]
        albedo_map, roughness_map, metallic_map = r["albedo_map"], r["roughness_map"], r["metallic_map"]
        ratio = torch.ones(3, device=dev) if albedo_ratio is None else torch.as_tensor(albedo_ratio, dtype=torch.float32, device=dev)
        res = pbr_shading(light=self.light, normals=r["normal_map"].permute(1, 2, 0), view_dirs=view_dirs,
                          mask=normal_mask.permute(1, 2, 0), albedo=(albedo_map * ratio[:, None, None]).permute(1, 2, 0),
                          roughness=roughness_map.permute(1, 2, 0),
                          metallic=metallic_map.permute(1, 2, 0) if self.metallic else None, tone=self.tone,
                          occlusion=r["occlusion_map"].permute(1, 2, 0), gamma=self.gamma, brdf_lut=self.brdf_lut)
        render_direct = res["render_rgb"].permute(2, 0, 1)
        render_direct = torch.where(normal_mask, render_direct, background[:, None, None])
        ssr = Gaussian_SSR(cam["tanfovx"], cam["tanfovy"], W, H, gi["radius"], gi["bias"], gi["thick"], gi["delta"],
                           gi["step"], gi["start"])
        if self.metallic:  # relight.py:236-240, as written
            F0 = torch.ones_like(albedo_map) * 0.04
            metallic_in = torch.zeros_like(roughness_map)
        else:
            F0 = (1.0 - float(self.metallic)) * 0.04 + albedo_map * metallic_map
            metallic_in = metallic_map
        linear_rgb = pipeline.srgb_to_linear(render_direct)
        IRR, _ = ssr(r["out_normal_view"], r["depth_pos"], linear_rgb, albedo_map, roughness_map, metallic_in, F0)
        IRR_s = filters.median_blur(pipeline.linear_to_srgb(IRR)[None, ...], (3, 3))[0]
        render_rgb = render_direct + IRR_s
        if alpha_mask is not None:
            render_rgb = render_rgb * alpha_mask
        return dict(render_rgb=render_rgb, render_direct=render_direct, IRR=IRR, occlusion=r["occlusion_map"],
                    depth_map=r["depth_map"], normal_map=r["normal_map"], normal_mask=normal_mask, radii=r["radii"])

    # -- the same arithmetic as five launches behind the rasterizer ----------------------------------------------
    def _fused(self, cam, g, view_dirs, alpha_mask, albedo_ratio):
        dev = g["means3D"].device
        gi = self.gi
        background = torch.zeros(3, device=dev)
        (out, _, st) = pipeline.rasterize(cam, g, self.sh_degree, background, gi, inference=True, derive_normal=True)
        (_, radii, _, depth_map, _, normal_map, occlusion, albedo_map, roughness_map, metallic_map, out_normal_view,
         depth_pos) = out
        H, W = cam["image_height"], cam["image_width"]
        new = lambda name, *shape: self._buf(name, shape, torch.float32, dev)  # noqa: E731
        normals_view, onv = new("normals_view", 3, H, W), new("onv", 3, H, W)
        mask_u8 = self._buf("mask_u8", (H, W), torch.uint8, dev)
        mask_f = new("mask_f", 1, H, W)
        render_direct, linear_rgb = torch.empty((3, H, W), device=dev), new("linear_rgb", 3, H, W)
        render_rgb, acc, loss = torch.empty((3, H, W), device=dev), new("acc", 4 + 4 * 256), new("loss", 1)
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        albedo_shade = albedo_map
        if albedo_ratio is not None:
            albedo_shade = albedo_map * torch.as_tensor(albedo_ratio, dtype=torch.float32, device=dev)[:, None, None]
        light = self.light
        spec = [s.contiguous() for s in light.specular]
        spec_ptr = _ptr_array(spec)
        spec_res = (C.c_int * len(spec))(*[int(s.shape[1]) for s in spec])
        lut = self.brdf_lut
        ext = gigs_lib.ShadeExt(planar=1, rough_scale=1.0, rough_bias=0.0, out_linear=p(linear_rgb))
        vm = st.viewmatrix.contiguous().float()
        vd = view_dirs.contiguous().float()
        with torch.cuda.device(dev):
            s = torch.cuda.current_stream().cuda_stream
            gigs_lib.check(_lib.gigs_gbuffer_post(H, W, p(normal_map), p(out_normal_view), p(vm), p(normals_view), p(mask_u8),
                                                  p(mask_f), p(onv), s), "gbuffer_post")
            gigs_lib.check(_lib.gigs_shade_fwd_ex(
                gigs_lib.ctx_ptr(), H, W, p(normals_view), p(vd), p(albedo_shade), p(roughness_map), p(mask_u8), p(occlusion),
                p(metallic_map) if self.metallic else None, None, p(light.diffuse), int(light.diffuse.shape[1]), len(spec),
                spec_ptr, spec_res, p(lut), int(lut.shape[-2]), int(lut.shape[-3]), int(self.tone), int(self.gamma),
                p(render_direct), None, None, None, C.addressof(ext), s), "shade_fwd_ex")
            if self.metallic:
                F0 = torch.full_like(albedo_map, 0.04)
                metallic_in = torch.zeros_like(roughness_map)
            else:
                F0 = torch.addcmul(torch.full_like(albedo_map, (1.0 - float(self.metallic)) * 0.04), albedo_map, metallic_map)
                metallic_in = metallic_map
            IRR, _ = _ops.SSR(W, H, W / (2.0 * cam["tanfovx"]), H / (2.0 * cam["tanfovy"]), gi["radius"], gi["bias"],
                              gi["thick"], gi["delta"], gi["step"], gi["start"], onv, depth_pos, linear_rgb, albedo_map,
                              roughness_map, metallic_in, F0)
            # render_rgb = render_direct + median3x3(linear_to_srgb(IRR)); the loss this entry point also forms is unused
            gigs_lib.check(_lib.gigs_stage2_loss_fwd(H, W, p(render_direct), p(IRR), p(render_direct), p(mask_f),
                                                     p(roughness_map), p(metallic_in), p(render_rgb), p(acc), p(loss), s),
                           "stage2_loss_fwd")
        if alpha_mask is not None:
            render_rgb = render_rgb * alpha_mask
        return dict(render_rgb=render_rgb, render_direct=render_direct, IRR=IRR, occlusion=occlusion, depth_map=depth_map,
                    normal_map=normals_view, normal_mask=mask_u8.bool()[None], radii=radii)
