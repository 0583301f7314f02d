"""pbr_shading with the reference's signature and result dict (pbr/shade.py:108-241); the three
texture lookups and all elementwise work of the reference run as ONE fused HIP kernel per
direction (gigs_shade_fwd / gigs_shade_bwd in libgigs_hip.so)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Union

import numpy as np
import torch

import gigs_lib

from .light import CubemapLight

_lib = gigs_lib.lib()


def saturate_dot(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return (a * b).sum(dim=-1, keepdim=True).clamp(min=1e-4, max=1.0)


def aces_film(rgb: Union[np.ndarray, torch.Tensor]) -> Union[np.ndarray, torch.Tensor]:
    a, b, c, d, e = 2.51, 0.03, 2.43, 0.59, 0.14
    rgb = (rgb * (a * rgb + b)) / (rgb * (c * rgb + d) + e)
    if isinstance(rgb, np.ndarray):
        return rgb.clip(min=0.0, max=1.0)
    return rgb.clamp(min=0.0, max=1.0)


def linear_to_srgb(linear: Union[np.ndarray, torch.Tensor]) -> Union[np.ndarray, torch.Tensor]:
    if isinstance(linear, torch.Tensor):
        eps = torch.finfo(torch.float32).eps
        srgb0 = 323 / 25 * linear
        srgb1 = (211 * torch.clamp(linear, min=eps) ** (5 / 12) - 11) / 200
        return torch.where(linear <= 0.0031308, srgb0, srgb1)
    elif isinstance(linear, np.ndarray):
        eps = np.finfo(np.float32).eps
        srgb0 = 323 / 25 * linear
        srgb1 = (211 * np.maximum(eps, linear) ** (5 / 12) - 11) / 200
        return np.where(linear <= 0.0031308, srgb0, srgb1)
    raise NotImplementedError


def get_brdf_lut() -> torch.Tensor:
    """256x256x2 fp32 split-sum LUT (data file copied verbatim from the reference, pbr/shade.py:100-105)."""
    path = os.path.join(os.path.dirname(__file__), "brdf_256_256.bin")
    return torch.from_numpy(np.fromfile(path, dtype=np.float32).reshape(1, 256, 256, 2))


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


class _PbrShade(torch.autograd.Function):
    """inputs: normals, view_dirs, albedo, roughness, mask(u8), occlusion|None, metallic|None,
    background|None, lut, tone, gamma, diffuse, *specular -> render_rgb, diffuse_rgb, specular_rgb,
    diffuse_light (all [H,W,3])."""

    @staticmethod
    def forward(ctx, normals, view_dirs, albedo, roughness, mask, occlusion, metallic, background, lut, tone, gamma,
                diffuse, *specular):
        dev = normals.device
        H, W, _ = normals.shape
        ctx.set_materialize_grads(False)  # an output the loss does not use: None, not a zero tensor (the kernel takes NULL)
        f = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        normals, view_dirs, albedo, roughness = f(normals), f(view_dirs), f(albedo), f(roughness)
        occlusion, metallic, background = f(occlusion), f(metallic), f(background)
        lut, diffuse = f(lut), f(diffuse)
        specular = [f(s) for s in specular]
        mask8 = mask.contiguous().to(torch.uint8)
        outs = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(4)]
        spec_ptr = _ptr_array(specular)
        spec_res = (C.c_int * len(specular))(*[int(s.shape[1]) for s in specular])
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        with torch.cuda.device(dev):
            gigs_lib.check(_lib.gigs_shade_fwd(
                H, W, p(normals), p(view_dirs), p(albedo), p(roughness), p(mask8), p(occlusion), p(metallic),
                p(background), p(diffuse), int(diffuse.shape[1]), len(specular), spec_ptr, spec_res, p(lut),
                int(lut.shape[-2]), int(lut.shape[-3]), int(bool(tone)), int(bool(gamma)), outs[0].data_ptr(),
                outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), torch.cuda.current_stream().cuda_stream),
                "shade_fwd")
        ctx.save_for_backward(normals, view_dirs, albedo, roughness, mask8, occlusion, metallic, lut, diffuse, *specular)
        ctx.flags = (bool(tone), bool(gamma))
        ctx.needs = (ctx.needs_input_grad[11], [ctx.needs_input_grad[12 + i] for i in range(len(specular))])
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_render, g_diffuse_rgb, g_specular_rgb, g_diffuse_light):
        (normals, view_dirs, albedo, roughness, mask8, occlusion, metallic, lut, diffuse, *specular) = ctx.saved_tensors
        tone, gamma = ctx.flags
        dev = normals.device
        H, W, _ = normals.shape
        g = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        gs = [g(x) for x in (g_render, g_diffuse_rgb, g_specular_rgb, g_diffuse_light)]
        d_albedo = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        d_rough = torch.empty((H, W, 1), dtype=torch.float32, device=dev)
        d_metal = torch.empty((H, W, 1), dtype=torch.float32, device=dev) if metallic is not None else None
        need_d, need_s = ctx.needs
        d_diffuse = torch.zeros_like(diffuse) if need_d else None
        d_spec = [torch.zeros_like(s) if n else None for s, n in zip(specular, need_s)]
        spec_ptr = _ptr_array(specular)
        dspec_ptr = _ptr_array(d_spec)
        spec_res = (C.c_int * len(specular))(*[int(s.shape[1]) for s in specular])
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        with torch.cuda.device(dev):
            gigs_lib.check(_lib.gigs_shade_bwd(
                H, W, p(normals), p(view_dirs), p(albedo), p(roughness), p(mask8), p(occlusion), p(metallic),
                p(diffuse), int(diffuse.shape[1]), len(specular), spec_ptr, spec_res, p(lut), int(lut.shape[-2]),
                int(lut.shape[-3]), int(tone), int(gamma), p(gs[0]), p(gs[1]), p(gs[2]), p(gs[3]), p(d_albedo),
                p(d_rough), p(d_metal), p(d_diffuse), dspec_ptr, torch.cuda.current_stream().cuda_stream), "shade_bwd")
        return (None, None, d_albedo, d_rough, None, None, d_metal, None, None, None, None, d_diffuse, *d_spec)


def pbr_shading(
    light: CubemapLight,
    normals: torch.Tensor,  # [H, W, 3]
    view_dirs: torch.Tensor,  # [H, W, 3]
    albedo: torch.Tensor,  # [H, W, 3]
    roughness: torch.Tensor,  # [H, W, 1]
    mask: torch.Tensor,  # [H, W, 1]
    tone: bool = False,
    gamma: bool = False,
    occlusion: Optional[torch.Tensor] = None,  # [H, W, 1]
    metallic: Optional[torch.Tensor] = None,
    brdf_lut: Optional[torch.Tensor] = None,
    background: Optional[torch.Tensor] = None,
) -> Dict:
    if not normals.is_cuda:
        raise RuntimeError("normals must be a CUDA/HIP tensor: pbr (gigs-hip) has no CPU path")
    H, W, _ = normals.shape
    # normals / view_dirs / occlusion carry no gradient in the reference's call sites (they are
    # detached or constants: train.py:343, 351); the fused backward returns None for them.
    render_rgb, diffuse_rgb, specular_rgb, diffuse_light = _PbrShade.apply(
        normals.reshape(H, W, 3), view_dirs.reshape(H, W, 3), albedo.reshape(H, W, 3), roughness.reshape(H, W, 1),
        mask.reshape(H, W, 1), None if occlusion is None else occlusion.reshape(H, W, 1),
        None if metallic is None else metallic.reshape(H, W, 1), background, brdf_lut, tone, gamma, light.diffuse,
        *light.specular)
    return {"diffuse_light": diffuse_light, "render_rgb": render_rgb, "diffuse_rgb": diffuse_rgb,
            "specular_rgb": specular_rgb}
