"""Cube-map lookups for a list of directions: the `dr.texture(cubemap[None], dirs[None], filter_mode="linear",
boundary_mode="cube")` calls the reference makes outside pbr_shading (train.py:409-417 for the envmap TV term,
render.py:80 and relight.py:108 to export the environment map), on libgigs_hip with the sampling rule written in
include/gigs_hip.h (nvdiffrast is third-party and absent here: parity unpinned beyond that rule)."""
from __future__ import annotations

import torch

import gigs_lib

_lib = gigs_lib.lib()


class _CubeTexture(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cubemap, dirs, planar):
        if not cubemap.is_cuda:
            raise RuntimeError("cube_texture needs CUDA/HIP tensors: gigs-hip has no CPU path")
        if cubemap.dim() != 4 or cubemap.shape[0] != 6 or cubemap.shape[1] != cubemap.shape[2] or cubemap.shape[3] != 3:
            raise ValueError("cube_texture: cubemap must be [6,res,res,3]")
        if dirs.shape[-1] != 3:
            raise ValueError("cube_texture: dirs must be [...,3]")
        cubemap = cubemap.contiguous().float()
        d = dirs.contiguous().float()
        lead = tuple(d.shape[:-1])
        n = d.numel() // 3
        out = torch.empty(((3,) + lead) if planar else (lead + (3,)), dtype=torch.float32, device=cubemap.device)
        with torch.cuda.device(cubemap.device):
            gigs_lib.check(_lib.gigs_cube_texture_fwd(int(cubemap.shape[1]), cubemap.data_ptr(), n, d.data_ptr(),
                                                      out.data_ptr(), int(bool(planar)),
                                                      torch.cuda.current_stream().cuda_stream), "cube_texture_fwd")
        ctx.save_for_backward(d)
        ctx.res, ctx.planar, ctx.n = int(cubemap.shape[1]), bool(planar), n
        return out

    @staticmethod
    def backward(ctx, g_out):
        (d,) = ctx.saved_tensors
        g = g_out.contiguous().float()
        d_tex = torch.zeros((6, ctx.res, ctx.res, 3), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            gigs_lib.check(_lib.gigs_cube_texture_bwd(ctx.res, ctx.n, d.data_ptr(), g.data_ptr(), d_tex.data_ptr(),
                                                      int(ctx.planar), torch.cuda.current_stream().cuda_stream),
                           "cube_texture_bwd")
        return d_tex, None, None


def cube_texture(cubemap: torch.Tensor, dirs: torch.Tensor, planar: bool = False) -> torch.Tensor:
    """cubemap [6,res,res,3], dirs [...,3] -> [...,3] (or [3,...] planes); differentiable w.r.t. the cubemap."""
    return _CubeTexture.apply(cubemap, dirs, planar)
