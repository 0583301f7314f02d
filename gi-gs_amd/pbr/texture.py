"""Cube-map lookups for a list of directions: the `dr.texture(cubemap[None], dirs[None], filter_mode="linear",
boundary_mode="cube")` calls the reference makes outside pbr_shading (train.py:409-417 for the envmap TV term,
render.py:80 and relight.py:108 to export the environment map), on libgigs_hip with the sampling rule written in
include/gigs_hip.h (nvdiffrast is third-party and absent here: parity unpinned beyond that rule)."""
from __future__ import annotations

import torch

import gigs_lib

_lib = gigs_lib.lib()

# Gather plans of the lookup's backward for direction sets that come back every iteration (the envmap TV's panorama grid):
# taps sorted by texel into a CSR list, built once per (directions tensor, resolution) outside any graph capture.
# GIGS_CUBE_BWD_GATHER=0 keeps the atomic scatter.
_PLAN_MIN_DIRS = 1 << 14
_HEAVY = 64
_plans = {}


def _gather_plan(d: torch.Tensor, res: int, n: int, build: bool):
    import os
    if os.environ.get("GIGS_CUBE_BWD_GATHER", "1") != "1" or n < _PLAN_MIN_DIRS:
        return None
    key = (d.data_ptr(), d._version, res, n, str(d.device))
    hit = _plans.get(key)
    if hit is not None and hit["dirs"] is d:
        return hit
    if not build or torch.cuda.is_current_stream_capturing():
        return None
    dev = d.device
    idx = torch.empty((n, 4), dtype=torch.int32, device=dev)
    w = torch.empty((n, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        gigs_lib.check(_lib.gigs_cube_taps(res, n, d.data_ptr(), idx.data_ptr(), w.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), "cube_taps")
    flat = idx.reshape(-1)
    valid = flat >= 0
    texel = flat[valid].long()
    sample = (torch.arange(4 * n, device=dev) // 4)[valid]
    weight = w.reshape(-1)[valid]
    order = torch.argsort(texel, stable=True)  # a texel's entries stay in sample order: the sums are reproducible
    n_tex = 6 * res * res
    counts = torch.bincount(texel, minlength=n_tex)
    offsets = torch.zeros(n_tex + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(counts, 0)
    heavy_ids = torch.nonzero(counts > _HEAVY).reshape(-1).int().contiguous()
    plan = dict(dirs=d, offsets=offsets.int().contiguous(), sample=sample[order].int().contiguous(),
                weight=weight[order].contiguous(), heavy_ids=heavy_ids, n_heavy=int(heavy_ids.numel()))
    if len(_plans) >= 16:
        _plans.clear()
    _plans[key] = plan
    return plan


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
        if cubemap.requires_grad or ctx.needs_input_grad[0]:
            _gather_plan(d, ctx.res, n, build=True)  # built on the first (eager) call, found again by the backward
        return out

    @staticmethod
    def backward(ctx, g_out):
        (d,) = ctx.saved_tensors
        g = g_out.contiguous().float()
        plan = _gather_plan(d, ctx.res, ctx.n, build=False)
        if plan is not None:
            d_tex = torch.empty((6, ctx.res, ctx.res, 3), dtype=torch.float32, device=g.device)  # every texel is written
            with torch.cuda.device(g.device):
                gigs_lib.check(_lib.gigs_cube_texture_bwd_gather(
                    ctx.res, ctx.n, int(ctx.planar), plan["offsets"].data_ptr(), plan["sample"].data_ptr(),
                    plan["weight"].data_ptr(), _HEAVY, plan["n_heavy"],
                    plan["heavy_ids"].data_ptr() if plan["n_heavy"] else None, g.data_ptr(), d_tex.data_ptr(),
                    torch.cuda.current_stream().cuda_stream), "cube_texture_bwd_gather")
            return d_tex, None, None
        d_tex = torch.zeros((6, ctx.res, ctx.res, 3), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            gigs_lib.check(_lib.gigs_cube_texture_bwd(ctx.res, ctx.n, d.data_ptr(), g.data_ptr(), d_tex.data_ptr(),
                                                      int(ctx.planar), torch.cuda.current_stream().cuda_stream),
                           "cube_texture_bwd")
        return d_tex, None, None


def cube_texture(cubemap: torch.Tensor, dirs: torch.Tensor, planar: bool = False) -> torch.Tensor:
    """cubemap [6,res,res,3], dirs [...,3] -> [...,3] (or [3,...] planes); differentiable w.r.t. the cubemap."""
    return _CubeTexture.apply(cubemap, dirs, planar)
