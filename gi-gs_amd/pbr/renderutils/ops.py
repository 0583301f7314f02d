"""diffuse_cubemap / specular_cubemap with the reference's signatures (pbr/renderutils/ops.py:404-458),
backed by the HIP kernels of libgigs_hip.so instead of the JIT-compiled CUDA plugin."""
from __future__ import annotations

import os

import numpy as np
import torch

import gigs_lib

_lib = gigs_lib.lib()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _gpu(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor: pbr.renderutils (gigs-hip) has no CPU path")
    return t.contiguous().float()


class _diffuse_cubemap_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cubemap):
        c = _gpu(cubemap, "cubemap")
        out = torch.empty_like(c)
        with torch.cuda.device(c.device):
            gigs_lib.check(_lib.gigs_diffuse_cubemap_fwd(c.shape[1], c.data_ptr(), out.data_ptr(), _stream()),
                           "diffuse_cubemap_fwd")
        ctx.res = c.shape[1]
        return out

    @staticmethod
    def backward(ctx, dout):
        d = _gpu(dout, "dout")
        g = torch.empty_like(d)
        with torch.cuda.device(d.device):
            gigs_lib.check(_lib.gigs_diffuse_cubemap_bwd(ctx.res, d.data_ptr(), g.data_ptr(), _stream()),
                           "diffuse_cubemap_bwd")
        return g


def diffuse_cubemap(cubemap, use_python=False):
    if use_python:
        assert False
    assert cubemap.shape[0] == 6 and cubemap.shape[1] == cubemap.shape[2] and cubemap.shape[3] == 3
    out = _diffuse_cubemap_func.apply(cubemap)
    if torch.is_anomaly_enabled():
        assert not torch.isnan(out).any(), "Output of diffuse_cubemap contains inf or NaN"
    return out


class _specular_cubemap(torch.autograd.Function):
    """`tables` = None (recompute the pair weights every call, like the reference) or the cached
    (offsets, weights_fwd, weights_bwd) of `_weight_tables`."""

    @staticmethod
    def forward(ctx, cubemap, roughness, costheta_cutoff, bounds, tables=None):
        c = _gpu(cubemap, "cubemap")
        res = c.shape[1]
        out = torch.empty((6, res, res, 4), dtype=torch.float32, device=c.device)
        with torch.cuda.device(c.device):
            if tables is None:
                gigs_lib.check(_lib.gigs_specular_cubemap_fwd(res, c.data_ptr(), bounds.data_ptr(), float(roughness),
                                                              float(costheta_cutoff), out.data_ptr(), _stream()),
                               "specular_cubemap_fwd")
            else:
                gigs_lib.check(_lib.gigs_specular_cubemap_fwd_w(gigs_lib.ctx_ptr(), res, c.data_ptr(), bounds.data_ptr(), tables[0].data_ptr(),
                                                                tables[1].data_ptr(), _avg_window(tables, res),
                                                                out.data_ptr(), None, _stream()),
                               "specular_cubemap_fwd_w")
        ctx.save_for_backward(bounds)
        ctx.tables = tables
        ctx.lib_ctx = gigs_lib.current()  # for the backward, which autograd runs on its own thread
        ctx.res, ctx.roughness, ctx.theta_cutoff = res, roughness, costheta_cutoff
        return out

    @staticmethod
    @gigs_lib.with_forward_context
    def backward(ctx, dout):
        (bounds,) = ctx.saved_tensors
        d = _gpu(dout, "dout")
        g = torch.empty((6, ctx.res, ctx.res, 3), dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            if ctx.tables is None:
                gigs_lib.check(_lib.gigs_specular_cubemap_bwd(ctx.res, bounds.data_ptr(), d.data_ptr(), float(ctx.roughness),
                                                              float(ctx.theta_cutoff), g.data_ptr(), _stream()),
                               "specular_cubemap_bwd")
            else:
                gigs_lib.check(_lib.gigs_specular_cubemap_bwd_w(gigs_lib.ctx_ptr(), ctx.res, bounds.data_ptr(), ctx.tables[0].data_ptr(),
                                                                ctx.tables[2].data_ptr(), _avg_window(ctx.tables, ctx.res),
                                                                d.data_ptr(), 0, g.data_ptr(), _stream()),
                               "specular_cubemap_bwd_w")
        return g, None, None, None, None


def _avg_window(tables, res: int) -> int:
    """Mean number of window candidates per texel (scheduling hint of the *_w entry points)."""
    return max(1, tables[1].numel() // (6 * res * res))


def _ndf_cutoff(roughness: float, cutoff: float) -> float:
    """Cosine of the GGX cone that keeps `cutoff` of the energy (ops.py:428-440, host numpy)."""
    def ndfGGX(alphaSqr, costheta):
        costheta = np.clip(costheta, 0.0, 1.0)
        d = (costheta * alphaSqr - costheta) * costheta + 1.0
        return alphaSqr / (d * d * np.pi)

    nSamples = 1000000
    costheta = np.cos(np.linspace(0, np.pi / 2.0, nSamples))
    D = np.cumsum(ndfGGX(roughness ** 4, costheta))
    idx = np.argmax(D >= D[..., -1] * cutoff)
    return float(costheta[idx])


_ndfBoundsDict = {}


def _ndf_bounds(res, roughness, cutoff, device):
    key = (res, roughness, cutoff, str(device))
    if key not in _ndfBoundsDict:
        cos_cut = _ndf_cutoff(roughness, cutoff)
        bounds = torch.zeros((6, res, res, 24), dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            gigs_lib.check(_lib.gigs_specular_bounds(res, cos_cut, bounds.data_ptr(), _stream()), "specular_bounds")
        _ndfBoundsDict[key] = (cos_cut, bounds)
    return _ndfBoundsDict[key]


class _specular_cubemap_normalized(torch.autograd.Function):
    """rgb / wsum in ONE launch (table path): forward = streaming filter + division, backward =
    (g / wsum) then the gather; replaces kernel + slice + div and their ~10 autograd launches."""

    @staticmethod
    def forward(ctx, cubemap, bounds, tables):
        c = _gpu(cubemap, "cubemap")
        res = c.shape[1]
        out = torch.empty((6, res, res, 3), dtype=torch.float32, device=c.device)
        wsum = torch.empty((6, res, res, 1), dtype=torch.float32, device=c.device)
        with torch.cuda.device(c.device):
            gigs_lib.check(_lib.gigs_specular_cubemap_fwd_w(gigs_lib.ctx_ptr(), res, c.data_ptr(), bounds.data_ptr(), tables[0].data_ptr(),
                                                            tables[1].data_ptr(), _avg_window(tables, res), out.data_ptr(),
                                                            wsum.data_ptr(), _stream()),
                           "specular_cubemap_fwd_w")
        ctx.save_for_backward(bounds, wsum)
        ctx.tables, ctx.res = tables, res
        ctx.lib_ctx = gigs_lib.current()
        return out

    @staticmethod
    @gigs_lib.with_forward_context
    def backward(ctx, dout):
        bounds, wsum = ctx.saved_tensors
        if len(ctx.tables) > 3 and ctx.tables[3] is not None:
            # the 1 / wsum of d(rgb / w) / d(rgb) is folded into the cached table (gigs_specular_weights_divide)
            d, table = _gpu(dout, "dout"), ctx.tables[3]
        else:
            d, table = (_gpu(dout, "dout") / wsum).contiguous(), ctx.tables[2]  # w does not depend on the cubemap
        g = torch.empty((6, ctx.res, ctx.res, 3), dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            gigs_lib.check(_lib.gigs_specular_cubemap_bwd_w(gigs_lib.ctx_ptr(), ctx.res, bounds.data_ptr(), ctx.tables[0].data_ptr(),
                                                            table.data_ptr(), _avg_window(ctx.tables, ctx.res),
                                                            d.data_ptr(), 1, g.data_ptr(), _stream()),
                           "specular_cubemap_bwd_w")
        return g, None, None


bwd_head_start_ns = 0


class _specular_levels(torch.autograd.Function):
    """specular_cubemap of every level of a light in ONE launch each way (gigs_specular_cubemap_multi_w).  Inputs:
    the per-level (bounds, tables) as a Python list, then the level mips; needs the cached tables incl. the pre-divided
    backward table of every level (otherwise CubemapLight.build_mips falls back to one call per level)."""

    @staticmethod
    def forward(ctx, meta, *mips):
        import ctypes as C
        cs = [_gpu(m, "cubemap") for m in mips]
        dev = cs[0].device
        outs, wsums, arr = [], [], (gigs_lib.SpecLevel * len(cs))()
        for i, (c, (bounds, tables)) in enumerate(zip(cs, meta)):
            res = c.shape[1]
            out = torch.empty((6, res, res, 3), dtype=torch.float32, device=dev)
            wsum = torch.empty((6, res, res, 1), dtype=torch.float32, device=dev)
            outs.append(out)
            wsums.append(wsum)
            arr[i] = gigs_lib.SpecLevel(res, _avg_window(tables, res), c.data_ptr(), bounds.data_ptr(), tables[0].data_ptr(),
                                        tables[1].data_ptr(), out.data_ptr(), wsum.data_ptr())
        with torch.cuda.device(dev):
            gigs_lib.check(_lib.gigs_specular_cubemap_multi_w(gigs_lib.ctx_ptr(), len(cs), C.cast(arr, C.c_void_p), 0, _stream()),
                           "specular_cubemap_multi_w")
        ctx.meta = meta
        ctx.lib_ctx = gigs_lib.current()
        ctx.shapes = [c.shape[1] for c in cs]
        return tuple(outs)

    @staticmethod
    @gigs_lib.with_forward_context
    def backward(ctx, *douts):
        import ctypes as C
        dev = next(d for d in douts if d is not None).device
        gs, arr, keep = [], (gigs_lib.SpecLevel * len(douts))(), []
        for i, (d, res, (bounds, tables)) in enumerate(zip(douts, ctx.shapes, ctx.meta)):
            d = torch.zeros((6, res, res, 3), dtype=torch.float32, device=dev) if d is None else _gpu(d, "dout")
            keep.append(d)
            g = torch.empty((6, res, res, 3), dtype=torch.float32, device=dev)
            gs.append(g)
            arr[i] = gigs_lib.SpecLevel(res, _avg_window(tables, res), d.data_ptr(), bounds.data_ptr(), tables[0].data_ptr(),
                                        tables[3].data_ptr(), g.data_ptr(), None)
        with torch.cuda.device(dev):
            if bwd_head_start_ns > 0:  # see gigs_stream_delay: set by pipeline.WholeStepGraph while it captures the backward
                gigs_lib.check(_lib.gigs_stream_delay(int(bwd_head_start_ns), _stream()), "stream_delay")
            gigs_lib.check(_lib.gigs_specular_cubemap_multi_w(gigs_lib.ctx_ptr(), len(douts), C.cast(arr, C.c_void_p), 1, _stream()),
                           "specular_cubemap_multi_w")
        return (None, *gs)


def specular_cubemap_levels(mips, roughnesses, cutoff=0.99):
    """[specular_cubemap(m, r, cutoff) for m, r in zip(mips, roughnesses)] as one launch each way, or None if a level
    lacks its cached tables (then the caller filters level by level)."""
    if os.environ.get("GIGS_SPEC_MULTI", "1") != "1" or torch.is_anomaly_enabled() or not mips[0].is_cuda:
        return None
    meta = []
    for m, r in zip(mips, roughnesses):
        _, bounds = _ndf_bounds(m.shape[1], r, cutoff, m.device)
        tables = _weight_tables(m.shape[1], r, cutoff, m.device)
        if tables is None or len(tables) < 4 or tables[3] is None:
            return None
        meta.append((bounds, tables))
    return list(_specular_levels.apply(meta, *mips))


# Cached pair-weight tables (see csrc/pbr.hip): 0.74 GB for the 256..16 chain, x2 for the backward.
# HBM is 288 GB on MI355X; set GIGS_SPEC_TABLE_MAX_GB=0 to recompute the weights every call instead.
_TABLE_MAX_BYTES = float(os.environ.get("GIGS_SPEC_TABLE_MAX_GB", "8")) * 1e9
_weightTables = {}


def _weight_tables(res, roughness, cutoff, device):
    key = (res, roughness, cutoff, str(device))
    if key not in _weightTables:
        cos_cut, bounds = _ndf_bounds(res, roughness, cutoff, device)
        b = bounds.view(-1, 6, 4)
        w = torch.where(b[..., 0] <= b[..., 1], b[..., 1] - b[..., 0] + 1, torch.zeros_like(b[..., 0]))
        h = torch.where(b[..., 2] <= b[..., 3], b[..., 3] - b[..., 2] + 1, torch.zeros_like(b[..., 0]))
        cnt = (w * h).to(torch.int64).reshape(-1)
        total = int(cnt.sum().item())
        used = sum(t[1].numel() * (12 if t[3] is not None else 8) for t in _weightTables.values() if t is not None)
        if total == 0 or total >= 2 ** 31 or used + total * 8 > _TABLE_MAX_BYTES:
            _weightTables[key] = None
        else:
            offsets = (torch.cumsum(cnt, 0) - cnt).to(torch.int32).contiguous()
            tabs = []
            with torch.cuda.device(device):
                for swap in (0, 1):
                    wt = torch.empty(total, dtype=torch.float32, device=device)
                    gigs_lib.check(_lib.gigs_specular_weights(res, bounds.data_ptr(), offsets.data_ptr(), float(roughness),
                                                              float(cos_cut), swap, wt.data_ptr(), _stream()),
                                   "specular_weights")
                    tabs.append(wt)
            scaled = None
            if os.environ.get("GIGS_SPEC_PRESCALED", "1") == "1" and used + total * 12 <= _TABLE_MAX_BYTES:
                # weight sums of the forward (independent of the cubemap's values), then the backward table divided by them
                with torch.cuda.device(device):
                    ones = torch.ones((6, res, res, 3), dtype=torch.float32, device=device)
                    tmp = torch.empty_like(ones)
                    wsum = torch.empty((6, res, res), dtype=torch.float32, device=device)
                    gigs_lib.check(_lib.gigs_specular_cubemap_fwd_w(gigs_lib.ctx_ptr(), res, ones.data_ptr(), bounds.data_ptr(),
                                                                    offsets.data_ptr(), tabs[0].data_ptr(), 0,
                                                                    tmp.data_ptr(), wsum.data_ptr(), _stream()),
                                   "specular_cubemap_fwd_w")
                    scaled = torch.empty(total, dtype=torch.float32, device=device)
                    gigs_lib.check(_lib.gigs_specular_weights_divide(res, bounds.data_ptr(), offsets.data_ptr(),
                                                                     tabs[1].data_ptr(), wsum.data_ptr(), scaled.data_ptr(),
                                                                     _stream()), "specular_weights_divide")
            _weightTables[key] = (offsets, tabs[0], tabs[1], scaled)
    return _weightTables[key]


def specular_cubemap(cubemap, roughness, cutoff=0.99, use_python=False):
    assert cubemap.shape[0] == 6 and cubemap.shape[1] == cubemap.shape[2], \
        "Bad shape for cubemap tensor: %s" % str(cubemap.shape)
    if use_python:
        assert False
    if not cubemap.is_cuda:
        raise RuntimeError("cubemap must be a CUDA/HIP tensor: pbr.renderutils (gigs-hip) has no CPU path")
    cos_cut, bounds = _ndf_bounds(cubemap.shape[1], roughness, cutoff, cubemap.device)
    tables = _weight_tables(cubemap.shape[1], roughness, cutoff, cubemap.device)
    if tables is not None and not torch.is_anomaly_enabled():
        return _specular_cubemap_normalized.apply(cubemap, bounds, tables)
    out = _specular_cubemap.apply(cubemap, roughness, cos_cut, bounds, tables)
    if torch.is_anomaly_enabled():
        assert not torch.isnan(out).any(), "Output of specular_cubemap contains inf or NaN"
    return out[..., 0:3] / out[..., 3:]
