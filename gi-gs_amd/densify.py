"""Densification of the Gaussian set, restructured for the GPU.

The reference interleaves decisions and data movement: clone -> cat every tensor -> split -> cat -> prune -> prune,
each a boolean-mask index or `torch.cat` over ten parameters and twenty optimizer moments
(scene/gaussian_model.py:595-931).  Here one *plan* is computed first on [P]-sized vectors (which source row every
surviving / new Gaussian copies, which rows are new), then ONE launch of gigs_gather_rows rebuilds all thirty tensors,
and the few rows whose position / scale are re-sampled are patched.  The result is the reference's, row for row:

    rows = [originals that were not split, clones, split children (N copies, source order repeated)] minus the final
           opacity / world-size prune

`add_densification_stats` (every iteration until densify_until_iter) is one kernel, gigs_densify_stats.
Random draws come from the torch.Generator the caller passes: view-parallel ranks that seed it identically make
identical decisions and identical samples, which keeps the replicated parameters consistent without a broadcast.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

import gigs_lib

_lib = gigs_lib.lib()

NAMES = ["xyz", "f_dc", "f_rest", "opacity", "normal", "albedo", "roughness", "metallic", "scaling", "rotation"]


class DensifyState:
    """xyz_gradient_accum, xyz_gradient_accum_abs, xyz_gradient_accum_abs_max, denom [P,1] and max_radii2D [P]
    (scene/gaussian_model.py:320-323, 60)."""

    def __init__(self, P: int, device):
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)  # noqa: E731
        self.xyz_gradient_accum, self.xyz_gradient_accum_abs = z(P, 1), z(P, 1)
        self.xyz_gradient_accum_abs_max, self.denom, self.max_radii2D = z(P, 1), z(P, 1), z(P)

    def tensors(self):
        return (self.xyz_gradient_accum, self.xyz_gradient_accum_abs, self.xyz_gradient_accum_abs_max, self.denom,
                self.max_radii2D)


def add_densification_stats(state: DensifyState, viewspace_grad: torch.Tensor, radii: torch.Tensor) -> None:
    """train.py:494-498 + scene/gaussian_model.py:933-945 with update_filter = radii > 0."""
    if not viewspace_grad.is_cuda:
        raise RuntimeError("add_densification_stats needs CUDA/HIP tensors: gigs-hip has no CPU path")
    P = int(radii.shape[0])
    g = viewspace_grad.contiguous().float()
    r = radii.contiguous().to(torch.int32)
    if g.shape != (P, 3) or state.denom.shape[0] != P:
        raise ValueError("add_densification_stats: shapes differ")
    a, b, c, d, m = state.tensors()
    with torch.cuda.device(g.device):
        gigs_lib.check(_lib.gigs_densify_stats(P, g.data_ptr(), r.data_ptr(), a.data_ptr(), b.data_ptr(), c.data_ptr(),
                                               d.data_ptr(), m.data_ptr(), torch.cuda.current_stream().cuda_stream),
                       "densify_stats")


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    """utils/general_utils.py:89-110."""
    q = r / torch.sqrt((r * r).sum(dim=1))[:, None]
    w, x, y, z = q.unbind(dim=1)
    return torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)), dim=1).reshape(-1, 3, 3)


def _groups(optimizer) -> Dict[str, dict]:
    by_name = {g["name"]: g for g in optimizer.param_groups if "name" in g}
    missing = [n for n in NAMES if n not in by_name]
    if missing:
        raise ValueError(f"densify: optimizer has no parameter group named {missing}")
    for n in NAMES:
        if len(by_name[n]["params"]) != 1:
            raise ValueError("densify: every group must hold exactly one tensor (scene/gaussian_model.py:640)")
    return by_name


def _rebuild(optimizer, groups, src_index, is_new, n_in):
    """One gather launch over the ten parameters and their moments; returns {name: new Parameter}."""
    dev = src_index.device
    n_out = int(src_index.shape[0])
    zero_row = is_new.to(torch.uint8).contiguous()
    idx = src_index.to(torch.int32).contiguous()
    table, keep_alive, new_params, new_states = [], [], {}, {}
    for n in NAMES:
        p = groups[n]["params"][0]
        if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.shape[0] != n_in:
            raise RuntimeError("densify: parameters must be contiguous fp32 CUDA/HIP tensors of equal length")
        rf = 1
        for d in p.shape[1:]:
            rf *= int(d)
        shape = (n_out,) + tuple(p.shape[1:])
        dst = torch.empty(shape, dtype=torch.float32, device=dev)
        table.append(gigs_lib.GatherTensor(p.data_ptr(), dst.data_ptr(), rf, 0))
        new_params[n] = dst
        st = optimizer.state.get(p, None)
        if st is not None and len(st) > 0:
            ns = dict(st)
            for k in ("exp_avg", "exp_avg_sq"):
                m = st[k].contiguous()
                d2 = torch.empty(shape, dtype=torch.float32, device=dev)
                table.append(gigs_lib.GatherTensor(m.data_ptr(), d2.data_ptr(), rf, 1))
                keep_alive.append(m)
                ns[k] = d2
            new_states[n] = ns
    arr = (gigs_lib.GatherTensor * len(table))(*table)
    with torch.cuda.device(dev):
        gigs_lib.check(_lib.gigs_gather_rows(len(table), C.cast(arr, C.c_void_p), n_out, n_in, idx.data_ptr(),
                                             zero_row.data_ptr(), torch.cuda.current_stream().cuda_stream), "gather_rows")
    return new_params, new_states


def _install(optimizer, groups, new_params, new_states):
    out = {}
    for n in NAMES:
        old = groups[n]["params"][0]
        if old in optimizer.state:
            del optimizer.state[old]
        param = torch.nn.Parameter(new_params[n].requires_grad_(True))
        groups[n]["params"][0] = param
        if n in new_states:
            optimizer.state[param] = new_states[n]
        out[n] = param
    return out


@torch.no_grad()
def densify_and_prune(optimizer, state: DensifyState, max_grad: float, min_opacity: float, extent: float,
                      max_screen_size: Optional[int], percent_dense: float = 0.01, N: int = 2,
                      generator: Optional[torch.Generator] = None, _noise=None):
    """GaussianModel.densify_and_prune (scene/gaussian_model.py:907-931) -> ({name: new Parameter}, new DensifyState).

    `optimizer` holds the ten single-tensor groups by name (any torch optimizer with exp_avg / exp_avg_sq moments).
    `_noise(kind, n)` (tests only) supplies the standard-normal draws instead of `generator`."""
    groups = _groups(optimizer)
    p = {n: groups[n]["params"][0].detach() for n in NAMES}
    dev = p["xyz"].device
    if not p["xyz"].is_cuda:
        raise RuntimeError("densify_and_prune needs CUDA/HIP tensors: gigs-hip has no CPU path")
    P0 = int(p["xyz"].shape[0])
    grads = state.xyz_gradient_accum / state.denom
    grads[grads.isnan()] = 0.0
    grads_abs = state.xyz_gradient_accum_abs / state.denom
    grads_abs[grads_abs.isnan()] = 0.0
    g, ga = grads.abs().squeeze(-1), grads_abs.abs().squeeze(-1)  # torch.norm over the size-1 last dim
    ratio = (g >= max_grad).float().mean()
    Q = torch.quantile(grads_abs.reshape(-1), 1 - ratio)
    scale = torch.exp(p["scaling"])
    big = scale.max(dim=1).values > percent_dense * extent
    clone = ((g >= max_grad) | (ga >= Q)) & ~big
    # the split test uses the zero-padded, un-normed gradients (:712-718)
    split = ((grads.squeeze(-1) >= max_grad) | (grads_abs.squeeze(-1) >= Q)) & big
    ar = torch.arange(P0, device=dev)
    i_keep, i_clone, i_split = ar[~split], ar[clone], ar[split]
    nk, nc, ns = int(i_keep.shape[0]), int(i_clone.shape[0]), int(i_split.shape[0])
    src = torch.cat((i_keep, i_clone, i_split.repeat(N)))
    is_new = torch.cat((torch.zeros(nk, dtype=torch.bool, device=dev), torch.ones(nc + N * ns, dtype=torch.bool, device=dev)))

    def draw(kind, n):
        if _noise is not None:
            return _noise(kind, n).to(dev)
        return torch.randn((n, 3), device=dev, generator=generator)

    # positions / scales of the new rows (clone samples first, then split samples: the reference's order of draws)
    new_xyz = torch.empty((nc + N * ns, 3), device=dev)
    new_scaling = torch.empty((nc + N * ns, 3), device=dev)
    if nc:
        smp = scale[i_clone] * draw("clone", nc)
        new_xyz[:nc] = torch.bmm(build_rotation(p["rotation"][i_clone]), smp.unsqueeze(-1)).squeeze(-1) + p["xyz"][i_clone]
        new_scaling[:nc] = p["scaling"][i_clone]
    if ns:
        rep = i_split.repeat(N)
        smp = scale[rep] * draw("split", N * ns)
        new_xyz[nc:] = torch.bmm(build_rotation(p["rotation"][rep]), smp.unsqueeze(-1)).squeeze(-1) + p["xyz"][rep]
        new_scaling[nc:] = torch.log(scale[rep] / (0.8 * N))
    # final prune (:919-927) decided on the planned rows; max_radii2D was reset by the reference's postfix, so only
    # opacity and world size can fire
    scaling_rows = torch.cat((p["scaling"][i_keep], new_scaling))
    prune = (torch.sigmoid(p["opacity"][src]) < min_opacity).squeeze(-1)
    if max_screen_size:
        prune = prune | (torch.exp(scaling_rows).max(dim=1).values > 0.1 * extent)
    keep = ~prune
    src_f, new_f = src[keep], is_new[keep]
    new_params, new_states = _rebuild(optimizer, groups, src_f, new_f, P0)
    # patch the re-sampled attributes of the surviving new rows
    keep_new = keep[nk:]
    new_params["xyz"][new_f] = new_xyz[keep_new]
    new_params["scaling"][new_f] = new_scaling[keep_new]
    out = _install(optimizer, groups, new_params, new_states)
    return out, DensifyState(int(src_f.shape[0]), dev)


@torch.no_grad()
def prune_points(optimizer, state: DensifyState, mask: torch.Tensor):
    """GaussianModel.prune_points (:613-633): drops the rows where mask is True, statistics included."""
    groups = _groups(optimizer)
    P0 = int(groups["xyz"]["params"][0].shape[0])
    src = torch.arange(P0, device=mask.device)[~mask]
    new_params, new_states = _rebuild(optimizer, groups, src, torch.zeros_like(src, dtype=torch.bool), P0)
    out = _install(optimizer, groups, new_params, new_states)
    ns = DensifyState(0, mask.device)
    (ns.xyz_gradient_accum, ns.xyz_gradient_accum_abs, ns.xyz_gradient_accum_abs_max, ns.denom, ns.max_radii2D) = (
        t[src] for t in state.tensors())
    return out, ns


@torch.no_grad()
def reset_opacity(optimizer):
    """GaussianModel.reset_opacity (:467-472) with replace_tensor_to_optimizer (:580-593)."""
    groups = _groups(optimizer)
    old = groups["opacity"]["params"][0]
    o = torch.sigmoid(old.detach())
    o = torch.min(o, torch.ones_like(o) * 0.01)
    new = torch.nn.Parameter(torch.log(o / (1 - o)).requires_grad_(True))
    st = optimizer.state.get(old, None)
    if st is not None and len(st) > 0:
        st = dict(st)
        st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(new), torch.zeros_like(new)
        del optimizer.state[old]
        optimizer.state[new] = st
    groups["opacity"]["params"][0] = new
    return new
