"""View-parallel data parallelism for the rasterizer hot path.

The reference is single-GPU, one view per step (train.py:247-279).  Views are independent given
replicated Gaussian parameters, so N ranks (one process per GPU, torch.distributed over
RCCL/xGMI; `gloo` on CPU for tests) each render a different view and the parameter gradients
are summed once per step.

Collective design for MI355X: the 8 GPUs of a node are fully connected by point-to-point xGMI
links (7 x ~153 GB/s per GPU), so a ring is bound by ONE link.  All gradient tensors are packed
into a single flat fp32 bucket (268 B per Gaussian at SH degree 3 -> 0.8 GB at 3 M Gaussians)
and reduced with one call, which lets RCCL use its direct (all links busy) algorithms; many
small per-tensor all-reduces would each pay the launch + latency floor.  Gradients only become
final when the preprocess-backward kernel finishes (it writes every tensor at once), so there
is nothing to overlap inside a step.

Densification statistics must be reduced as STATISTICS, not recomputed from reduced gradients:
the norm of `means2D.grad[:, :2]` and the abs accumulator are per-view quantities that the
reference sums over views (scene/gaussian_model.py:933-945), and `max_radii2D` is a max
(train.py:495-498).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def view_for(step: int, rank: int, world: int, n_views: int) -> int:
    """Round-robin view assignment: global step `step` renders views step*world + rank."""
    return (step * world + rank) % n_views


def flatten_grads(params: Sequence[torch.Tensor]) -> torch.Tensor:
    """One contiguous fp32 bucket with every parameter's gradient (zeros where absent)."""
    parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]
    return torch.cat(parts) if parts else torch.zeros(0)


def unflatten_to_grads(flat: torch.Tensor, params: Sequence[torch.Tensor]) -> None:
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n


def allreduce_gradients(params: Sequence[torch.Tensor], group: Optional[dist.ProcessGroup] = None,
                        average: bool = False) -> torch.Tensor:
    """Sum (or average) the gradients of `params` over all ranks with ONE all-reduce."""
    flat = flatten_grads(params)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= dist.get_world_size(group)
    unflatten_to_grads(flat, params)
    return flat


def reduce_densification_stats(xyz_gradient_accum: torch.Tensor, xyz_gradient_accum_abs: torch.Tensor,
                               denom: torch.Tensor, max_radii2D: torch.Tensor,
                               group: Optional[dist.ProcessGroup] = None) -> None:
    """In-place reduction of the per-view densification statistics: sums for the gradient-norm
    accumulators and the visit counter, max for the screen-space radius."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    packed = torch.cat([xyz_gradient_accum.reshape(-1), xyz_gradient_accum_abs.reshape(-1), denom.reshape(-1)])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    n = xyz_gradient_accum.numel()
    xyz_gradient_accum.copy_(packed[:n].view_as(xyz_gradient_accum))
    xyz_gradient_accum_abs.copy_(packed[n:2 * n].view_as(xyz_gradient_accum_abs))
    denom.copy_(packed[2 * n:].view_as(denom))
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)


def per_view_densification_stats(viewspace_grad: torch.Tensor, radii: torch.Tensor) -> Dict[str, torch.Tensor]:
    """What ONE view contributes (scene/gaussian_model.py:933-945): |grad[:, :2]|, grad[:, 2:], 1 for
    visible Gaussians; radii for the running max."""
    vis = radii > 0
    z = torch.zeros((viewspace_grad.shape[0], 1), dtype=viewspace_grad.dtype, device=viewspace_grad.device)
    accum = torch.where(vis[:, None], torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True), z)
    accum_abs = torch.where(vis[:, None], viewspace_grad[:, 2:3], z)
    return dict(xyz_gradient_accum=accum, xyz_gradient_accum_abs=accum_abs, denom=vis[:, None].to(viewspace_grad.dtype),
                max_radii2D=radii.to(viewspace_grad.dtype))
