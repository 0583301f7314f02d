"""View-parallel data parallelism for the rasterizer hot path.

The reference is single-GPU, one view per step (train.py:247-279).  Views are independent given
replicated Gaussian parameters, so N ranks (one process per GPU, torch.distributed over
RCCL/xGMI; `gloo` on CPU for tests) each render a different view and the parameter gradients
are summed once per step.

Collective design for MI355X: the 8 GPUs of a node are fully connected by point-to-point xGMI
links (7 x ~153 GB/s per GPU), so a ring is bound by ONE link.  All gradients live in ONE
persistent flat fp32 slab (`GradSlab`: every parameter's `.grad` is a view into it, so the backward
kernels write straight into the bucket -- no per-step `torch.cat`, no second 0.8 GB copy at 3 M
Gaussians) and are reduced with one call, which lets RCCL use its direct (all links busy) algorithms;
many small per-tensor all-reduces would each pay the launch + latency floor.

Overlap (`GradSlab.allreduce_async` / `.wait`): the all-reduce is issued on a dedicated communication
stream behind an event recorded after the backward, and the compute stream only waits for it where the
reduced gradients are consumed (the optimizer step).  Everything the training loop does between
`loss.backward()` and `optimizer.step()` -- densification statistics, logging, the NEXT view's camera
set-up -- runs beside the collective.  The update stays exact (no one-step gradient delay): gradients
only become final when the preprocess-backward kernel finishes (it writes every tensor at once), so
there is nothing earlier inside a step to overlap with.

A parameter without a gradient (e.g. the SH rest or the light in stage 1) keeps `.grad = None` after
the reduction, exactly as on one GPU: Adam must not decay its moments or advance its step.  In an SPMD
training loop the set of such parameters is a function of the iteration number and therefore the same on
every rank (the default, `agree_presence=False`: no extra traffic); `agree_presence=True` agrees it with a
tiny presence-mask all-reduce (MAX) issued together with the bucket, for loops where it can differ.

Zero-copy packing: `GradSlab.sink()` hands the slab's views to the rasterizer's backward
(`diff_gaussian_rasterization.grad_sink`), whose kernels then write the per-Gaussian gradients straight into
the bucket.  What that saves depends on who hands the gradient to the parameter: `pipeline.WholeStepGraph`
(torch.autograd.grad inside the capture) assigns the sink tensor itself as `.grad` -- no copy at all; eager
`loss.backward()` goes through AccumulateGrad, which CLONES a gradient that something else still references (the
sink dictionary does), so `.grad` is a fresh tensor and `_gather_stray` copies it into the slab once (one copy
instead of the `torch.cat` of every gradient).  The sink is only meaningful when the rasterizer's inputs ARE the
leaf parameters: with activations in between, the view receives dL/d(activated input), not the parameter's
gradient.  `attach()` and the sink are mutually exclusive per parameter (a `.grad` that already aliases the sink
tensor would be added to itself by AccumulateGrad): `attach(skip=sink)` leaves sinked parameters at `.grad = None`.

Densification statistics must be reduced as STATISTICS, not recomputed from reduced gradients:
the norm of `means2D.grad[:, :2]` and the abs accumulator are per-view quantities that the
reference sums over views (scene/gaussian_model.py:933-945), `xyz_gradient_accum_abs_max` and
`max_radii2D` are maxima (gaussian_model.py:944, train.py:495-498).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


def view_for(step: int, rank: int, world: int, n_views: int) -> int:
    """Round-robin view assignment: global step `step` renders views step*world + rank."""
    return (step * world + rank) % n_views


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class GradSlab:
    """One persistent flat fp32 buffer holding the gradients of `params`, in order.

        slab = GradSlab(params)
        each step:  slab.attach()            # p.grad = zeroed view into the slab (autograd accumulates in place)
                    loss.backward()
                    slab.allreduce_async()   # comm stream; returns at once
                    ... statistics, logging ...
                    slab.wait()              # compute stream waits; gradient-less parameters get .grad = None
                    optimizer.step()

    `attach(zero=True)` clears the slab with one fill instead of one per tensor.  With world size 1 the
    collective is skipped entirely (no copies, no stream hops)."""

    def __init__(self, params: Sequence[torch.Tensor], group=None, agree_presence: bool = False):
        self.params: List[torch.Tensor] = list(params)
        self.group = group
        self.agree_presence = bool(agree_presence)
        if not self.params:
            raise ValueError("GradSlab needs at least one parameter")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("GradSlab: all parameters must share one device and dtype")
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += p.numel()
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        self.views = [self.flat[o:o + p.numel()].view(p.shape) for o, p in zip(self.offsets, self.params)]
        self.presence = torch.zeros(len(self.params), dtype=torch.float32, device=dev)
        self._comm = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._work: list = []
        self._pending = False
        # timing=True: three events per collective (backward done on the compute stream; collective begin / end on the
        # communication stream), read by comm_stats() after a synchronize
        self.timing = False
        self._events: list = []
        self._bytes = 0

    def rebuild(self, params: Sequence[torch.Tensor]) -> bool:
        """After densification: the reference REPLACES the nn.Parameter objects (cat_tensors_to_optimizer /
        _prune_optimizer, scene/gaussian_model.py:595-706), so the caller passes the current list.  Re-carves the slab
        when any tensor is a new object or has a new shape (returns True if it did); an all-reduce must not be pending."""
        params = list(params)
        same = len(params) == len(self.params) and all(a is b and v.shape == a.shape
                                                       for a, b, v in zip(params, self.params, self.views))
        if same:
            return False
        if self._pending:
            raise RuntimeError("GradSlab.rebuild: an all-reduce is pending (call wait() first)")
        timing = self.timing
        self.__init__(params, self.group, self.agree_presence)
        self.timing = timing
        return True

    def sink(self, names: Sequence[str]) -> Dict[str, torch.Tensor]:
        """{name: slab view} for the first len(names) parameters, for diff_gaussian_rasterization.grad_sink: the
        rasterizer's backward writes the gradient of its input `name` into that view."""
        return {n: v for n, v in zip(names, self.views)}

    def attach(self, zero: bool = True, skip: Optional[Dict[str, torch.Tensor]] = None) -> None:
        """Point every parameter's .grad at its slab view (autograd then accumulates into the slab).  `skip`: the
        dictionary handed to grad_sink -- parameters whose view is a sink target keep `.grad = None`, because the
        rasterizer's backward OVERWRITES the view and returns it, and AccumulateGrad would then add it to itself."""
        if zero:
            self.flat.zero_()
        sinked = set(t.data_ptr() for t in skip.values()) if skip else ()
        for p, v in zip(self.params, self.views):
            p.grad = None if v.data_ptr() in sinked else v

    def _gather_stray(self) -> None:
        # a gradient that is not already the slab view (autograd made a fresh tensor) is copied in once
        pres = []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                pres.append(0.0)
                v.zero_()
                continue
            pres.append(1.0)
            if g.data_ptr() != v.data_ptr() or g.stride() != v.stride():
                v.copy_(g)
                p.grad = v
        self._local_presence = pres
        if self.agree_presence:
            self.presence.copy_(torch.tensor(pres, dtype=torch.float32), non_blocking=True)

    def _ranges(self, only) -> List[torch.Tensor]:
        """The slab as the fewest contiguous pieces that cover the parameters `only` (indices into the slab's order)."""
        if only is None:
            return [self.flat]
        idx = sorted(set(int(i) for i in only))
        pieces, k = [], 0
        while k < len(idx):
            j = k
            while j + 1 < len(idx) and idx[j + 1] == idx[j] + 1:
                j += 1
            pieces.append(self.flat[self.offsets[idx[k]]:self.offsets[idx[j]] + self.params[idx[j]].numel()])
            k = j + 1
        return pieces

    def allreduce_async(self, average: bool = False, force: bool = False, only: Optional[Sequence[int]] = None,
                        check_rest_zero: bool = False, packed: bool = False) -> None:
        """Sum (or average) the slab over all ranks on the communication stream.  Parameters whose .grad is None
        here contribute zeros.  `force` issues the collective even with a single rank (a one-GPU rehearsal of the
        whole code path: process group, communication stream, RCCL launch).

        `only` (indices into the slab's parameter order): reduce just these parameters' stretches of the slab -- for
        steps whose other gradients are IDENTICALLY ZERO on every rank, e.g. a stage-2 iteration of train.py, where the
        loss reaches only albedo / roughness / metallic and the light (the blend weights are detached from the material
        planes: SURVEY App. D, tests/test_gpu_parity.py::test_backward_stage2_pattern_and_linearity); a sum of zeros is
        zero, so the result equals the full reduction with 5 of 46 floats per Gaussian on the wire.  Adjacent parameters
        share one collective (order the slab accordingly).  `check_rest_zero` verifies the premise (one device read).

        `packed`: the slab already HOLDS this step's gradients -- the backward wrote them into the views (grad_sink of the
        rasterizer / of the activations inside a captured step) -- and `.grad` of the parameters is not in play (the
        captured Adam launch reads the views): nothing is gathered before and nothing assigned after the collective."""
        world = _world(self.group)
        self._packed = bool(packed)
        if world == 1 and not (force and dist.is_available() and dist.is_initialized()):
            self._pending = False
            return
        if not packed:
            self._gather_stray()
        self._average = average
        pieces = self._ranges(only)
        if only is not None and check_rest_zero:
            keep = set(int(i) for i in only)
            rest = [v for i, v in enumerate(self.views) if i not in keep]
            if rest and float(torch.stack([v.abs().max() for v in rest if v.numel()]).max()) != 0.0:
                raise RuntimeError("GradSlab.allreduce_async(only=...): a parameter outside `only` has a non-zero gradient")
        self._pieces = pieces
        self._bytes = sum(p.numel() * p.element_size() for p in pieces)
        if self._comm is not None:
            main = torch.cuda.current_stream(self.flat.device)
            ev = None
            if self.timing:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record(main)
            self._comm.wait_stream(main)
            with torch.cuda.stream(self._comm):
                if ev:
                    ev[1].record(self._comm)
                for piece in pieces:
                    dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group)
                if self.agree_presence:
                    dist.all_reduce(self.presence, op=dist.ReduceOp.MAX, group=self.group)
                if average:
                    for piece in pieces:
                        piece.div_(world)
                if ev:
                    ev[2].record(self._comm)
                    self._events.append(ev)
        else:  # CPU tensors (gloo tests): genuinely asynchronous work handles
            self._work = [dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for piece in pieces]
            if self.agree_presence:
                self._work.append(dist.all_reduce(self.presence, op=dist.ReduceOp.MAX, group=self.group, async_op=True))
        self._pending = True

    def wait(self) -> None:
        """Make the reduced gradients visible to the compute stream and restore `.grad = None` where no rank had one."""
        if not self._pending:
            return
        if self._comm is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._comm)
        else:
            for w in self._work:
                w.wait()
            self._work = []
            if self._average:
                for piece in self._pieces:
                    piece.div_(_world(self.group))
        if not getattr(self, "_packed", False):
            # agree_presence: one small read-back per step, after the collective; otherwise the local pattern is everyone's
            pres = self.presence.tolist() if self.agree_presence else self._local_presence
            for p, v, has in zip(self.params, self.views, pres):
                p.grad = v if has > 0 else None
        self._pending = False

    def comm_stats(self, reset: bool = True) -> Optional[Dict[str, float]]:
        """Averages over the collectives issued since the last reset (timing=True; call after a synchronize):
        `allreduce_ms` = the collective itself on the communication stream, `exposed_ms` = from the end of the backward
        on the compute stream to the end of the collective -- what a step that consumes the reduced gradients right
        away (wait() directly after allreduce_async(), as bench.py does) pays for it -- and `bytes` per collective."""
        if not self._events:
            return None
        n = len(self._events)
        out = {"collectives": n, "bytes": int(self._bytes),
               "allreduce_ms": round(sum(e[1].elapsed_time(e[2]) for e in self._events) / n, 4),
               "exposed_ms": round(sum(e[0].elapsed_time(e[2]) for e in self._events) / n, 4)}
        if reset:
            self._events = []
        return out

    def allreduce(self, average: bool = False) -> torch.Tensor:
        self.allreduce_async(average)
        self.wait()
        return self.flat


def allreduce_gradients(params: Sequence[torch.Tensor], group: Optional[dist.ProcessGroup] = None,
                        average: bool = False, slab: Optional[GradSlab] = None) -> Optional[torch.Tensor]:
    """Sum (or average) the gradients of `params` over all ranks with ONE all-reduce.  World size 1: nothing to do
    (gradients stay where they are, None stays None).  Pass a persistent `slab` to avoid the per-call packing;
    without one the presence pattern is agreed across ranks (a parameter that has a gradient on any rank gets the sum)."""
    if _world(group) == 1:
        return None
    if slab is None:
        slab = GradSlab(params, group, agree_presence=True)
    return slab.allreduce(average)


def reduce_densification_stats(xyz_gradient_accum: torch.Tensor, xyz_gradient_accum_abs: torch.Tensor,
                               denom: torch.Tensor, max_radii2D: torch.Tensor,
                               xyz_gradient_accum_abs_max: Optional[torch.Tensor] = None,
                               group: Optional[dist.ProcessGroup] = None) -> None:
    """In-place reduction of the per-view densification statistics: sums for the gradient-norm
    accumulators and the visit counter, max for the screen-space radius and the abs-gradient maximum."""
    if _world(group) == 1:
        return
    packed = torch.cat([xyz_gradient_accum.reshape(-1), xyz_gradient_accum_abs.reshape(-1), denom.reshape(-1)])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    n = xyz_gradient_accum.numel()
    xyz_gradient_accum.copy_(packed[:n].view_as(xyz_gradient_accum))
    xyz_gradient_accum_abs.copy_(packed[n:2 * n].view_as(xyz_gradient_accum_abs))
    denom.copy_(packed[2 * n:].view_as(denom))
    if xyz_gradient_accum_abs_max is not None:
        mx = torch.cat([max_radii2D.reshape(-1).to(xyz_gradient_accum_abs_max.dtype), xyz_gradient_accum_abs_max.reshape(-1)])
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        m = max_radii2D.numel()
        max_radii2D.copy_(mx[:m].view_as(max_radii2D).to(max_radii2D.dtype))
        xyz_gradient_accum_abs_max.copy_(mx[m:].view_as(xyz_gradient_accum_abs_max))
    else:
        dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)


def per_view_densification_stats(viewspace_grad: torch.Tensor, radii: torch.Tensor) -> Dict[str, torch.Tensor]:
    """What ONE view contributes (scene/gaussian_model.py:933-945): ||grad[:, :2]|| to xyz_gradient_accum,
    |grad_x| + |grad_y| to xyz_gradient_accum_abs and (as a running max) to xyz_gradient_accum_abs_max, 1 to denom, for
    visible Gaussians; radii for the running max of max_radii2D (train.py:495-498).  Same arithmetic as the
    library's one-pass gigs_densify_stats (densify.add_densification_stats), which is what a training loop calls."""
    vis = radii > 0
    z = torch.zeros((viewspace_grad.shape[0], 1), dtype=viewspace_grad.dtype, device=viewspace_grad.device)
    accum = torch.where(vis[:, None], torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True), z)
    abs_xy = viewspace_grad[:, 0:1].abs() + viewspace_grad[:, 1:2].abs()
    accum_abs = torch.where(vis[:, None], abs_xy, z)
    return dict(xyz_gradient_accum=accum, xyz_gradient_accum_abs=accum_abs, xyz_gradient_accum_abs_max=accum_abs.clone(),
                denom=vis[:, None].to(viewspace_grad.dtype), max_radii2D=radii.to(viewspace_grad.dtype))
