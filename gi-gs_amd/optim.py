"""Adam over all parameter groups in one launch.

The reference builds `torch.optim.Adam(l, lr=0.0, eps=1e-15)` over ten single-tensor groups (xyz, f_dc, f_rest,
opacity, normal, albedo, roughness, metallic, scaling, rotation; scene/gaussian_model.py:325-346) and steps it
once per iteration (train.py:518-520): ten or more small launches per step plus ten memsets for
`zero_grad(set_to_none=False)`.  `FusedAdam` keeps torch's optimizer interface and state layout
(`state[p] = {"step", "exp_avg", "exp_avg_sq"}`, so the reference's densification code that cats / prunes these
tensors in place, scene/gaussian_model.py:628-706, keeps working) and runs gigs_adam_step over every group at once.
"""
from __future__ import annotations

import ctypes as C

from typing import Optional

import torch

import gigs_lib

_lib = gigs_lib.lib()


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) without weight decay / amsgrad, fp32 parameters on one GPU.

    The param-group dictionaries carry every key torch.optim.Adam's groups carry (at their inert defaults), so a
    `state_dict()` written here -- e.g. inside a `chkpntN.pth` by scene_io.capture -- loads into the reference's
    `torch.optim.Adam` (GaussianModel.restore, scene/gaussian_model.py:151-176) and steps there."""

    # torch.optim.Adam's remaining group keys and the only values this optimizer implements
    _INERT = dict(weight_decay=0.0, amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                  fused=None, decoupled_weight_decay=False)

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, **kw):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("FusedAdam: invalid lr / eps / betas")
        for k, v in kw.items():
            if k not in self._INERT:
                raise TypeError(f"FusedAdam: unexpected argument {k!r}")
            self._check_inert(k, v)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, **self._INERT))

    @classmethod
    def _check_inert(cls, key, value):
        if key in ("weight_decay",) and float(value or 0.0) != 0.0:
            raise ValueError("FusedAdam implements no weight decay (the reference uses none)")
        if key in ("amsgrad", "maximize", "capturable", "differentiable", "decoupled_weight_decay") and bool(value):
            raise ValueError(f"FusedAdam does not implement {key}=True")

    def load_state_dict(self, state_dict):
        for g in state_dict.get("param_groups", []):
            for k in self._INERT:
                if k in g:
                    self._check_inert(k, g[k])
        super().load_state_dict(state_dict)
        for g in self.param_groups:  # a dict saved by an older FusedAdam lacks the inert keys
            for k, v in self._INERT.items():
                g.setdefault(k, v)

    @torch.no_grad()
    def step(self, closure=None, zero_grad: bool = False):
        """One update; `zero_grad=True` also clears the gradients in the same pass (train.py:520)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # one launch per distinct (betas, eps); the reference has a single setting
        buckets = {}
        device = None
        for group in self.param_groups:
            key = (float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]))
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous fp32 CUDA/HIP tensors "
                                       "(gigs-hip has no CPU path)")
                if p.grad.is_sparse or not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                    raise RuntimeError("FusedAdam: gradients must be dense contiguous fp32")
                if device is None:
                    device = p.device
                elif p.device != device:
                    raise RuntimeError("FusedAdam: all parameters must live on one device")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if m.shape != p.shape or v.shape != p.shape or not m.is_contiguous() or not v.is_contiguous():
                    raise RuntimeError("FusedAdam: optimizer state does not match its parameter")
                buckets.setdefault(key, []).append(
                    gigs_lib.AdamGroup(p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                       float(group["lr"]), int(st["step"])))
        if device is None:
            return loss
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream().cuda_stream
            for (b1, b2, eps), groups in buckets.items():
                arr = (gigs_lib.AdamGroup * len(groups))(*groups)
                gigs_lib.check(_lib.gigs_adam_step(len(groups), C.cast(arr, C.c_void_p), b1, b2, eps,
                                                   int(bool(zero_grad)), stream), "adam_step")
        return loss


class CapturedAdam:
    """FusedAdam's update as a hipGraph-capturable launch (pipeline.WholeStepGraph captures it behind the backward).

    A captured kernel's by-value arguments are frozen, but Adam's step size changes every iteration (bias corrections,
    learning-rate schedules).  The launch captured here reads the two per-step scalars of every group -- lr / (1 - b1^t)
    and sqrt(1 - b2^t) -- from a small DEVICE table (gigs_adam_step_dyn); `advance()` -- called by the host before each
    replay -- counts the step in the optimizer's state (so state_dict() / the reference's densification code see what
    torch.optim.Adam would have written), evaluates the scalars exactly as the eager launch does (gigs_adam_scalars:
    double arithmetic, torch's order) for the CURRENT param_group["lr"], and queues one pinned-memory copy of the table.

        cap = CapturedAdam([gaussian_optimizer, light_optimizer], params, grads)   # grads: the graph's static buffers
        with torch.cuda.graph(g): cap.launch()                                      # inside the capture
        each step: cap.advance(); g.replay()

    Every listed parameter must have a gradient buffer (a parameter without one is simply not listed: its moments and
    step count stay untouched, as torch.optim.Adam leaves a parameter whose .grad is None)."""

    def __init__(self, optimizers, params, grads, absent_is_zero: bool = False):
        """absent_is_zero: a listed parameter whose gradient is None is NOT skipped but updated with g = 0 (gigs_adam_group
        with grad == NULL) -- the declared stage-2 gradient set of pipeline.WholeStepGraph, where None stands for an exact
        zero that nobody materialised (the reference hands Adam a tensor of zeros there: the moments decay, the parameter
        follows its momentum)."""
        self.entries = []  # (optimizer, group, parameter, gradient buffer)
        by_id = {id(p): g for p, g in zip(params, grads) if g is not None or absent_is_zero}
        listed = {id(p) for p in params}
        self.buckets = {}
        dev = None
        for opt in optimizers:
            if not isinstance(opt, FusedAdam):
                raise TypeError("CapturedAdam needs FusedAdam optimizers")
            for group in opt.param_groups:
                key = (float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]))
                for p in group["params"]:
                    gr = by_id.get(id(p))
                    if gr is None and not (absent_is_zero and id(p) in listed):
                        continue
                    if (not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or (gr is not None and (
                            gr.dtype != torch.float32 or not gr.is_contiguous() or gr.shape != p.shape))):
                        raise RuntimeError("CapturedAdam: parameters and gradient buffers must be contiguous fp32 on the GPU")
                    dev = p.device if dev is None else dev
                    st = opt.state[p]
                    if len(st) == 0:
                        st["step"] = torch.tensor(0.0)
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    self.buckets.setdefault(key, []).append(len(self.entries))
                    self.entries.append((opt, group, p, gr))
        if not self.entries:
            raise ValueError("CapturedAdam: no parameter with a gradient buffer")
        self.device = dev
        n = len(self.entries)
        # table rows follow the launch order: bucket by bucket
        self.order = [i for idx in self.buckets.values() for i in idx]
        self.table = torch.zeros(n, 2, dtype=torch.float32, device=dev)
        # two pinned staging buffers, used in turn: the host may run ahead of the device, and a buffer is rewritten only once
        # the copy that read it has completed (its event)
        self.hosts = [torch.zeros(n, 2, dtype=torch.float32).pin_memory() for _ in range(2)]
        self._copied = [None, None]
        self._turn = 0
        self._sc = (C.c_float * 2)()

    def warmup(self) -> None:
        CapturedAdam.warmup_device(self.device)

    @staticmethod
    def warmup_device(device) -> None:
        """One launch of the kernel on a scratch group, OUTSIDE any capture (the first launch of a kernel loads its code
        object, which must not happen while a stream is capturing)."""
        t = [torch.zeros(16, dtype=torch.float32, device=device) for _ in range(4)]
        tab = torch.ones(2, dtype=torch.float32, device=device)
        grp = (gigs_lib.AdamGroup * 1)(gigs_lib.AdamGroup(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                                                          16, 0.0, 0))
        with torch.cuda.device(device):
            gigs_lib.check(_lib.gigs_adam_step_dyn(1, C.cast(grp, C.c_void_p), 0.9, 0.999, 1e-8, 0, tab.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream), "adam_step_dyn")
            torch.cuda.current_stream().synchronize()

    def key(self):
        """What a capture depends on: the tensors' addresses (a replaced parameter / moment needs a re-capture)."""
        out = []
        for o, _, p, gr in self.entries:
            st = o.state.get(p)
            if not st:  # the optimizer no longer holds this tensor (densification replaced it): the capture is stale
                return ("stale", id(self))
            out.append((p.data_ptr(), 0 if gr is None else gr.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()))
        return tuple(out)

    def launch(self, watch=None, changed: Optional[torch.Tensor] = None, guard: Optional[torch.Tensor] = None) -> None:
        """`watch` (group names) + `changed` (device int32[1]): gigs_adam_step_watch ORs 1 into `changed` when the update
        moves a bit of a watched group (pipeline.GeometryCache: the geometry groups of a stage-2 iteration).  `guard` (device
        int32[1]): while it is non-zero the launch changes nothing (gigs_adam_step_guarded: the violation counter of a
        declared gradient set)."""
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            row = 0
            for (b1, b2, eps), idx in self.buckets.items():
                groups, flags = [], []
                for i in idx:
                    o, grp, p, gr = self.entries[i]
                    st = o.state[p]
                    groups.append(gigs_lib.AdamGroup(p.data_ptr(), None if gr is None else gr.data_ptr(), st["exp_avg"].data_ptr(),
                                                     st["exp_avg_sq"].data_ptr(), p.numel(), 0.0, 0))
                    flags.append(1 if (watch is not None and grp.get("name") in watch) else 0)
                arr = (gigs_lib.AdamGroup * len(groups))(*groups)
                if guard is not None:
                    use_watch = changed is not None and any(flags)
                    gigs_lib.check(_lib.gigs_adam_step_guarded(len(groups), C.cast(arr, C.c_void_p), b1, b2, eps, 0,
                                                               self.table[row:].data_ptr(), bytes(flags) if use_watch else None,
                                                               changed.data_ptr() if use_watch else None, guard.data_ptr(),
                                                               stream), "adam_step_guarded")
                elif changed is not None and any(flags):
                    gigs_lib.check(_lib.gigs_adam_step_watch(len(groups), C.cast(arr, C.c_void_p), b1, b2, eps, 0,
                                                             self.table[row:].data_ptr(), bytes(flags), changed.data_ptr(),
                                                             stream), "adam_step_watch")
                else:
                    gigs_lib.check(_lib.gigs_adam_step_dyn(len(groups), C.cast(arr, C.c_void_p), b1, b2, eps, 0,
                                                           self.table[row:].data_ptr(), stream), "adam_step_dyn")
                row += len(idx)

    def advance(self) -> None:
        """Host side of one update: step counts += 1, table := this step's scalars (async copy on the current stream)."""
        host, ev = self.hosts[self._turn], self._copied[self._turn]
        if ev is not None:
            ev.synchronize()
        for row, i in enumerate(self.order):
            o, group, p, _ = self.entries[i]
            st = o.state[p]
            st["step"] += 1
            _lib.gigs_adam_scalars(float(group["lr"]), int(st["step"]), float(group["betas"][0]), float(group["betas"][1]),
                                   self._sc)
            host[row, 0], host[row, 1] = self._sc[0], self._sc[1]
        self.table.copy_(host, non_blocking=True)
        if ev is None:
            ev = self._copied[self._turn] = torch.cuda.Event()
        ev.record()
        self._turn ^= 1

    def retreat(self) -> None:
        """Undo advance() for a step that is being repeated (binning overflow: its update was never replayed)."""
        for i in self.order:
            o, _, p, _ = self.entries[i]
            o.state[p]["step"] -= 1
