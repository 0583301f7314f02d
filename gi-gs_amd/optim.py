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
