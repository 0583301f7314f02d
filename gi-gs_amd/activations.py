"""The parameter getters of the reference's GaussianModel (scene/gaussian_model.py:48-58, 178-263) as one kernel
forward and one backward (gigs_activate_fwd / _bwd) instead of eight getters' worth of torch ops per iteration.

    g = activate(raw)     raw: {"xyz", "f_dc", "f_rest", "opacity", "normal", "albedo", "roughness", "metallic",
                                "scaling", "rotation"} (the optimizer's tensors)
                          g:   {"means3D", "shs", "opacities", "normal", "albedo", "roughness", "metallic", "scales",
                                "rotations"} (what GaussianRasterizer takes, gaussian_renderer/__init__.py:66-155)
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Dict

import torch

import gigs_lib

_lib = gigs_lib.lib()

RAW = ("f_dc", "f_rest", "opacity", "normal", "albedo", "roughness", "metallic", "scaling", "rotation")
OUT = ("shs", "opacities", "normal", "albedo", "roughness", "metallic", "scales", "rotations")
_OUT_OF = dict(opacities="opacity", normal="normal", albedo="albedo", roughness="roughness", metallic="metallic",
               scales="scaling", rotations="rotation")
_UPSTREAM_OF = dict(f_dc="shs", f_rest="shs", opacity="opacities", normal="normal", albedo="albedo", roughness="roughness",
                    metallic="metallic", scaling="scales", rotation="rotations")  # raw parameter -> the getter it feeds


class _Scope(threading.local):
    sink = None


_st = _Scope()  # per thread, like the scopes of diff_gaussian_rasterization


class grad_sink:
    """`with grad_sink({"albedo": t0, "f_dc": t1, ...}):` -- the backward of `activate` writes the gradient of the named raw
    parameter into the given tensor instead of a fresh one (names: RAW; a tensor is used only if shape, dtype and device
    fit).  With views of a dp.GradSlab the raw gradients are born inside the all-reduce buffer, which the captured Adam
    launch then reads in place (train_iteration.Stage2Trainer.data_parallel)."""

    def __init__(self, tensors):
        self.tensors = dict(tensors)

    def __enter__(self):
        self._prev, _st.sink = _st.sink, self.tensors
        return self

    def __exit__(self, *exc):
        _st.sink = self._prev
        return False


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *raw):
        if not raw[0].is_cuda:
            raise RuntimeError("activate needs CUDA/HIP tensors: gigs-hip has no CPU path")
        raw = tuple(t.contiguous().float() for t in raw)
        r = dict(zip(RAW, raw))
        P = int(r["f_dc"].shape[0])
        K = 1 + int(r["f_rest"].shape[1])
        if r["f_dc"].shape != (P, 1, 3) or r["f_rest"].shape != (P, K - 1, 3) or r["rotation"].shape != (P, 4):
            raise ValueError("activate: unexpected parameter shapes")
        dev = raw[0].device
        # gigs_ctx_set_split_sh (the context's sh_rest IS this call's f_rest tensor): the rasterizer reads f_dc and f_rest as they
        # are, nothing is concatenated; the node's "shs" output is then an empty placeholder (activate() hands out f_dc instead)
        rest = gigs_lib.current().sh_rest
        split = rest is not None and K > 1 and rest.data_ptr() == r["f_rest"].data_ptr()
        out = {"shs": torch.empty((0,) if split else (P, K, 3), dtype=torch.float32, device=dev)}
        for o, src in _OUT_OF.items():
            out[o] = torch.empty_like(r[src])
        a = gigs_lib.ActivationRaw(*[r[n].data_ptr() for n in RAW])
        b = gigs_lib.ActivationOut(*[None if (split and n == "shs") else out[n].data_ptr() for n in OUT])
        with torch.cuda.device(dev):
            gigs_lib.check(_lib.gigs_activate_fwd(P, K, C.addressof(a), C.addressof(b),
                                                  torch.cuda.current_stream().cuda_stream), "activate_fwd")
        ctx.save_for_backward(*raw)
        ctx.P, ctx.K = P, K
        ctx.sink = _st.sink  # backward() runs on autograd's device thread: the sink of THIS call travels with the node
        # Declared stage-2 gradient set (gigs_lib.Context.materials_only): the rasterizer's backward hands back None for the
        # gradients that are exact zeros; so does this node (no tensor, nothing written) -- the captured Adam launch then
        # updates those groups with g = 0.  Without the declaration an absent gradient is written as zeros, as before.
        ctx.absent_is_none = gigs_lib.current().materials_only is not None
        ctx.set_materialize_grads(False)
        return tuple(out[n] for n in OUT)

    @staticmethod
    def backward(ctx, *g_out):
        raw = ctx.saved_tensors
        dev = raw[0].device
        g = [None if t is None else t.contiguous().float() for t in g_out]
        sink = ctx.sink or _st.sink or {}
        upstream = dict(zip(OUT, g))
        d = []
        for name, t in zip(RAW, raw):
            if ctx.absent_is_none and upstream[_UPSTREAM_OF[name]] is None:
                d.append(None)
                continue
            v = sink.get(name)
            ok = v is not None and v.shape == t.shape and v.dtype == t.dtype and v.device == t.device and v.is_contiguous()
            d.append(v if ok else torch.empty_like(t))
        if all(t is None for t in d):
            return tuple(d)
        a = gigs_lib.ActivationRaw(*[t.data_ptr() for t in raw])
        b = gigs_lib.ActivationOut(*[None if t is None else t.data_ptr() for t in g])
        c = gigs_lib.ActivationRaw(*[None if t is None else t.data_ptr() for t in d])
        with torch.cuda.device(dev):
            gigs_lib.check(_lib.gigs_activate_bwd(ctx.P, ctx.K, C.addressof(a), C.addressof(b), C.addressof(c),
                                                  torch.cuda.current_stream().cuda_stream), "activate_bwd")
        return tuple(d)


def activate(raw: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = _Activate.apply(*[raw[n] for n in RAW])
    g = dict(zip(OUT, out))
    if g["shs"].numel() == 0 and raw["f_dc"].numel() != 0:
        g["shs"] = raw["f_dc"]  # split SH (gigs_ctx_set_split_sh): the rasterizer takes the degree-0 tensor, f_rest via its context
    g["means3D"] = raw["xyz"]
    return g


def activate_torch(raw: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The same getters as the reference writes them (scene/gaussian_model.py:178-263): eight torch op chains.  The
    checker of `activate` and the 'torch glue' leg of the iteration benchmarks."""
    F = torch.nn.functional
    return dict(means3D=raw["xyz"], shs=torch.cat((raw["f_dc"], raw["f_rest"]), dim=1), opacities=torch.sigmoid(raw["opacity"]),
                normal=F.normalize(raw["normal"], dim=-1), albedo=torch.sigmoid(raw["albedo"]),
                roughness=torch.sigmoid(raw["roughness"]), metallic=torch.sigmoid(raw["metallic"]),
                scales=torch.exp(raw["scaling"]), rotations=F.normalize(raw["rotation"]))
