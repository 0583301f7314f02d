"""CubemapLight with the reference's interface (pbr/light.py:84-207); mip chain and GGX
pre-filter run in libgigs_hip.so.  Image I/O (cv2) parts of the reference class are out of scope."""
from __future__ import annotations

import os
import threading
from typing import Optional

import torch
import torch.nn as nn

import gigs_lib

from .renderutils import diffuse_cubemap, specular_cubemap
from .renderutils.ops import specular_cubemap_levels

_lib = gigs_lib.lib()


class _Tls(threading.local):
    build_here = False


_tls = _Tls()


class build_on_current_stream:
    """`with build_on_current_stream():` -- CubemapLight.build_mips() runs its launches on the caller's current stream instead
    of the light's own one (a stepper that schedules the light itself: pipeline._fused_begin)."""

    def __enter__(self):
        self._prev, _tls.build_here = _tls.build_here, True
        return self

    def __exit__(self, *exc):
        _tls.build_here = self._prev
        return False


class cubemap_mip(torch.autograd.Function):
    """pbr/light.py:54-79: forward = 2x2 average pool, backward = bilinear cube lookup of 0.25*dout."""

    @staticmethod
    def forward(ctx, cubemap: torch.Tensor) -> torch.Tensor:
        if not cubemap.is_cuda:
            raise RuntimeError("cubemap must be a CUDA/HIP tensor: pbr (gigs-hip) has no CPU path")
        c = cubemap.contiguous().float()
        r, C = c.shape[1] // 2, c.shape[3]
        out = torch.empty((6, r, r, C), dtype=torch.float32, device=c.device)
        with torch.cuda.device(c.device):
            gigs_lib.check(_lib.gigs_cubemap_mip_fwd(r, C, c.data_ptr(), out.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream), "cubemap_mip_fwd")
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor) -> torch.Tensor:
        d = dout.contiguous().float()
        r = d.shape[1]
        assert d.shape[3] == 3
        out = torch.empty((6, 2 * r, 2 * r, 3), dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            gigs_lib.check(_lib.gigs_cubemap_mip_bwd(r, d.data_ptr(), out.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream), "cubemap_mip_bwd")
        return out


class _mip_chain(torch.autograd.Function):
    """base -> (base, mip1, ..., mipN, mipN') by repeated cubemap_mip, as ONE autograd node.  Every level also feeds a GGX
    filter, the coarsest one the diffuse filter as well (through its copy mipN') and the base the chain itself, so op-by-op
    autograd runs a separate accumulation pass per level before every cubemap_mip backward; here every sum is formed inside
    the mip backward kernel (gigs_cubemap_mip_bwd_add2: the same additions, in the same order of operations per element).
    Same arithmetic as the chain of cubemap_mip calls."""

    @staticmethod
    def forward(ctx, base, n_levels):
        if not base.is_cuda:
            raise RuntimeError("cubemap must be a CUDA/HIP tensor: pbr (gigs-hip) has no CPU path")
        ctx.set_materialize_grads(False)  # an unused output arrives as None, not as a zero tensor
        cur = base.contiguous().float()
        outs = []
        with torch.cuda.device(cur.device):
            s = torch.cuda.current_stream().cuda_stream
            for _ in range(n_levels):
                r = cur.shape[1] // 2
                nxt = torch.empty((6, r, r, 3), dtype=torch.float32, device=cur.device)
                gigs_lib.check(_lib.gigs_cubemap_mip_fwd(r, 3, cur.data_ptr(), nxt.data_ptr(), s), "cubemap_mip_fwd")
                outs.append(nxt)
                cur = nxt
        return (base.view_as(base), *outs, outs[-1].clone())

    @staticmethod
    def backward(ctx, g_base, *gs):
        # walk back from the coarsest level; gs[k] is the gradient level k+1 receives from its filter (None = zero), the
        # last entry the one the coarsest level's copy receives (the diffuse filter's)
        gs = [None if g is None else g.contiguous().float() for g in gs]
        g_base = None if g_base is None else g_base.contiguous().float()
        gs, g_dup = gs[:-1], gs[-1]
        G = gs[-1]
        if G is None:
            G, g_dup = g_dup, None
        if G is None:
            raise RuntimeError("_mip_chain.backward: the coarsest level received no gradient")
        dev = G.device
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        with torch.cuda.device(dev):
            s = torch.cuda.current_stream().cuda_stream
            for k in range(len(gs) - 1, -1, -1):
                r = G.shape[1]
                fine = torch.empty((6, 2 * r, 2 * r, 3), dtype=torch.float32, device=dev)
                add = gs[k - 1] if k > 0 else g_base
                gigs_lib.check(_lib.gigs_cubemap_mip_bwd_add2(r, G.data_ptr(), p(g_dup), p(add), fine.data_ptr(), s),
                               "cubemap_mip_bwd_add2")
                G, g_dup = fine, None
        return G, None


class CubemapLight(nn.Module):
    LIGHT_MIN_RES = 16
    MIN_ROUGHNESS = 0.08
    MAX_ROUGHNESS = 0.5

    def __init__(self, base_res: int = 16, scale: float = 0.5, bias: float = 0.25, path=None, device="cuda") -> None:
        super().__init__()
        self.mtx = None
        self.path = path
        base = torch.rand(6, base_res, base_res, 3, dtype=torch.float32, device=device) * scale + bias
        self.base = nn.Parameter(base)
        self.register_parameter("env_base", self.base)
        self._pre, self._wanted, self._prefetch_ok, self._side, self._ready = None, False, True, None, None
        self.prefetch_stats = dict(started=0, adopted=0, discarded=0)
        gigs_lib.prefetchers.add(self)

    def xfm(self, mtx) -> None:
        self.mtx = mtx

    def clamp_(self, min: Optional[float] = None, max: Optional[float] = None) -> None:
        self.base.clamp_(min, max)

    def get_mip(self, roughness: torch.Tensor) -> torch.Tensor:
        return torch.where(
            roughness < self.MAX_ROUGHNESS,
            (torch.clamp(roughness, self.MIN_ROUGHNESS, self.MAX_ROUGHNESS) - self.MIN_ROUGHNESS)
            / (self.MAX_ROUGHNESS - self.MIN_ROUGHNESS) * (len(self.specular) - 2),
            (torch.clamp(roughness, self.MAX_ROUGHNESS, 1.0) - self.MAX_ROUGHNESS) / (1.0 - self.MAX_ROUGHNESS)
            + len(self.specular) - 2,
        )

    def build_mips(self, cutoff: float = 0.99) -> None:
        pre, self._pre = self._pre, None
        if not self.base.is_cuda or _tls.build_here or torch.cuda.is_current_stream_capturing():
            # inside a graph capture, or for a stepper that schedules the light itself: on the current stream
            self.specular, self.diffuse = self._build(cutoff)
            return
        self._wanted = True  # the caller filters every iteration: the next rasterizer forward may start the next one early
        main = torch.cuda.current_stream()
        if pre is not None:
            main.wait_event(pre["ready"])  # also when it is thrown away: the filter tables it may have built are complete
            if pre["key"] != self._prefetch_key(cutoff):
                # the base changed between the rasterizer's forward and this call (or the grad mode / cutoff did): this
                # caller's order of calls does not suit the prefetch -- stop guessing
                self.prefetch_stats["discarded"] += 1
                self._prefetch_ok = False
                pre = None
            else:
                self.prefetch_stats["adopted"] += 1
        if pre is None:
            # not prefetched: the same launches on the light's stream, in the caller's order (the stream waits for the
            # caller's, the caller's for the result).  Every graph over `base` is built there, so `base`'s gradient
            # accumulator belongs to that stream too: see prefetch()
            side = self._side_stream()
            side.wait_stream(main)
            pre = self._build_on_side(cutoff)
            main.wait_event(pre["ready"])
        for t in [pre["diffuse"], *pre["specular"]]:
            t.record_stream(main)
        self.specular, self.diffuse = pre["specular"], pre["diffuse"]

    # ---- drop-in overlap (gigs-hip extension; GIGS_LIGHT_PREFETCH=0 switches it off) ------------------------------------
    # train.py:330-345 calls render() (the rasterizer, with its SSAO march) and THEN cubemap.build_mips(): op by op, the
    # 0.3 ms GGX pre-filter and its 0.3 ms backward sit on the step's critical path.  They depend on `base` alone, which only
    # the optimizer step changes, so the rasterizer's forward starts the filter the caller is about to ask for on this light's
    # side stream, at the moment its own blend kernel starts (diff_gaussian_rasterization.GaussianRasterizer.forward); the
    # build_mips() call that follows adopts the result if `base` is still the same tensor at the same version, under the same
    # grad mode and cutoff, and waits for it on the caller's stream.  autograd runs a node's backward on the stream of its
    # forward, so the filter's backward overlaps the rasterizer's backward the same way -- provided `base`'s gradient
    # accumulator lives on the light's stream as well: it runs as soon as the filter's backward is queued, BEFORE the
    # rasterizer's backward node, and one that lived on the caller's stream would make that stream wait for the filter's
    # backward right there.  A light is prefetched only when its build_mips() was called since the previous rasterizer
    # forward, and never again after a prefetch had to be thrown away: a caller with another order of calls pays for one
    # wasted filter, not for one per iteration.
    def _prefetch_key(self, cutoff):
        return (id(self.base), self.base._version, float(cutoff), bool(torch.is_grad_enabled() and self.base.requires_grad))

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.base.device)
            self._ready = torch.cuda.Event()
        return self._side

    def _build_on_side(self, cutoff):
        key = self._prefetch_key(cutoff)
        with torch.cuda.stream(self._side):
            specular, diffuse = self._build(cutoff)
            self._ready.record()
        return dict(key=key, specular=specular, diffuse=diffuse, ready=self._ready)

    def wants_prefetch(self) -> bool:
        return (self._wanted and self._prefetch_ok and self._pre is None and self.base.is_cuda
                and os.environ.get("GIGS_LIGHT_PREFETCH", "1") == "1")

    def prefetch(self, step_event, blend_event, cutoff: float = 0.99) -> None:
        self._wanted = False
        side = self._side_stream()
        side.wait_event(step_event)   # everything that wrote `base` (the optimizer step, clamp_) is in front of this event
        side.wait_event(blend_event)  # start with the blend kernel: beside the sort passes the filter only slows them down
        self._pre = self._build_on_side(cutoff)
        self.prefetch_stats["started"] += 1

    def _build(self, cutoff: float = 0.99):
        specular = [self.base]
        diffuse_in = None
        if os.environ.get("GIGS_MIP_CHAIN", "1") == "1" and self.base.shape[3] == 3:
            n_levels, r = 0, self.base.shape[1]
            while r > self.LIGHT_MIN_RES:
                n_levels, r = n_levels + 1, r // 2
            if n_levels:
                chain = _mip_chain.apply(self.base, n_levels)
                specular = list(chain[:-1])  # the base as the node's output: its GGX gradient enters the chain's last step
                diffuse_in = chain[-1]
        else:
            while specular[-1].shape[1] > self.LIGHT_MIN_RES:
                specular += [cubemap_mip.apply(specular[-1])]
        n = len(specular)
        rough = [(idx / (n - 2)) * (self.MAX_ROUGHNESS - self.MIN_ROUGHNESS) + self.MIN_ROUGHNESS for idx in range(n - 1)] + [1.0]
        coarsest = diffuse_in if diffuse_in is not None else specular[-1]
        diffuse_first = os.environ.get("GIGS_LIGHT_DIFFUSE_LAST", "1") != "1"
        diffuse = None
        if diffuse_first:
            diffuse = diffuse_cubemap(coarsest)
        # the levels are independent: one launch filters them all (and one launch back-propagates them all)
        merged = specular_cubemap_levels(specular, rough, cutoff)
        if merged is not None:
            if not diffuse_first:
                # created after the GGX node, the diffuse filter's backward runs BEFORE the GGX backward (autograd walks
                # the later node first): 0.02 ms of short workgroups in front of the launch that floods every CU
                diffuse = diffuse_cubemap(coarsest)
            return merged, diffuse
        if not diffuse_first:
            diffuse = diffuse_cubemap(coarsest)
        for idx in range(n - 1):
            specular[idx] = specular_cubemap(specular[idx], rough[idx], cutoff)
        specular[-1] = specular_cubemap(specular[-1], 1.0, cutoff)
        return specular, diffuse
