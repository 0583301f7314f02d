"""Drop-in for the reference's `diff_gaussian_rasterization` package on MI355X.

Same public names, argument order, return tuples and error messages as
submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py ("R/…/__init__.py")
of piotrmwojcik/GI-GS:

    GaussianRasterizationSettings   R/…/__init__.py:31-51
    GaussianRasterizer              :375-537   (forward 12-tuple, markVisible)
    Gaussian_SSR                    :696-743
    _C.{rasterize_gaussians, rasterize_gaussians_backward, mark_visible, depth_to_normal,
        SSAO, SSR}                  R/ext.cpp:16-23, R/rasterize_points.cu

but every kernel behind it is the HIP implementation in libgigs_hip.so, reached through the
C ABI of include/gigs_hip.h with raw device pointers and the current torch stream.  The
kornia 3x3 median / bilateral blurs the reference calls inside `GaussianRasterizer.forward`
(:478, :491, :504) are HIP kernels as well (`filters`).  There is no CPU path: tensors must
live on a HIP device and the shared library must be present.
"""
from __future__ import annotations

import ctypes as C
import contextlib
import os
import sys
import threading
from types import SimpleNamespace
from typing import NamedTuple, Optional, Tuple

import torch
import torch.nn as nn

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG_ROOT not in sys.path:
    sys.path.insert(0, _PKG_ROOT)

import gigs_lib  # noqa: E402
from gigs_lib import GigsError  # noqa: E402,F401

_lib = gigs_lib.lib()  # fail at import time if the native library is missing

NUM_CHANNELS = 3


def cpu_deep_copy_tuple(input_tuple: Tuple) -> Tuple:
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    radius: float
    bias: float
    thick: float
    delta: float
    step: int
    start: int
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    inference: bool
    argmax_depth: bool


# ------------------------------------------------------------------------------------------
# pointer plumbing
# ------------------------------------------------------------------------------------------
def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _Scopes(threading.local):
    """The dynamic scopes of this module (`with OutputPool / after_blend / grad_sink / AsyncBinning / view_cache`), per
    THREAD like the library contexts of gigs_lib: two Python threads driving two streams do not see each other's."""
    pool = None         # OutputPool
    after_blend = None  # callable
    prefetch_events = None  # {device: (step begin, blend begin)} events of the drop-in light prefetch
    grad_sink = None    # {name: tensor}
    async_ = None       # AsyncBinning
    view = None         # (ViewSlot, "record" | "replay" | "replay_rec") while a frozen-geometry view cache is active


_st = _Scopes()


class OutputPool:
    """Optional reuse of the operator's output planes across calls (gigs-hip extension, off by default).

    A training loop that replays the rest of its iteration from a hipGraph wants the rasterizer's planes at
    the same addresses every step (the graph then reads them in place instead of copying them into its static
    inputs).  Inside `with pool:` every output plane of this package is taken from the pool -- so tensors
    returned by an earlier call under the same pool are overwritten by the next one."""

    def __init__(self):
        self.buffers, self.counts = {}, {}

    def __enter__(self):
        self.counts = {}
        self._prev, _st.pool = _st.pool, self
        return self

    def __exit__(self, *exc):
        _st.pool = self._prev
        return False

    def get(self, tag, shape, dtype, device):
        k = self.counts.get(tag, 0)
        self.counts[tag] = k + 1
        key = (tag, k, tuple(shape), dtype, str(device))
        t = self.buffers.get(key)
        if t is None:
            t = self.buffers[key] = torch.empty(tuple(shape), dtype=dtype, device=device)
        return t




class after_blend:
    """`with after_blend(fn):` -- GaussianRasterizer.forward calls fn() once the alpha-blend kernel has been
    queued and before the in-op filters and SSAO are (gigs-hip extension, off by default)."""

    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        self._prev, _st.after_blend = _st.after_blend, self.fn
        return self

    def __exit__(self, *exc):
        _st.after_blend = self._prev
        return False




class ViewSlot:
    """Where a view's geometry-dependent, material-independent state lives while a step runs (gigs-hip extension, off by
    default): the binning and image chunks of the rasterizer's forward (ranges, tile order, sorted point list) and the
    operator's occlusion plane, at FIXED addresses, so that a hipGraph can be captured once to write them ("record") and
    once to read them ("replay").  `export()` / `load()` move the parts that matter (point_list, ranges, tile_order,
    occlusion: ~18 MB at 800 x 800 with 1.9 M instances) to and from a per-view entry -- pipeline.GeometryCache."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.binning, self.img = _Scratch(self.device), _Scratch(self.device)
        self.occlusion = None
        self.meta = None  # (P, W, H, capacity) of the forward that sized the chunks

    def _parts(self):
        P, W, H, cap = self.meta
        T = ((W + 15) // 16) * ((H + 15) // 16)
        b, i = self.binning.t, self.img.t
        ob = int(_lib.gigs_binning_offset(cap, 3))
        orng, oord = int(_lib.gigs_image_offset(W, H, 2)), int(_lib.gigs_image_offset(W, H, 3))
        parts = {"point_list": b[ob:ob + 4 * cap], "ranges": i[orng:orng + 8 * T], "tile_order": i[oord:oord + 4 * T]}
        if self.occlusion is not None:
            parts["occlusion"] = self.occlusion
        return parts

    def export(self) -> dict:
        return {k: v.clone() for k, v in self._parts().items()}

    def export_hits(self) -> dict:
        n = int(self.ssr_total_host[0])  # valid: the caller exports after the recording forward has finished
        return {"ssr_offsets": self.ssr_offsets.clone(), "ssr_entries": self.ssr_entries[:n].clone()}

    def load(self, entry: dict) -> None:
        for k, v in self._parts().items():
            v.copy_(entry[k], non_blocking=True)
        self.ssr_loaded = "ssr_offsets" in entry
        if self.ssr_loaded:
            self.ssr_offsets.copy_(entry["ssr_offsets"], non_blocking=True)
            self.ssr_entries[:entry["ssr_entries"].shape[0]].copy_(entry["ssr_entries"], non_blocking=True)

    # -- the indirect-light hit list (gigs_ssr_hits / gigs_ssr_apply) ------------------------------------------------
    ssr_offsets = ssr_counts = ssr_entries = ssr_total_host = None
    ssr_capacity = 0
    ssr_loaded = False

    def ssr_ok(self) -> bool:
        """After a recording forward has finished: did the hit list fit its buffer?"""
        return self.ssr_total_host is not None and int(self.ssr_total_host[0]) <= self.ssr_capacity

    def ssr_prepare(self, n_pixels: int, total_hint=None) -> None:
        """(Re)allocates the hit-list buffers: offsets / counts by the image size, entries with 30 % headroom over
        `total_hint` hits.  Outside a hipGraph capture only (the recording step's eager warm-up sizes them)."""
        if self.ssr_offsets is None or self.ssr_offsets.numel() != 4 * n_pixels + 1:
            self.ssr_offsets = torch.zeros(4 * n_pixels + 1, dtype=torch.int32, device=self.device)
            self.ssr_counts = torch.zeros(4 * n_pixels, dtype=torch.int32, device=self.device)
            self.ssr_total_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        if total_hint is not None and total_hint > self.ssr_capacity:
            self.ssr_capacity = max(1 << 16, int(1.3 * total_hint))
            self.ssr_entries = torch.empty((self.ssr_capacity, 2), dtype=torch.int32, device=self.device)


class view_cache:
    """`with view_cache(slot, "record"):` -- the rasterizer's forward bins into the slot's chunks and the operator's SSAO
    writes the slot's occlusion plane; `with view_cache(slot, "replay"):` -- the forward REUSES the slot's tile lists
    (gigs_ctx_set_reuse_binning: preprocess + blend only), SSAO returns the slot's plane without marching and Gaussian_SSR
    gathers at the slot's hit list when it holds one (gigs_ssr_apply); "replay_rec" = replay whose Gaussian_SSR still marches
    and RECORDS that hit list (a view's first visit after its geometry is known to be frozen).  Valid only while positions,
    covariances, opacities and normals are what they were at the recording -- the caller's knowledge."""

    def __init__(self, slot: ViewSlot, mode: str):
        if mode not in ("record", "replay", "replay_rec"):
            raise ValueError("view_cache: mode must be 'record', 'replay' or 'replay_rec'")
        self.pair = (slot, mode)

    def __enter__(self):
        self._prev, _st.view = _st.view, self.pair
        return self

    def __exit__(self, *exc):
        _st.view = self._prev
        return False


class BinningOverflow(RuntimeError):
    """An asynchronous forward produced more instances than the binning capacity holds (`.needed` = the count)."""

    def __init__(self, needed: int, capacity: int):
        super().__init__(f"async binning overflow: {needed} instances, capacity {capacity}")
        self.needed, self.capacity = int(needed), int(capacity)


class AsyncBinning:
    """`with AsyncBinning(capacity, device):` -- rasterizer forwards inside bin into a persistent buffer of `capacity`
    instances and read NOTHING back (gigs_ctx_set_async_binning: the setting lives in the library context that is current
    inside the block, not in the process; gigs-hip extension, off by default).  The reference reads
    the instance count in the middle of every forward (rasterizer_impl.cu:589-594), which stalls the host and keeps the
    forward out of a hipGraph; here the count stays on the device and `num_rendered` is the capacity (the backward
    carves the same layout from it).  Overflow protocol: `snapshot()` after a forward queues a copy of the device
    counters to pinned memory, `check()` later waits for that copy only and raises BinningOverflow if the forward had
    more instances than the capacity (its surplus was dropped) -- the caller grows the capacity and repeats the step."""

    def __init__(self, capacity: int, device):
        self.capacity = int(capacity)
        if self.capacity <= 0:
            raise ValueError("AsyncBinning: capacity must be positive")
        self.device = torch.device(device)
        self.counters = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.host = torch.zeros(2, dtype=torch.int32).pin_memory()
        self.event = torch.cuda.Event()
        self._snap = False

    def __enter__(self):
        self._prev, _st.async_ = _st.async_, self
        self._use = gigs_lib.use(gigs_lib.current().derive(async_binning=(self.capacity, self.counters)))
        self._use.__enter__()
        return self

    def __exit__(self, *exc):
        self._use.__exit__(*exc)
        _st.async_ = self._prev
        return False

    def snapshot(self) -> None:
        with torch.cuda.device(self.device):
            self.host.copy_(self.counters, non_blocking=True)
            self.event.record()
        self._snap = True

    def check(self) -> int:
        """Instance count of the forward before the last snapshot(); raises BinningOverflow if it did not fit."""
        if not self._snap:
            return -1
        self.event.synchronize()
        r, over = int(self.host[0]), int(self.host[1])
        if over:
            raise BinningOverflow(over, self.capacity)
        return r


class grad_sink:
    """`with grad_sink({"means3D": t0, "opacity": t1, ...}):` -- the rasterizer's backward writes the gradient of the
    named operator input into the given tensor instead of a fresh one (gigs-hip extension, off by default).  Names:
    means3D, opacity, normal, albedo, roughness, metallic, sh, scales, rotations, colors, cov3D, means2D; a tensor is
    used only if its shape, dtype and device fit.  dp.GradSlab.sink() provides views of one flat bucket, so the
    gradients are born inside the all-reduce buffer (no torch.cat per step)."""

    def __init__(self, tensors):
        self.tensors = dict(tensors)

    def __enter__(self):
        self._prev, _st.grad_sink = _st.grad_sink, self.tensors
        return self

    def __exit__(self, *exc):
        _st.grad_sink = self._prev
        return False


def _new(tag: str, shape, device, dtype=torch.float32) -> torch.Tensor:
    """Uninitialised output tensor (every element is written by the kernel that receives it)."""
    if _st.pool is not None:
        return _st.pool.get(tag, shape, dtype, device)
    return torch.empty(tuple(shape), dtype=dtype, device=device)


def _need_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} must be a CUDA/HIP tensor: diff_gaussian_rasterization (gigs-hip) has no CPU path")


def _fptr(t: Optional[torch.Tensor], name: str, device=None) -> Tuple[Optional[int], Optional[torch.Tensor]]:
    """float32 device pointer of `t`, or NULL for None / empty (the reference's empty-tensor
    convention, R/…/__init__.py:435-445).  Returns (ptr, keepalive)."""
    if t is None or t.numel() == 0:
        return None, None
    _need_gpu(t, name)
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.contiguous().float()
    return t.data_ptr(), t


class _Scratch:
    """One resizable byte buffer + the ctypes callback that hands it to the library; the
    counterpart of resizeFunctional (R/rasterize_points.cu:31-37)."""

    def __init__(self, device):
        self.t = torch.empty(0, dtype=torch.uint8, device=device)

        def _alloc(nbytes, _user):
            try:
                self.t.resize_(int(nbytes))
                return self.t.data_ptr()
            except Exception:  # noqa: BLE001 - reported by the library as GIGS_ERR_ALLOC
                return 0

        self.cb = gigs_lib.ALLOC_FN(_alloc)


# ------------------------------------------------------------------------------------------
# `_C` functions
# ------------------------------------------------------------------------------------------
def _rasterize_gaussians(bg, means3D, colors_precomp, opacities, normal, albedo, roughness, metallic,
                         scales, rotations, cov3Ds_precomp, sh, campos, viewmatrix, projmatrix,
                         scale_modifier, tanfovx, tanfovy, image_height, image_width, sh_degree,
                         prefiltered, argmax_depth, inference, debug):
    """_C.rasterize_gaussians: 25 positional args -> 14-tuple (R/rasterize_points.cu:130-252)."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _need_gpu(means3D, "means3D")
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    fopts = dict(dtype=torch.float32, device=dev)
    # the reference zero-fills every output (rasterize_points.cu:170-179); the kernels overwrite every
    # pixel / every Gaussian when P > 0, so one uninitialised slab is carved instead of 10 fill launches
    if P != 0:
        slab = _new("gbuffer", (20, H, W), dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
    else:
        slab = torch.zeros((20, H, W), **fopts)
        radii = torch.zeros((P,), dtype=torch.int32, device=dev)
    out_color, out_normal, out_normal_view = slab[0:3], slab[3:6], slab[6:9]
    out_pos, out_albedo = slab[9:12], slab[12:15]
    out_opacity, out_depth, out_roughness, out_metallic = slab[15:16], slab[16:17], slab[17:18], slab[18:19]
    geom, binning, img = _Scratch(dev), _Scratch(dev), _Scratch(dev)
    ctx_ptr = gigs_lib.ctx_ptr()
    if _st.view is not None and P != 0:
        # frozen-geometry view cache: bin into (record) / blend from (replay) the slot's fixed chunks
        slot, mode = _st.view
        if _st.async_ is None:
            raise RuntimeError("view_cache needs asynchronous binning (AsyncBinning): the slot's chunks have its capacity")
        binning, img = slot.binning, slot.img
        slot.meta = (P, W, H, _st.async_.capacity)
        if mode != "record":
            ctx_ptr = gigs_lib.current().derive(reuse_binning=True).ptr
    rendered = 0
    if P != 0:
        M = int(sh.size(1)) if sh is not None and sh.numel() != 0 else 0
        rest = gigs_lib.current().sh_rest  # gigs_ctx_set_split_sh: `sh` is the degree-0 tensor, the others live in `rest`
        if rest is not None and M:
            if M != 1 or int(rest.shape[0]) != P or rest.dim() != 3 or not rest.is_contiguous() or rest.device != dev:
                raise RuntimeError("split SH: shs must be the [P,1,3] degree-0 tensor and the context's sh_rest a contiguous "
                                   "[P,M-1,3] tensor on the same device")
            M = 1 + int(rest.shape[1])
        keep = []

        def p(t, name):
            ptr, k = _fptr(t, name)
            keep.append(k)
            return ptr

        with torch.cuda.device(dev):
            rendered = _lib.gigs_forward(
                ctx_ptr, geom.cb, None, binning.cb, None, img.cb, None, P, int(sh_degree), M, p(bg, "bg"), W, H,
                p(means3D, "means3D"), p(sh, "sh"), p(colors_precomp, "colors_precomp"),
                p(opacities, "opacities"), p(normal, "normal"), p(albedo, "albedo"),
                p(roughness, "roughness"), p(metallic, "metallic"), p(scales, "scales"),
                float(scale_modifier), p(rotations, "rotations"), p(cov3Ds_precomp, "cov3Ds_precomp"),
                p(viewmatrix, "viewmatrix"), p(projmatrix, "projmatrix"), p(campos, "campos"),
                float(tanfovx), float(tanfovy), int(bool(prefiltered)), int(bool(argmax_depth)),
                int(bool(inference)), out_color.data_ptr(), out_opacity.data_ptr(), out_depth.data_ptr(),
                out_normal.data_ptr(), out_normal_view.data_ptr(), out_pos.data_ptr(),
                out_albedo.data_ptr(), out_roughness.data_ptr(), out_metallic.data_ptr(),
                radii.data_ptr(), int(debug), _stream())
        gigs_lib.check(rendered, "rasterize_gaussians")
    return (rendered, out_color, radii, geom.t, binning.t, img.t, out_opacity, out_depth, out_normal,
            out_normal_view, out_pos, out_albedo, out_roughness, out_metallic)


def _rasterize_gaussians_backward(bg, means3D, radii, colors_precomp, normal, albedo, roughness,
                                  metallic, scales, rotations, cov3Ds_precomp, sh, campos, viewmatrix,
                                  projmatrix, scale_modifier, tanfovx, tanfovy, sh_degree, grad_depth,
                                  grad_color, grad_opacity, grad_normal, grad_albedo, grad_roughness,
                                  grad_metallic, geomBuffer, binningBuffer, imgBuffer, num_rendered, debug,
                                  image_size=None):
    """_C.rasterize_gaussians_backward: 31 args -> 12 tensors (R/rasterize_points.cu:254-364).  An incoming gradient
    may be None (= zeros, nothing is materialised); `image_size` = (H, W) is then needed if grad_color is None."""
    _need_gpu(means3D, "means3D")
    dev = means3D.device
    P = int(means3D.size(0))
    if grad_color is not None:
        H, W = int(grad_color.size(1)), int(grad_color.size(2))
    elif image_size is not None:
        H, W = int(image_size[0]), int(image_size[1])
    else:
        raise RuntimeError("rasterize_gaussians_backward: grad_color is None and no image_size was given")
    M = int(sh.size(1)) if sh is not None and sh.numel() != 0 else 0
    if M and gigs_lib.current().sh_rest is not None:  # split SH (the forward's context): M counts both tensors
        M = 1 + int(gigs_lib.current().sh_rest.shape[1])
    # every element of every gradient tensor is written by the backward kernels when P > 0
    # (the reference zero-fills 14 tensors first, rasterize_points.cu:299-312)
    _mk = torch.empty if P != 0 else torch.zeros
    sink = _st.grad_sink or {}

    # gigs_ctx_set_materials_only (the context of the forward travels with the node): the declared stage-2 gradient set --
    # the other gradients are exact zeros the library neither writes nor is given a tensor for; they come back as None
    materials_only = P != 0 and gigs_lib.current().materials_only is not None

    def z(name, *shape):
        if materials_only and name not in ("means2D", "albedo", "roughness", "metallic"):
            return None
        t = sink.get(name)
        if (t is not None and P != 0 and tuple(t.shape) == shape and t.dtype == torch.float32 and t.device == dev
                and t.is_contiguous()):
            return t
        return _mk(shape, dtype=torch.float32, device=dev)

    dL_dmeans3D, dL_dmeans2D, dL_dcolors = z("means3D", P, 3), z("means2D", P, 3), z("colors", P, NUM_CHANNELS)
    # dL_dconic [P,2,2] / dL_ddepth [P]: scratch the reference binding allocates and never returns
    # (rasterize_points.cu:302-303) -- not requested from the library (NULL)
    dL_dopacity = z("opacity", P, 1)
    dL_dnormal, dL_dalbedo = z("normal", P, 3), z("albedo", P, 3)
    dL_droughness, dL_dmetallic = z("roughness", P, 1), z("metallic", P, 1)
    dL_dcov3D, dL_dsh = z("cov3D", P, 6), z("sh", P, M, 3)
    dL_dscales, dL_drotations = z("scales", P, 3), z("rotations", P, 4)
    if P != 0:
        keep = []

        def p(t, name):
            ptr, k = _fptr(t, name)
            keep.append(k)
            return ptr

        def dp(t):
            return None if t is None else t.data_ptr()

        with torch.cuda.device(dev):
            rc = _lib.gigs_backward(
                gigs_lib.ctx_ptr(), P, int(sh_degree), M, int(num_rendered), p(bg, "bg"), W, H, p(means3D, "means3D"),
                p(sh, "sh"), p(colors_precomp, "colors_precomp"), p(normal, "normal"), p(albedo, "albedo"),
                p(roughness, "roughness"), p(metallic, "metallic"), p(scales, "scales"),
                p(rotations, "rotations"), p(cov3Ds_precomp, "cov3Ds_precomp"), p(viewmatrix, "viewmatrix"),
                p(projmatrix, "projmatrix"), p(campos, "campos"), radii.data_ptr(), float(scale_modifier),
                float(tanfovx), float(tanfovy), geomBuffer.data_ptr(), binningBuffer.data_ptr(),
                imgBuffer.data_ptr(), p(grad_depth, "grad_depth"), p(grad_color, "grad_color"),
                p(grad_opacity, "grad_opacity"), p(grad_normal, "grad_normal"), p(grad_albedo, "grad_albedo"),
                p(grad_roughness, "grad_roughness"), p(grad_metallic, "grad_metallic"),
                dL_dmeans2D.data_ptr(), None, None, dp(dL_dopacity),
                dp(dL_dnormal), dL_dalbedo.data_ptr(), dL_droughness.data_ptr(),
                dL_dmetallic.data_ptr(), dp(dL_dcolors), dp(dL_dmeans3D),
                dp(dL_dcov3D), dp(dL_dsh) if M else None, dp(dL_dscales),
                dp(dL_drotations), int(debug), _stream())
        gigs_lib.check(rc, "rasterize_gaussians_backward")
    return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dnormal, dL_dalbedo, dL_droughness, dL_dmetallic,
            dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)


def _mark_visible(means3D, viewmatrix, projmatrix):
    _need_gpu(means3D, "means3D")
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P != 0:
        m, k0 = _fptr(means3D, "means3D")
        v, k1 = _fptr(viewmatrix, "viewmatrix")
        pr, k2 = _fptr(projmatrix, "projmatrix")
        with torch.cuda.device(means3D.device):
            gigs_lib.check(_lib.gigs_mark_visible(P, m, v, pr, present.data_ptr(), _stream()), "mark_visible")
    return present


def _depth_to_normal(width, height, focal_x, focal_y, viewmatrix, depthMap):
    _need_gpu(depthMap, "depthMap")
    dev = depthMap.device
    normalMap = _new("normal_from_depth", (3, height, width), dev)  # every pixel is written
    depth_pos = _new("depth_pos", (3, height, width), dev)
    v, k0 = _fptr(viewmatrix, "viewmatrix")
    d, k1 = _fptr(depthMap, "depthMap")
    with torch.cuda.device(dev):
        gigs_lib.check(_lib.gigs_depth_to_normal(int(width), int(height), float(focal_x), float(focal_y), v, d,
                                                 normalMap.data_ptr(), depth_pos.data_ptr(), _stream()),
                       "depth_to_normal")
    return normalMap, depth_pos


def _SSAO(width, height, focal_x, focal_y, radius, bias, thick, delta, step, start, out_normal, out_pos):
    _need_gpu(out_normal, "out_normal")
    dev = out_normal.device
    if _st.view is not None:
        slot, mode = _st.view
        if slot.occlusion is None or tuple(slot.occlusion.shape) != (1, height, width):
            if mode != "record":
                raise RuntimeError("view_cache('replay'): the slot holds no occlusion plane of this size")
            slot.occlusion = torch.empty((1, height, width), dtype=torch.float32, device=dev)
        if mode != "record":
            return slot.occlusion  # what the march wrote when this view was recorded: geometry has not changed since
        occlusion = slot.occlusion
    else:
        occlusion = _new("occlusion", (1, height, width), dev)  # every pixel is written
    n, k0 = _fptr(out_normal, "out_normal")
    ps, k1 = _fptr(out_pos, "out_pos")
    scratch = _gi_scratch(int(width), int(height), dev)
    with torch.cuda.device(dev):
        gigs_lib.check(_lib.gigs_ssao_ex(gigs_lib.ctx_ptr(), int(width), int(height), float(focal_x), float(focal_y), float(radius),
                                         float(bias), float(thick), float(delta), int(step), int(start), n, ps,
                                         occlusion.data_ptr(), None if scratch is None else scratch.data_ptr(), _stream()),
                       "SSAO")
    return occlusion


def _gi_scratch(width: int, height: int, dev):
    """Scratch for the march's certification table (gigs_gi_scratch_bytes): a stream-ordered torch allocation."""
    nbytes = int(_lib.gigs_gi_scratch_bytes(width, height))
    return torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None


def _SSR(width, height, focal_x, focal_y, radius, bias, thick, delta, step, start, out_normal, out_pos,
         out_rgb, out_albedo, out_roughness, out_metallic, out_F0):
    _need_gpu(out_roughness, "out_roughness")
    dev = out_roughness.device
    color = torch.empty((3, height, width), dtype=torch.float32, device=dev)  # every pixel is written
    abd = torch.empty((3, height, width), dtype=torch.float32, device=dev)
    ptrs, keep = [], []
    for t, name in ((out_normal, "out_normal"), (out_pos, "out_pos"), (out_rgb, "out_rgb"),
                    (out_albedo, "out_albedo"), (out_roughness, "out_roughness"),
                    (out_metallic, "out_metallic"), (out_F0, "out_F0")):
        ptr, k = _fptr(t, name)
        ptrs.append(ptr)
        keep.append(k)
    scratch = _gi_scratch(int(width), int(height), dev)
    sp = None if scratch is None else scratch.data_ptr()
    W, H = int(width), int(height)
    a = (W, H, float(focal_x), float(focal_y), float(radius), float(bias), float(thick), float(delta), int(step), int(start))
    hit_list = (_st.view is not None and gigs_lib.current().option("gi_march") == 4 and int(start) < int(step)
                and os.environ.get("GIGS_SSR_HIT_LIST", "1") == "1")
    with torch.cuda.device(dev):
        if hit_list and _st.view[1] == "replay" and _st.view[0].ssr_loaded:
            # frozen geometry: gather the radiance at the recorded hits instead of marching (gigs_ssr_apply)
            slot = _st.view[0]
            normal_p, pos_p, rgb_p, albedo_p, _rough_p, metallic_p, F0_p = ptrs
            gigs_lib.check(_lib.gigs_ssr_apply(W, H, float(delta), slot.ssr_offsets.data_ptr(), slot.ssr_entries.data_ptr(),
                                               normal_p, pos_p, rgb_p, albedo_p, metallic_p, F0_p, color.data_ptr(),
                                               abd.data_ptr(), _stream()), "SSR (gather)")
        elif hit_list and _st.view[1] == "replay_rec":
            # record the hit list beside the normal outputs: count per (pixel, wave), prefix, fill.  Outside a capture (the
            # recording step's eager warm-up) the buffers are sized from the count; inside, an overflow is noticed by the
            # caller (ViewSlot.ssr_ok) after the forward and the view is simply not cached with its hit list
            slot = _st.view[0]
            capturing = torch.cuda.is_current_stream_capturing()
            if not capturing:
                slot.ssr_prepare(W * H)
            if slot.ssr_offsets is None:
                raise RuntimeError("view_cache('replay_rec'): the hit-list buffers must be sized by an eager step before a capture")
            gigs_lib.check(_lib.gigs_ssr_hits(gigs_lib.ctx_ptr(), *a, *ptrs, color.data_ptr(), abd.data_ptr(), 1,
                                              slot.ssr_counts.data_ptr(), None, None, 0, sp, _stream()), "SSR (count)")
            torch.cumsum(slot.ssr_counts, 0, dtype=torch.int32, out=slot.ssr_offsets[1:])
            if not capturing:
                slot.ssr_prepare(W * H, total_hint=int(slot.ssr_offsets[-1]))  # one read-back, in the warm-up only
            gigs_lib.check(_lib.gigs_ssr_hits(gigs_lib.ctx_ptr(), *a, *ptrs, color.data_ptr(), abd.data_ptr(), 2, None,
                                              slot.ssr_offsets.data_ptr(), slot.ssr_entries.data_ptr(), slot.ssr_capacity, sp,
                                              _stream()), "SSR (fill)")
            slot.ssr_total_host.copy_(slot.ssr_offsets[-1:], non_blocking=True)
        else:
            gigs_lib.check(_lib.gigs_ssr_ex(gigs_lib.ctx_ptr(), *a, *ptrs, color.data_ptr(), abd.data_ptr(), sp, _stream()), "SSR")
    return color, abd


def _SSR_BACKWARD(*_args):
    # The reference exports BACKWARD::SSRCUDA (R/cuda_rasterizer/backward.cu:632-808) but never calls
    # it: _SSR.backward is closed-form (R/…/__init__.py:666-673).  Kept as a named symbol only.
    raise NotImplementedError(
        "SSR_BACKWARD is unreachable in the reference (commented out at __init__.py:666-670); "
        "Gaussian_SSR's backward is grad_albedo = grad_out * abd")


def _lite_rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, cov3D_precomp, sh, campos,
                              viewmatrix, projmatrix, scale_modifier, tan_fovx, tan_fovy, image_height, image_width,
                              degree, prefiltered, argmax_depth):
    """_C.lite_rasterize_gaussians: colour / opacity / depth only "for baking" (R/rasterize_points.cu:39-127,
    liteRenderCUDA forward.cu:279-418) -> (rendered, out_color, out_opacity, radii, out_depth).  The lite kernel
    composites exactly like the full one (same tests, same depth = view-space z): gigs_lite_forward runs the full
    forward over zero material attributes and returns the three planes; the reference never calls it from Python."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _need_gpu(means3D, "means3D")
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    zero = P == 0
    mk = torch.zeros if zero else torch.empty
    out_color = mk((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
    out_opacity = mk((1, H, W), dtype=torch.float32, device=dev)
    out_depth = mk((1, H, W), dtype=torch.float32, device=dev)
    radii = mk((P,), dtype=torch.int32, device=dev)
    rendered = 0
    if not zero:
        M = int(sh.size(1)) if sh is not None and sh.numel() != 0 else 0
        geom, binning, img = _Scratch(dev), _Scratch(dev), _Scratch(dev)
        keep = []

        def p(t, name):
            ptr, k = _fptr(t, name)
            keep.append(k)
            return ptr

        with torch.cuda.device(dev):
            rendered = _lib.gigs_lite_forward(
                gigs_lib.ctx_ptr(), geom.cb, None, binning.cb, None, img.cb, None, P, int(degree), M, p(background, "background"), W, H,
                p(means3D, "means3D"), p(sh, "sh"), p(colors, "colors"), p(opacity, "opacity"), p(scales, "scales"),
                float(scale_modifier), p(rotations, "rotations"), p(cov3D_precomp, "cov3D_precomp"), p(viewmatrix, "viewmatrix"),
                p(projmatrix, "projmatrix"), p(campos, "campos"), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                int(bool(argmax_depth)), out_color.data_ptr(), out_opacity.data_ptr(), out_depth.data_ptr(), radii.data_ptr(),
                0, _stream())
        gigs_lib.check(rendered, "lite_rasterize_gaussians")
    return rendered, out_color, out_opacity, radii, out_depth


_C = SimpleNamespace(
    rasterize_gaussians=_rasterize_gaussians,
    rasterize_gaussians_backward=_rasterize_gaussians_backward,
    lite_rasterize_gaussians=_lite_rasterize_gaussians,
    mark_visible=_mark_visible,
    depth_to_normal=_depth_to_normal,
    SSAO=_SSAO,
    SSR=_SSR,
    SSR_BACKWARD=_SSR_BACKWARD,
)


# ------------------------------------------------------------------------------------------
# 3x3 filters (replace kornia.filters.median_blur / bilateral_blur for the (3, 3) case)
# ------------------------------------------------------------------------------------------
class _Median3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):  # x: [C, H, W]
        _need_gpu(x, "input")
        xc = x.contiguous().float()
        out = _new("median3x3", xc.shape, xc.device)
        Cn, H, W = xc.shape
        with torch.cuda.device(xc.device):
            gigs_lib.check(_lib.gigs_median3x3(Cn, H, W, xc.data_ptr(), out.data_ptr(), _stream()), "median3x3")
        ctx.save_for_backward(xc)
        return out

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        gc = g.contiguous().float()
        gin = torch.zeros_like(xc)
        Cn, H, W = xc.shape
        with torch.cuda.device(xc.device):
            gigs_lib.check(_lib.gigs_median3x3_backward(Cn, H, W, xc.data_ptr(), gc.data_ptr(), gin.data_ptr(),
                                                        _stream()), "median3x3_backward")
        return gin


def _median_blur(input: torch.Tensor, kernel_size=(3, 3)) -> torch.Tensor:
    """kornia.filters.median_blur for [B, C, H, W] input and a (3, 3) kernel."""
    if tuple(kernel_size) != (3, 3):
        raise NotImplementedError("only the (3, 3) median used by GI-GS is implemented")
    B, Cn, H, W = input.shape
    return _Median3x3.apply(input.reshape(B * Cn, H, W)).reshape(B, Cn, H, W)


def _bilateral_blur(input: torch.Tensor, kernel_size=(3, 3), sigma_color=1.0, sigma_space=(3.0, 3.0)):
    """kornia.filters.bilateral_blur for [B, C, H, W], (3, 3) kernel, reflect border, L1 colour
    distance (kornia defaults).  Forward only -- the reference applies it to a tensor without a
    graph (R/…/__init__.py:491)."""
    if tuple(kernel_size) != (3, 3):
        raise NotImplementedError("only the (3, 3) bilateral used by GI-GS is implemented")
    _need_gpu(input, "input")
    B, Cn, H, W = input.shape
    x = input.detach().contiguous().float()
    out = _new("bilateral3x3", x.shape, x.device)
    with torch.cuda.device(x.device):
        for b in range(B):
            gigs_lib.check(_lib.gigs_bilateral3x3(Cn, H, W, float(sigma_color), float(sigma_space[1]),
                                                  float(sigma_space[0]), x[b].data_ptr(), out[b].data_ptr(),
                                                  _stream()), "bilateral3x3")
    return out


filters = SimpleNamespace(median_blur=_median_blur, bilateral_blur=_bilateral_blur)


# ------------------------------------------------------------------------------------------
# autograd glue (A14): argument re-ordering identical to R/…/__init__.py:54-372
# ------------------------------------------------------------------------------------------
class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, opacities, normal, albedo, roughness, metallic, sh, colors_precomp,
                scales, rotations, cov3Ds_precomp, raster_settings):
        # outputs the loss does not use reach backward() as None instead of freshly filled zero planes
        ctx.set_materialize_grads(False)
        args = (
            raster_settings.bg, means3D, colors_precomp, opacities, normal, albedo, roughness, metallic,
            scales, rotations, cov3Ds_precomp, sh, raster_settings.campos, raster_settings.viewmatrix,
            raster_settings.projmatrix, raster_settings.scale_modifier, raster_settings.tanfovx,
            raster_settings.tanfovy, raster_settings.image_height, raster_settings.image_width,
            raster_settings.sh_degree, raster_settings.prefiltered, raster_settings.argmax_depth,
            raster_settings.inference, raster_settings.debug,
        )
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                res = _C.rasterize_gaussians(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            res = _C.rasterize_gaussians(*args)
        (num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, opacity_map, depth, out_normal,
         out_normal_view, out_pos, albedo_map, roughness_map, metallic_map) = res
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        # autograd runs backward() on its own device thread: the library context and the gradient sink of THIS call travel
        # with the node (gigs_lib.with_forward_context; the per-thread scopes of this module are the caller's)
        ctx.lib_ctx = gigs_lib.current()
        ctx.grad_sink = _st.grad_sink
        ctx.save_for_backward(colors_precomp, normal, albedo, roughness, metallic, means3D, scales, rotations,
                              cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        return (color, radii, opacity_map, depth, out_normal, albedo_map, roughness_map, metallic_map,
                out_normal_view, out_pos)

    @staticmethod
    @gigs_lib.with_forward_context
    def backward(ctx, grad_out_color, gard_radii=None, grad_out_opacity=None, grad_depth=None,
                 grad_out_normal=None, grad_out_albedo=None, grad_out_roughness=None, grad_out_metallic=None,
                 grad_out_normal_view=None, grad_out_pos=None):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        (colors_precomp, normal, albedo, roughness, metallic, means3D, scales, rotations, cov3Ds_precomp,
         radii, sh, geomBuffer, binningBuffer, imgBuffer) = ctx.saved_tensors
        # grad_out_normal_view and grad_out_pos are dropped, as in the reference (:243-275)
        if raster_settings.debug:  # the snapshot path keeps the reference's fully materialised argument tuple
            H, W = int(raster_settings.image_height), int(raster_settings.image_width)
            zf = lambda t, c: t if t is not None else torch.zeros((c, H, W), device=means3D.device)  # noqa: E731
            grad_out_color, grad_out_normal, grad_out_albedo = zf(grad_out_color, 3), zf(grad_out_normal, 3), zf(grad_out_albedo, 3)
            grad_out_opacity, grad_depth = zf(grad_out_opacity, 1), zf(grad_depth, 1)
            grad_out_roughness, grad_out_metallic = zf(grad_out_roughness, 1), zf(grad_out_metallic, 1)
        args = (
            raster_settings.bg, means3D, radii, colors_precomp, normal, albedo, roughness, metallic, scales,
            rotations, cov3Ds_precomp, sh, raster_settings.campos, raster_settings.viewmatrix,
            raster_settings.projmatrix, raster_settings.scale_modifier, raster_settings.tanfovx,
            raster_settings.tanfovy, raster_settings.sh_degree, grad_depth, grad_out_color, grad_out_opacity,
            grad_out_normal, grad_out_albedo, grad_out_roughness, grad_out_metallic, geomBuffer, binningBuffer,
            imgBuffer, num_rendered, raster_settings.debug,
        )
        prev_sink, _st.grad_sink = _st.grad_sink, (ctx.grad_sink if ctx.grad_sink is not None else _st.grad_sink)
        try:
            if raster_settings.debug:
                cpu_args = cpu_deep_copy_tuple(args)
                try:
                    res = _C.rasterize_gaussians_backward(*args)
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_bw.dump")
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                    raise ex
            else:
                res = _rasterize_gaussians_backward(*args, image_size=(raster_settings.image_height,
                                                                       raster_settings.image_width))
        finally:
            _st.grad_sink = prev_sink
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_normal, grad_albedo, grad_roughness,
         grad_metallic, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales, grad_rotations) = res
        return (grad_means3D, grad_means2D, grad_opacities, grad_normal, grad_albedo, grad_roughness,
                grad_metallic, grad_sh, grad_colors_precomp, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, normal, albedo, roughness, metallic, shs=None,
                colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, derive_normal=True):
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")

        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
            (scales is not None or rotations is not None) and cov3D_precomp is not None
        ):
            raise Exception(
                "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!"
            )

        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])

        # drop-in overlap: a light whose pre-filter the caller will ask for next (pbr.light.CubemapLight.build_mips, train.py:340)
        # starts it on its side stream when this forward's blend kernel starts -- unless a stepper of this package drives the
        # light itself (after_blend) or the call is being captured into a graph
        pre = ()
        if _st.after_blend is None and gigs_lib.prefetchers and means3D.is_cuda and not torch.cuda.is_current_stream_capturing():
            pre = [x for x in list(gigs_lib.prefetchers) if x.wants_prefetch()]
        if pre:
            pre = [x for x in pre if x.base.device == means3D.device]
        if pre:
            if _st.prefetch_events is None:
                _st.prefetch_events = {}
            if means3D.device not in _st.prefetch_events:  # an event belongs to the device it is first recorded on
                _st.prefetch_events[means3D.device] = (torch.cuda.Event(), torch.cuda.Event())
            step_ev, blend_ev = _st.prefetch_events[means3D.device]
            step_ev.record(torch.cuda.current_stream(means3D.device))
            lib_scope = gigs_lib.use(gigs_lib.current().derive(blend_event=blend_ev))
        else:
            lib_scope = contextlib.nullcontext()
        with lib_scope:
            (color, radii, opacity_map, depth, out_normal, albedo_map, roughness_map, metallic_map,
             out_normal_view, _) = _RasterizeGaussians.apply(
                means3D, means2D, opacities, normal, albedo, roughness, metallic, shs, colors_precomp, scales,
                rotations, cov3D_precomp, raster_settings)
        for x in pre:
            x.prefetch(step_ev, blend_ev)
        if _st.after_blend is not None:
            # gigs-hip extension: the blend kernel is queued, the VALU-bound SSAO march is next -- the point at
            # which a caller can start independent memory-bound work on another stream (see after_blend)
            _st.after_blend()

        focal_x = raster_settings.image_width / (2.0 * raster_settings.tanfovx)
        focal_y = raster_settings.image_height / (2.0 * raster_settings.tanfovy)
        W_, H_ = int(raster_settings.image_width), int(raster_settings.image_height)
        if derive_normal and W_ > 1 and H_ > 1 and os.environ.get("GIGS_FUSED_DERIVE", "1") == "1":
            # the four passes below as one launch (gigs_derive_normal; bit-identical, tests/test_gpu_parity.py)
            normal_from_depth, depth_pos_filter = _derive_normal(W_, H_, focal_x, focal_y, raster_settings.viewmatrix,
                                                                 depth.detach())
        else:
            if derive_normal:
                depth_filter = filters.median_blur(depth.detach()[None, ...], (3, 3))[0]
                normal_from_depth, depth_pos = _C.depth_to_normal(
                    raster_settings.image_width, raster_settings.image_height, focal_x, focal_y,
                    raster_settings.viewmatrix, depth_filter)
            else:
                normal_from_depth = torch.zeros_like(out_normal)
                depth_pos = torch.zeros_like(out_normal)

            normal_from_depth = filters.bilateral_blur(normal_from_depth[None, ...], (3, 3), 1, (3, 3))[0]

            depth_pos_filter = filters.median_blur(depth_pos[None, ...], (3, 3))[0]
        occlusion = _C.SSAO(
            raster_settings.image_width, raster_settings.image_height, focal_x, focal_y,
            raster_settings.radius, raster_settings.bias, raster_settings.thick, raster_settings.delta,
            raster_settings.step, raster_settings.start, out_normal_view, depth_pos_filter)

        return (color, radii, opacity_map, depth, normal_from_depth, out_normal, occlusion, albedo_map,
                roughness_map, metallic_map, out_normal_view, depth_pos_filter)


def _derive_normal(width, height, focal_x, focal_y, viewmatrix, depth):
    """median3x3(depth) -> depth_to_normal -> bilateral3x3(normal, 1, (3, 3)), median3x3(pos) in one launch."""
    _need_gpu(depth, "depth")
    d = depth.contiguous().float()
    dev = d.device
    normal = _new("normal_from_depth", (3, height, width), dev)  # every pixel is written
    pos = _new("depth_pos", (3, height, width), dev)
    vptr, vkeep = _fptr(viewmatrix, "viewmatrix")
    with torch.cuda.device(dev):
        gigs_lib.check(_lib.gigs_derive_normal(width, height, float(focal_x), float(focal_y), vptr, d.data_ptr(), 1.0, 3.0, 3.0,
                                               normal.data_ptr(), pos.data_ptr(), _stream()), "derive_normal")
    return normal, pos


class _SSR(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image_width, image_height, focal_x, focal_y, radius, bias, thick, delta, step, start,
                normal, pos, rgb, albedo, roughness, metallic, F0):
        (color, abd) = _C.SSR(image_width, image_height, focal_x, focal_y, radius, bias, thick, delta, step,
                              start, normal, pos, rgb, albedo, roughness, metallic, F0)
        ctx.save_for_backward(roughness, metallic, abd)
        ctx.set_materialize_grads(False)  # `abd` is never differentiated: no zero tensor for it
        return (color, abd)

    @staticmethod
    def backward(ctx, grad_out_color, grad_abd=None):
        roughness, metallic, abd = ctx.saved_tensors
        # closed form, R/…/__init__.py:671-673 (gigs_ssr_backward)
        if grad_out_color is None:
            grad_albedo = torch.zeros_like(abd)
        else:
            g = grad_out_color.contiguous().float()
            grad_albedo = torch.empty_like(abd)
            with torch.cuda.device(abd.device):
                gigs_lib.check(_lib.gigs_ssr_backward(int(abd.shape[2]), int(abd.shape[1]), g.data_ptr(), abd.data_ptr(),
                                                      grad_albedo.data_ptr(), _stream()), "SSR backward")
        grad_roughness = torch.zeros_like(roughness)
        grad_metallic = torch.zeros_like(metallic)
        return (None, None, None, None, None, None, None, None, None, None, None, None, None,
                grad_albedo, grad_roughness, grad_metallic, None)


class Gaussian_SSR(nn.Module):
    def __init__(self, tanfovx, tanfovy, image_width, image_height, radius, bias, thick, delta, step, start):
        super().__init__()
        self.tanfovx = tanfovx
        self.tanfovy = tanfovy
        self.image_width = image_width
        self.image_height = image_height
        self.radius = radius
        self.bias = bias
        self.thick = thick
        self.delta = delta
        self.step = step
        self.start = start

    def forward(self, normal, pos, rgb, albedo, roughness, metallic, F0):
        focal_x = self.image_width / (2.0 * self.tanfovx)
        focal_y = self.image_height / (2.0 * self.tanfovy)
        (color, abd) = _SSR.apply(self.image_width, self.image_height, focal_x, focal_y, self.radius,
                                  self.bias, self.thick, self.delta, self.step, self.start, normal, pos, rgb,
                                  albedo, roughness, metallic, F0)
        return (color, abd)
