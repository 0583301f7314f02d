"""Host-side glue around the operator: the reference's `render()` post-processing and one
stage-2 ("PBR + indirect") training step, restated so that bench.py and the tests exercise
the same operator sequence as the reference's train.py.

    render()            gaussian_renderer/__init__.py:30-220
    stage2_step()       train.py:266-422 (render -> pbr_shading -> Gaussian_SSR -> L1 ->
                        backward), without dataset / optimizer / densification
    srgb conversions    train.py:54-81

Everything numerical runs in the HIP library (rasterizer, filters, SSAO/SSR, shade) or in
plain torch elementwise ops on the GPU; nothing here touches the CPU oracle.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Dict, Optional

import torch
import torch.nn.functional as F

import gigs_lib  # noqa: E402  (importable once the package's __init__ has put this directory on sys.path)

from diff_gaussian_rasterization import (AsyncBinning, BinningOverflow, GaussianRasterizationSettings, GaussianRasterizer,
                                         Gaussian_SSR, OutputPool, _C as _ops, after_blend, filters)


def linear_to_srgb(linear: torch.Tensor) -> torch.Tensor:  # train.py:54-68
    eps = torch.finfo(torch.float32).eps
    srgb0 = 323 / 25 * linear
    srgb1 = (211 * torch.clamp(linear, min=eps) ** (5 / 12) - 11) / 200
    return torch.where(linear <= 0.0031308, srgb0, srgb1)


def srgb_to_linear(srgb: torch.Tensor) -> torch.Tensor:  # train.py:70-81
    linear0 = 25 / 323 * srgb
    linear1 = ((srgb + 0.055) / 1.055) ** 2.4
    return torch.where(srgb <= 0.04045, linear0, linear1)


def canonical_rays(cam: Dict, device) -> torch.Tensor:
    """scene/__init__.py:137-169 (pixel centres at +0.5, unlike the rasterizer)."""
    H, W = cam["image_height"], cam["image_width"]
    fx = W / (2.0 * cam["tanfovx"])
    fy = H / (2.0 * cam["tanfovy"])
    x, y = torch.meshgrid(torch.arange(W, device=device), torch.arange(H, device=device), indexing="xy")
    x = x.flatten()
    y = y.flatten()
    return F.pad(torch.stack([(x - W / 2 + 0.5) / fx, (y - H / 2 + 0.5) / fy], dim=-1), (0, 1), value=1.0)


def make_settings(cam: Dict, sh_degree: int, bg: torch.Tensor, gi: Dict, device, inference=False,
                  debug=False) -> GaussianRasterizationSettings:
    as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.as_tensor(a, device=device)  # noqa: E731
    return GaussianRasterizationSettings(
        image_height=int(cam["image_height"]), image_width=int(cam["image_width"]), tanfovx=cam["tanfovx"],
        tanfovy=cam["tanfovy"], radius=gi["radius"], bias=gi["bias"], thick=gi["thick"], delta=gi["delta"],
        step=gi["step"], start=gi["start"], bg=bg, scale_modifier=1.0, viewmatrix=as_t(cam["viewmatrix"]),
        projmatrix=as_t(cam["projmatrix"]), sh_degree=sh_degree, campos=as_t(cam["campos"]), prefiltered=False,
        debug=debug, inference=inference, argmax_depth=False)


def _collect_idle():
    """Destroy whatever hipGraph-owning garbage exists (WholeStepGraph <-> Stage2Step is a reference cycle: only a
    cyclic-GC pass frees it) NOW, with the device idle before and after -- never between a capture and its replays.
    Measured on ROCm 7.0 (round 3): a collector pass that destroyed a batch of older graph execs right before a new
    capture, with no synchronisation after it, left the NEW exec with a dead internal stream and its first replay
    crashed in hip::Graph::UpdateStreams."""
    import gc
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.synchronize()


def graphed(callable_, sample_args):
    """torch.cuda.make_graphed_callables with the garbage collector parked: a cyclic-GC pass that runs while
    the stream is capturing may destroy older HIP objects (graphs, events, pooled blocks), which HIP refuses
    during capture and aborts the process."""
    import gc
    _collect_idle()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        return torch.cuda.make_graphed_callables(callable_, sample_args, allow_unused_input=True)
    finally:
        if was_enabled:
            gc.enable()


def gbuffer_post(normal_map_from_depth: torch.Tensor, normal_map: torch.Tensor, out_normal_view: torch.Tensor,
                 viewmatrix: torch.Tensor, opacity_map: Optional[torch.Tensor] = None, pad_normal: bool = False):
    """The tensor post-processing of gaussian_renderer.render (:157-199): masks, the optional background padding
    of the two normal maps (pad_normal, :159-173; render.py:212 uses it), normalisation, 3x3 medians and the
    rotation of the shading normal into view space.  With pad_normal the thresholded opacity map is returned as a
    sixth value (the reference returns it in place of the raw one)."""
    normal_from_depth_mask = (normal_map_from_depth != 0).all(0)
    normal_mask = (normal_map != 0).all(0, keepdim=True)
    if pad_normal:
        if opacity_map is None:
            raise ValueError("pad_normal=True needs the opacity map")
        opacity_map = torch.where(opacity_map < 0.004, torch.zeros_like(opacity_map), opacity_map)  # filters out 1 / 255
        opacity_map = torch.where(opacity_map > 1.0 - 0.004, torch.ones_like(opacity_map), opacity_map)
        normal_bg = torch.tensor([0.0, 0.0, 1.0], device=normal_map.device)
        normal_map = normal_map * opacity_map + (1.0 - opacity_map) * normal_bg[:, None, None]
        mask_from_depth = (normal_map_from_depth == 0.0).all(0, keepdim=True).float()
        normal_map_from_depth = normal_map_from_depth * (1.0 - mask_from_depth) + mask_from_depth * normal_bg[:, None, None]
    normal_map_from_depth = torch.where(torch.norm(normal_map_from_depth, dim=0, keepdim=True) > 0,
                                        F.normalize(normal_map_from_depth, dim=0, p=2), normal_map_from_depth)
    normal_map = torch.where(torch.norm(normal_map, dim=0, keepdim=True) > 0, F.normalize(normal_map, dim=0, p=2),
                             normal_map)
    normal_map = filters.median_blur(normal_map[None, ...], (3, 3))[0]
    R = viewmatrix[:3, :3]
    normals_view = -(normal_map.permute(1, 2, 0) @ R).permute(2, 0, 1)
    out_normal_view = torch.where(torch.norm(out_normal_view, dim=0, keepdim=True) > 0,
                                  F.normalize(out_normal_view, dim=0, p=2), out_normal_view)
    out_normal_view = filters.median_blur(out_normal_view[None, ...], (3, 3))[0]
    if pad_normal:
        return normal_map_from_depth, normal_from_depth_mask, normals_view, normal_mask, out_normal_view, opacity_map
    return normal_map_from_depth, normal_from_depth_mask, normals_view, normal_mask, out_normal_view


class _GbufferPostFused(torch.autograd.Function):
    """gbuffer_post as two kernels forward (gigs_gbuffer_post, gigs_normalize_mask) and two backward
    (gigs_gbuffer_post_bwd): stage 1 differentiates through normal_map only (train.py:327-328)."""

    @staticmethod
    def forward(ctx, normal_map_from_depth, normal_map, out_normal_view, viewmatrix):
        import gigs_lib
        lib = gigs_lib.lib()
        if not normal_map.is_cuda:
            raise RuntimeError("gbuffer_post needs CUDA/HIP tensors: gigs-hip has no CPU path")
        f = lambda t: t.contiguous().float()  # noqa: E731
        nfd, nm, onv_in, vm = f(normal_map_from_depth), f(normal_map), f(out_normal_view), f(viewmatrix)
        _, H, W = nm.shape
        dev = nm.device
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
        nfd_out, normals_view, onv = new(3, H, W), new(3, H, W), new(3, H, W)
        nfd_mask = torch.empty((H, W), dtype=torch.uint8, device=dev)
        mask = torch.empty((1, H, W), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            s = torch.cuda.current_stream().cuda_stream
            gigs_lib.check(lib.gigs_normalize_mask(H, W, nfd.data_ptr(), nfd_out.data_ptr(), nfd_mask.data_ptr(), s),
                           "normalize_mask")
            gigs_lib.check(lib.gigs_gbuffer_post(H, W, nm.data_ptr(), onv_in.data_ptr(), vm.data_ptr(),
                                                 normals_view.data_ptr(), mask.data_ptr(), None, onv.data_ptr(), s),
                           "gbuffer_post")
        ctx.save_for_backward(nm, vm)
        # four of the five outputs carry no gradient: without this autograd hands the backward a zero tensor for each
        # (two [3,H,W] float fills and two bool fills, 0.05 ms per stage-1 iteration at 800x800)
        ctx.set_materialize_grads(False)
        nfd_mask_b, mask_b = nfd_mask.bool(), mask.bool()
        ctx.mark_non_differentiable(nfd_out, nfd_mask_b, mask_b, onv)
        return nfd_out, nfd_mask_b, normals_view, mask_b, onv

    @staticmethod
    def backward(ctx, _g_nfd, _g_nfd_mask, g_normals_view, _g_mask, _g_onv):
        import gigs_lib
        lib = gigs_lib.lib()
        nm, vm = ctx.saved_tensors
        _, H, W = nm.shape
        if g_normals_view is None:
            return None, None, None, None
        g = g_normals_view.contiguous().float()
        scratch, g_nm = torch.empty_like(nm), torch.empty_like(nm)
        with torch.cuda.device(nm.device):
            gigs_lib.check(lib.gigs_gbuffer_post_bwd(H, W, nm.data_ptr(), vm.data_ptr(), g.data_ptr(), scratch.data_ptr(),
                                                     g_nm.data_ptr(), torch.cuda.current_stream().cuda_stream),
                           "gbuffer_post_bwd")
        return None, g_nm, None, None


def gbuffer_post_fused(normal_map_from_depth: torch.Tensor, normal_map: torch.Tensor, out_normal_view: torch.Tensor,
                       viewmatrix: torch.Tensor):
    """Same five results as gbuffer_post; gradient flows to normal_map only (normal_map_from_depth and out_normal_view
    receive none from the reference's operator either: their incoming gradients are dropped)."""
    return _GbufferPostFused.apply(normal_map_from_depth, normal_map, out_normal_view, viewmatrix)


def rasterize(cam: Dict, g: Dict[str, torch.Tensor], sh_degree: int, bg: torch.Tensor, gi: Dict,
              inference: bool = False, derive_normal: bool = True, debug=False, means2D: Optional[torch.Tensor] = None):
    """The operator call of gaussian_renderer.render (:53-155): returns the raw 12-tuple + means2D.  `means2D`: a
    persistent all-zero [P, 3] leaf to use instead of a fresh one (its values are never read, only its gradient)."""
    means3D = g["means3D"]
    screenspace_points = means2D if means2D is not None else torch.zeros_like(means3D, requires_grad=True)
    st = make_settings(cam, sh_degree, bg, gi, means3D.device, inference=inference, debug=debug)
    out = GaussianRasterizer(st)(
        means3D=means3D, means2D=screenspace_points, opacities=g["opacities"], normal=g["normal"], shs=g["shs"],
        albedo=g["albedo"], roughness=g["roughness"], metallic=g["metallic"], scales=g["scales"],
        rotations=g["rotations"], derive_normal=derive_normal)
    return out, screenspace_points, st


def render(cam: Dict, g: Dict[str, torch.Tensor], sh_degree: int, bg: torch.Tensor, gi: Dict,
           inference: bool = False, derive_normal: bool = True, debug=False, fused_post: bool = False,
           pad_normal: bool = False) -> Dict[str, torch.Tensor]:
    """gaussian_renderer.render (:30-220); fused_post=True runs the pad_normal=False post-processing as fused kernels."""
    ((rendered_image, radii, opacity_map, depth_map, normal_map_from_depth, normal_map, occlusion_map, albedo_map,
      roughness_map, metallic_map, out_normal_view, depth_pos), screenspace_points, st) = rasterize(
        cam, g, sh_degree, bg, gi, inference=inference, derive_normal=derive_normal, debug=debug)
    if pad_normal:
        (normal_map_from_depth, normal_from_depth_mask, normals_view, normal_mask, out_normal_view, opacity_map) = gbuffer_post(
            normal_map_from_depth, normal_map, out_normal_view, st.viewmatrix, opacity_map=opacity_map, pad_normal=True)
    else:
        (normal_map_from_depth, normal_from_depth_mask, normals_view, normal_mask, out_normal_view) = (
            gbuffer_post_fused if fused_post else gbuffer_post)(normal_map_from_depth, normal_map, out_normal_view, st.viewmatrix)
    return {
        "render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
        "radii": radii, "opacity_map": opacity_map, "depth_map": depth_map,
        "normal_map_from_depth": normal_map_from_depth, "normal_from_depth_mask": normal_from_depth_mask,
        "normal_map": normals_view, "normal_mask": normal_mask, "albedo_map": albedo_map,
        "roughness_map": roughness_map, "metallic_map": metallic_map, "occlusion_map": occlusion_map,
        "out_normal_view": out_normal_view, "depth_pos": depth_pos,
    }


RASTER_KEYS = ("means3D", "opacities", "normal", "albedo", "roughness", "metallic", "shs", "scales", "rotations")


class RasterFront(torch.nn.Module):
    """The operator call of gaussian_renderer.render -- rasterizer, in-op filters, SSAO -- as a static-shape tensor
    function of the Gaussians and the camera matrices: what torch.cuda.make_graphed_callables needs.  It is capturable
    only under AsyncBinning (no host read-back, fixed binning capacity); image size, field of view and GI settings are
    baked into the capture, the camera pose is an input."""

    def __init__(self, H: int, W: int, tanfovx: float, tanfovy: float, gi: Dict, sh_degree: int, inference: bool = False):
        super().__init__()
        self.H, self.W, self.tanfovx, self.tanfovy = int(H), int(W), float(tanfovx), float(tanfovy)
        self.gi, self.sh_degree, self.inference = dict(gi), int(sh_degree), bool(inference)

    def forward(self, means2D, viewmatrix, projmatrix, campos, bg, means3D, opacities, normal, albedo, roughness, metallic,
                shs, scales, rotations):
        gi = self.gi
        st = GaussianRasterizationSettings(
            image_height=self.H, image_width=self.W, tanfovx=self.tanfovx, tanfovy=self.tanfovy, radius=gi["radius"],
            bias=gi["bias"], thick=gi["thick"], delta=gi["delta"], step=gi["step"], start=gi["start"], bg=bg,
            scale_modifier=1.0, viewmatrix=viewmatrix, projmatrix=projmatrix, sh_degree=self.sh_degree, campos=campos,
            prefiltered=False, debug=False, inference=self.inference, argmax_depth=False)
        return GaussianRasterizer(st)(means3D=means3D, means2D=means2D, opacities=opacities, normal=normal, shs=shs,
                                      albedo=albedo, roughness=roughness, metallic=metallic, scales=scales,
                                      rotations=rotations, derive_normal=True)


class DenseScene(RuntimeError):
    """Raised by the graph-capturing steppers only under long_lists = 0 (gigs_options; GIGS_LONG_LISTS=0) for a scene that averages more instances per
    tile than one workgroup sorts in LDS: the caller then keeps the synchronous path.  By default dense scenes (3 M
    Gaussians at the Mip-NeRF360 images_4 sizes: 7 000 instances per tile) are NOT declined: the library partitions
    their long tile lists by sampled splitters (csrc/binning.hip) and they take the asynchronous, captured path."""


def _declined_as_dense(probe: int, tiles: int) -> bool:
    c = gigs_lib.current()
    return probe > c.option("bucket_max_mean") * tiles and c.option("long_lists") == 0


class GraphedRaster:
    """RasterFront captured into a hipGraph (forward and backward) under AsyncBinning, with the overflow protocol:

        out = gr(cam, g, means2D, bg)     # replays the graph; queues a snapshot of the device-side instance counters
        ... the rest of the step ...
        gr.check()                         # waits for that snapshot only; raises BinningOverflow after growing the capacity

    The capacity starts at twice the instance count of a first, synchronous forward; memory, not time, scales with it."""

    def __init__(self, cam: Dict, g: Dict[str, torch.Tensor], gi: Dict, sh_degree: int, inference: bool = False,
                 capacity: Optional[int] = None):
        self.cfg = (int(cam["image_height"]), int(cam["image_width"]), float(cam["tanfovx"]), float(cam["tanfovy"]))
        self.gi, self.sh_degree, self.inference = dict(gi), int(sh_degree), bool(inference)
        self.dev = g["means3D"].device
        self.capacity = int(capacity) if capacity else 0
        self.fn = self.bin = None
        self.recaptures = 0

    def _probe(self, cam, g, bg) -> int:
        e = torch.Tensor([])
        H, W, tx, ty = self.cfg
        with torch.no_grad():
            res = _ops.rasterize_gaussians(bg, g["means3D"], e, g["opacities"], g["normal"], g["albedo"], g["roughness"],
                                           g["metallic"], g["scales"], g["rotations"], e, g["shs"], cam["campos"],
                                           cam["viewmatrix"], cam["projmatrix"], 1.0, tx, ty, H, W, self.sh_degree, False,
                                           False, self.inference, False)
        return int(res[0])

    def _args(self, cam, g, means2D, bg):
        return (means2D, cam["viewmatrix"], cam["projmatrix"], cam["campos"], bg, *[g[k] for k in RASTER_KEYS])

    def _capture(self, cam, g, means2D, bg):
        H_, W_ = self.cfg[0], self.cfg[1]
        tiles = ((H_ + 15) // 16) * ((W_ + 15) // 16)
        if self.capacity <= 0:
            probe = self._probe(cam, g, bg)
            if _declined_as_dense(probe, tiles):
                raise DenseScene(f"{probe} instances over {tiles} tiles")
            self.capacity = max(65536, -(-2 * probe // 65536) * 65536)
        self.bin = AsyncBinning(self.capacity, self.dev)
        H, W, tx, ty = self.cfg
        mod = RasterFront(H, W, tx, ty, self.gi, self.sh_degree, self.inference)
        # static inputs: the Gaussian tensors and means2D THEMSELVES (aliases: a replay then finds its inputs in place and
        # copies nothing -- 40 MB and ten launches per step otherwise); the small per-view camera tensors are copied in
        args = self._args(cam, g, means2D, bg)
        sample = tuple((a.detach() if i == 0 or i >= 5 else a.detach().clone()).requires_grad_(a.requires_grad)
                       for i, a in enumerate(args))
        with self.bin:
            if self.inference or not any(a.requires_grad for a in sample):
                self.fn = _graphed_inference(mod, sample)
            else:
                self.fn = graphed(mod, sample)
        self.recaptures += 1

    def __call__(self, cam, g, means2D, bg):
        if (int(cam["image_height"]), int(cam["image_width"]), float(cam["tanfovx"]), float(cam["tanfovy"])) != self.cfg:
            raise ValueError("GraphedRaster: image size / field of view differ from the captured ones")
        if self.fn is None:
            self._capture(cam, g, means2D, bg)
        out = self.fn(*self._args(cam, g, means2D, bg))
        self.bin.snapshot()
        return out

    def check(self) -> int:
        try:
            return self.bin.check()
        except BinningOverflow as ex:
            self.capacity = -(-int(1.5 * ex.needed) // 65536) * 65536
            self.close()  # next call re-captures with the larger buffer
            raise

    def close(self) -> None:
        """Release the captured callable with the device idle before and after (see WholeStepGraph._drop_graphs)."""
        if self.fn is not None:
            torch.cuda.synchronize(self.dev)
            self.fn = None
            torch.cuda.synchronize(self.dev)


def _graphed_inference(mod, sample):
    """A forward-only hipGraph of `mod` (torch.cuda.make_graphed_callables captures a backward and needs inputs that
    require grad): static inputs, three warm-up runs on a side stream, one capture, replay = copy inputs + launch."""
    import gc
    static_in = tuple(a.detach().clone() for a in sample)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(3):
            mod(*static_in)
    torch.cuda.current_stream().wait_stream(side)
    _collect_idle()
    was = gc.isenabled()
    gc.disable()
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            static_out = mod(*static_in)
    finally:
        if was:
            gc.enable()

    def run(*args):
        for s, a in zip(static_in, args):
            if s.data_ptr() != a.data_ptr():
                s.copy_(a)
        graph.replay()
        return static_out

    return run


class GeometryCache:
    """Per-view reuse of what frozen geometry makes constant in stage-2 training (secondary metric only; the headline step
    always runs everything).  A stage-2 iteration of train.py (:330-420) sends gradients to albedo / roughness / metallic
    and the light only, so once Adam's residual momentum has died out positions, covariances, opacities and normals stop
    changing bit for bit -- and with them, per view, the tile lists (`ranges`, `point_list`: the whole binning), the
    operator's occlusion plane (the SSAO march) and the indirect-light hit list.  288 GB of HBM hold them for a whole
    training set (~18 MB per 800 x 800 view, ~113 MB with the hit list of the C2 scene).

    Protocol.  The optimizer step reports whether it moved a bit of a geometry group (gigs_adam_step_watch -> `flag_dev`,
    copied to pinned memory by a node of the update graph).  A view is RECORDED by a complete step (dgr.view_cache: the
    forward bins into the slot's chunks, SSAO writes the slot's plane; `store` copies the parts to the view's entry),
    REPLAYED from its entry afterwards (`load` -> slot; preprocess + blend only, no binning, no SSAO march) -- the first
    replay still marches the indirect light and records its hit list ("replay_rec": a drifting phase, whose recordings are
    never kept, pays nothing for hit lists), later ones gather at the recorded hits ("replay").  The host
    learns about update k only while step k + 1 runs, so a replay is optimistic: if the flag read after the forward says
    the previous update moved geometry, every entry is dropped and the step is repeated as a recording before anything of
    it is consumed (the update graph has not been replayed yet) -- the same repeat protocol as a binning overflow.
    Entries are also dropped when the parameter tensors are replaced (densification) or modified behind the library's
    back (tensor version counters)."""

    WATCH = ("xyz", "scaling", "rotation", "opacity", "normal")              # raw (pre-activation) group names
    WATCH_ACTIVATED = ("means3D", "scales", "rotations", "opacities", "normal")  # the same for post-activation dictionaries

    def __init__(self, device):
        from diff_gaussian_rasterization import ViewSlot
        self.device = torch.device(device)
        self.slot = ViewSlot(self.device)
        self.entries: Dict = {}
        self.holds = None          # view key whose state the slot currently holds
        self.param_key = None
        self.capacity = 0
        self.flag_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.stats = dict(recorded=0, replayed=0, repeated=0, invalidated=0, hit_lists=0)
        self.recorded_R: Dict = {}
        # the hit list of the indirect-light march is recorded by the default march only (gigs_ssr_hits)
        self.hit_lists = os.environ.get("GIGS_SSR_HIT_LIST", "1") == "1" and gigs_lib.current().option("gi_march") == 4

    @staticmethod
    def view_key(cam: Dict):
        for k in ("uid", "index"):
            if k in cam:
                return (k, int(cam[k]), int(cam["image_width"]), int(cam["image_height"]))
        vm = cam["viewmatrix"]
        return ("ptr", vm.data_ptr(), vm._version, int(cam["image_width"]), int(cam["image_height"]))

    def _geometry_key(self, g: Dict[str, torch.Tensor]):
        names = self.WATCH if "xyz" in g else self.WATCH_ACTIVATED
        return tuple((n, g[n].data_ptr(), tuple(g[n].shape), g[n]._version) for n in names if n in g)

    def invalidate(self) -> None:
        if self.entries:
            self.stats["invalidated"] += 1
        self.entries.clear()
        self.holds = None

    def mode(self, vkey, g) -> str:
        """"record" (no entry: bin, march SSAO, store), "replay_rec" (entry without a hit list: reuse tile lists and
        occlusion, march the indirect light once more and record its hits), "replay" (reuse everything)."""
        k = self._geometry_key(g)
        if k != self.param_key:   # new tensors (densification) or an in-place edit outside the captured update
            self.invalidate()
            self.param_key = k
        e = self.entries.get(vkey)
        if e is None:
            return "record"
        return "replay_rec" if (self.hit_lists and "ssr_offsets" not in e) else "replay"

    def load(self, vkey) -> None:
        if self.holds != vkey:
            self.slot.load(self.entries[vkey])
            self.holds = vkey

    def store(self, vkey) -> None:
        self.entries[vkey] = self.slot.export()
        self.holds = vkey
        self.stats["recorded"] += 1

    def store_hits(self, vkey) -> bool:
        """After a "replay_rec" forward has finished: keep the view's hit list -- or, if it did not fit the buffer, enlarge
        the buffer (the variants that use it re-capture: its address is part of their key) and try again at the next visit."""
        slot = self.slot
        if slot.ssr_ok():
            self.entries[vkey].update(slot.export_hits())
            slot.ssr_loaded = True
            self.stats["hit_lists"] += 1
            return True
        torch.cuda.synchronize(self.device)
        slot.ssr_prepare(slot.ssr_counts.numel() // 4, total_hint=int(slot.ssr_total_host[0]))
        return False

    def geometry_moved(self) -> bool:
        """What the LAST completed update reported (valid once a later forward of the same stream has finished)."""
        return int(self.flag_host[0]) != 0


class WholeStepGraph:
    """One stage-2 iteration -- rasterizer, SSAO, light filter, shade, SSR, loss, and the whole backward -- captured by
    hand into TWO hipGraphs (forward, backward) that share one memory pool:

        replay(forward); record(event); replay(backward); wait(event); read the binning counters

    Compared with chaining torch.cuda.make_graphed_callables pieces (GIGS_RASTER_GRAPH=1: six graph launches) there are
    no staging copies of activations or of incoming gradients, no zero-filled placeholder gradients, and the ~45 kernel
    nodes of a step run without the 11-18 us the command processor spends between eager launches; compared with the
    eager rasterizer the light's filter still starts when the blend kernel starts (an event node recorded by the
    library inside the forward), not beside the single-workgroup scan kernels it would starve.

    Static inputs: the Gaussian tensors and light.base ARE the graph's inputs (aliases: an optimizer that updates them
    in place needs no copy; tensors replaced by densification trigger a re-capture); the camera matrices, view_dirs
    and gt_image are copied into fixed buffers before the replay (skipped when the caller passes the captured tensor).
    Gradients come from torch.autograd.grad inside the backward capture (no AccumulateGrad nodes, which would run on
    the parameters' creation stream) and are handed out as `.grad` after the replay: assigned when `.grad` is None
    (the graph's static output buffer itself, no copy), added otherwise -- the semantics of loss.backward().  A caller
    that accumulates over views without clearing still holds the previous hand-out, which the next replay overwrites:
    such a `.grad` is moved into a tensor of its own before the replay (one clone, only in that case).  With dp.GradSlab's sink active during the capture the
    rasterizer's backward writes straight into the slab, as in the eager step.

    Binning runs under AsyncBinning (fixed capacity, nothing read back inside the step); the counters are copied to
    pinned memory by a node at the end of the forward graph and examined after the backward has been queued, so the
    host waits for the forward only.  On overflow the capacity grows, both graphs are re-captured and the step is
    repeated; its gradients are never handed out."""

    def __init__(self, owner: "Stage2Step", cam: Dict, g: Dict[str, torch.Tensor], cache: Optional["GeometryCache"] = None,
                 mode: Optional[str] = None):
        self.cache, self.mode = cache, mode  # frozen-geometry variant: "record" / "replay_rec" / "replay" (dgr.view_cache)
        # a weak reference: the stepper owns its WholeStepGraphs, not the other way round -- no reference cycle, so the
        # graph execs die where the code says (close(), or the owner's last reference going away), never "whenever the
        # cyclic collector happens to run" (the round-3 host segfault in hip::Graph::UpdateStreams: DESIGN.md section 5)
        self.owner = weakref.proxy(owner)
        self.dev = next(iter(g.values())).device
        self.cfg = (int(cam["image_height"]), int(cam["image_width"]), float(cam["tanfovx"]), float(cam["tanfovy"]))
        self.capacity = 0
        self.gf = self.gb = self.go = self.key = self.adam = None
        self.res = self.grads = self.vp_grad = self.inner = self.bin = None
        self._src = {}
        self.recaptures = 0
        self.fwd_done = torch.cuda.Event()
        self._packs = {}
        self.viol_dev = self.viol_host = None  # declared stage-2 gradient set: violation counter (device) and its pinned copy
        self.split_rest = None                 # ... and the optimizer's f_rest tensor the rasterizer then reads directly

    CAM_TENSORS = ("viewmatrix", "projmatrix", "campos")

    def _drop_graphs(self):
        """Release the three graph execs and what only they keep alive, with the device idle before AND after: an exec is
        never destroyed while one of its replays may still be running, and nothing is captured or replayed before the
        destruction has completed."""
        if self.gf is None and self.gb is None and self.go is None:
            return
        torch.cuda.synchronize(self.dev)
        self.gf = self.gb = self.go = self.adam = self.key = None
        self.res = self.grads = self.vp_grad = None
        torch.cuda.synchronize(self.dev)

    def close(self):
        """Deterministic teardown: the graphs, the captured eager step, the binning buffer and the per-camera caches.  The
        object captures again if it is called afterwards."""
        self._drop_graphs()
        self.inner = self.bin = None
        self._packs.clear()
        self._src = {}
        self.check_declared()  # the device is idle: the last update's count has arrived

    def __del__(self):
        try:
            self._drop_graphs()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def _params(self, g):
        light = getattr(self.owner, "light", None)
        return [g[k] for k in g] + ([p for p in light.parameters()] if light is not None else [])

    def _key(self, g):
        import diff_gaussian_rasterization as dgr
        sink = dgr._st.grad_sink or {}
        slabv = getattr(self.owner, "grad_slab", None) or {}
        slot = self.cache.slot if self.cache is not None else None
        slot_key = None if slot is None else tuple(None if t is None else t.data_ptr() for t in (
            (slot.occlusion,) if self.mode == "record" else (slot.occlusion, slot.ssr_offsets, slot.ssr_entries)))
        return (slot_key, tuple((t.data_ptr(), tuple(t.shape)) for t in self._params(g)),
                tuple(sorted((k, v.data_ptr()) for k, v in sink.items())),
                tuple(sorted((k, v.data_ptr()) for k, v in slabv.items())),
                self.adam.key() if self.adam is not None else None)

    def _slab_sinks(self):
        """The owner's gradient slab (train_iteration: data_parallel) as gradient sinks of the captured step: the raw
        gradients are written into the slab's views by the activations' backward, xyz's by the rasterizer's.  Also enters
        the frozen-geometry view cache of this variant."""
        import contextlib
        slabv = getattr(self.owner, "grad_slab", None)
        stack = contextlib.ExitStack()
        if self.cache is not None:
            from diff_gaussian_rasterization import view_cache
            stack.enter_context(view_cache(self.cache.slot, self.mode))
        if slabv:
            import activations
            import diff_gaussian_rasterization as dgr
            stack.enter_context(activations.grad_sink({k: v for k, v in slabv.items() if k in activations.RAW}))
            if "xyz" in slabv:
                stack.enter_context(dgr.grad_sink({"means3D": slabv["xyz"]}))
        if self.viol_dev is not None:
            # the declared stage-2 gradient set: the forward's library context carries the violation counter, the backward
            # nodes (rasterizer, activations) then produce the material gradients only
            stack.enter_context(gigs_lib.use(gigs_lib.current().derive(materials_only=self.viol_dev, sh_rest=self.split_rest)))
        return stack

    def check_declared(self) -> None:
        """The declared stage-2 gradient set (Stage2Step.materials_only) is checked on the device by every backward; the
        count reaches pinned memory behind every update.  Non-zero: a gradient that was taken for zero was not."""
        if self.viol_host is not None and int(self.viol_host[0]) != 0:
            n = int(self.viol_host[0])
            self.viol_host.zero_()
            raise RuntimeError(
                "WholeStepGraph: %d wave(s) of the rasterizer's backward found a gradient outside the declared stage-2 set "
                "(albedo / roughness / metallic / light) -- the loss reaches colour, opacity, depth, normal or geometry.  "
                "The updates since then were withheld on the device (gigs_adam_step_guarded), the parameters are those of "
                "the last valid step.  Build the stepper / trainer with materials_only=False (GIGS_MATERIALS_ONLY=0)." % n)

    def _into_slab(self, g, grads):
        """Inside the backward capture: every gradient ends up in its slab view (most were born there; the light's is
        copied by one node), and the views are what the captured Adam launch reads."""
        slabv = getattr(self.owner, "grad_slab", None)
        if not slabv:
            return list(grads)
        names = list(g.keys()) + (["cubemap"] if getattr(self.owner, "light", None) is not None else [])
        out = list(grads)
        with torch.no_grad():
            for i, name in enumerate(names):
                v = slabv.get(name)
                if v is None:
                    continue
                if out[i] is None:
                    # the declared stage-2 gradient set: an absent gradient outside the collective's stretch stays absent
                    # (Adam takes it as g = 0); inside it the slab must hold the zeros the other ranks add to
                    reduced = getattr(self.owner, "grad_slab_reduced", None)
                    if self.viol_dev is not None and reduced is not None and name not in reduced:
                        continue
                    v.zero_()
                elif out[i].data_ptr() != v.data_ptr():
                    v.copy_(out[i])
                out[i] = v
        return out

    def _capture(self, cam, g, gt_image, view_dirs):
        import gc
        o = self.owner
        H, W, _, _ = self.cfg
        bg = torch.zeros(3, device=self.dev)
        prep = o.prepare if o.prepare is not None else (lambda raw: raw)
        with torch.no_grad():
            ga = prep(g)  # the rasterizer's inputs (for the probe and the shapes); the capture re-derives them
        if self.cache is not None and self.cache.capacity > self.capacity:
            self.capacity = self.cache.capacity  # the record and the replay variant share the slot's chunks
        if self.capacity <= 0:
            probe = GraphedRaster(cam, ga, o.gi, o.sh_degree)._probe(cam, ga, bg)
            tiles = ((H + 15) // 16) * ((W + 15) // 16)
            if _declined_as_dense(probe, tiles):
                raise DenseScene(f"{probe} instances over {tiles} tiles")
            self.capacity = max(65536, -(-2 * probe // 65536) * 65536)
        if self.cache is not None:
            if self.cache.capacity != self.capacity:
                self.cache.invalidate()  # entries are carved for the old capacity
            self.cache.capacity = self.capacity
        self.bin = AsyncBinning(self.capacity, self.dev)
        # Declared stage-2 gradient set (complete iterations only: the gradients are consumed inside the step, by an Adam
        # launch that takes an absent gradient as g = 0)
        self.viol_dev = self.viol_host = self.split_rest = None
        if o.optimizers and getattr(o, "materials_only", False) and os.environ.get("GIGS_MATERIALS_ONLY", "1") == "1":
            self.viol_dev = torch.zeros(1, dtype=torch.int32, device=self.dev)
            self.viol_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            # with it, the SH block is never concatenated: the rasterizer reads the optimizer's two tensors (gigs_ctx_set_split_sh;
            # its materials-only backward does not touch SH)
            rest = g.get("f_rest") if o.prepare is not None else None
            if (rest is not None and rest.dim() == 3 and rest.shape[1] > 0 and rest.is_contiguous()
                    and os.environ.get("GIGS_SPLIT_SH", "1") == "1"):
                self.split_rest = rest
        self.inner = o._make_inner()  # the eager step that is captured: returns its attached loss instead of differentiating
        self.inner._defer_backward = True
        self.inner._static_bg = bg
        self.inner._static_m2d = torch.zeros_like(ga["means3D"], requires_grad=True)
        del ga
        # the three camera tensors are views of ONE static buffer: one copy per step
        self.s_pack = self._pack(cam).clone()
        self.s_cam = dict(cam)
        off = 0
        for k in self.CAM_TENSORS:
            n = cam[k].numel()
            self.s_cam[k] = self.s_pack[off:off + n].view(cam[k].shape)
            off += n
        self.s_vd, self.s_gt = view_dirs.detach().clone(), gt_image.detach().clone()
        self._src = {}  # static buffer -> (data_ptr, version) of the tensor it was last filled from
        params = self._params(g)
        self._drop_graphs()
        _collect_idle()
        # warm-up on a side stream: builds every cached table / library buffer outside the capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                with self._slab_sinks():
                    with self.bin:
                        res = self.inner(self.s_cam, prep(g), self.s_gt, self.s_vd)
                    torch.autograd.grad(res.pop("_loss"), params + [res["viewspace_points"]], allow_unused=True)
                del res
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if o.optimizers:
            from optim import CapturedAdam
            CapturedAdam.warmup_device(self.dev)
            if o.post_update is not None:
                # warm the clamp kernel on a scratch tensor, not on the live parameter: the reference clamps only AFTER an
                # optimizer step (train.py:523), so a light handed in with negative texels must reach its first forward
                # unchanged on this path as on the eager one
                torch.zeros(8, device=self.dev).clamp_(min=0.0)
            torch.cuda.synchronize()
        self._seed = torch.ones((), dtype=torch.float32, device=self.dev)
        _collect_idle()
        was = gc.isenabled()
        gc.disable()  # see graphed(): a cyclic-GC pass during capture may destroy HIP objects, which HIP refuses
        try:
            gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            # GIGS_STEP_GRAPH_PRIO=1 (diagnostic): capture the main branch on a high-priority stream.  Measured: the graph
            # executor does not turn that into dispatch order (blend backward still 0.54 ms), see the delay node below
            prio = os.environ.get("GIGS_STEP_GRAPH_PRIO", "0") == "1"
            cap = torch.cuda.Stream(device=self.dev, priority=-1) if prio else None
            # thread_local: a re-capture (binning overflow) may happen while RCCL's proxy thread is alive and issuing HIP
            # calls of its own, which the default (global) capture mode turns into a capture failure
            with torch.cuda.graph(gf, stream=cap, capture_error_mode="thread_local"):
                with self.bin, self._slab_sinks():
                    res = self.inner(self.s_cam, prep(g), self.s_gt, self.s_vd)
                self.bin.host.copy_(self.bin.counters, non_blocking=True)
            loss = res.pop("_loss")
            # The blend backward and the light's GGX backward become ready together (after the shade backward).  The
            # former lasts as long as its longest tiles and must get them resident at once; the latter floods every CU
            # with bandwidth-bound workgroups.  Started in that order (as eager launches happen to be: the side stream's
            # event wait costs ~35 us) the blend backward takes 0.30 ms, the other way round 0.51: a delay node gives it
            # the head start (GIGS_LIGHT_BWD_HEAD_START_US, default 10).
            import pbr.renderutils.ops as light_ops
            light_ops.bwd_head_start_ns = int(1e3 * float(os.environ.get("GIGS_LIGHT_BWD_HEAD_START_US", "10")))
            try:
                with torch.cuda.graph(gb, pool=gf.pool(), stream=cap, capture_error_mode="thread_local"):
                    # the seed gradient as a persistent tensor: autograd's own ones_like(loss) is a fill node at the head of
                    # the backward graph
                    with self._slab_sinks():
                        grads = torch.autograd.grad(loss, params + [res["viewspace_points"]],
                                                    grad_outputs=self._seed.reshape(loss.shape), allow_unused=True)
                    grads = self._into_slab(g, grads)
            finally:
                light_ops.bwd_head_start_ns = 0
            del loss
            go = adam = None
            if o.optimizers:
                # the update as a third graph (train.py:517-522): it is replayed only once the host has seen that the
                # forward's binning did not overflow, and -- multi-GPU -- after the gradient all-reduce
                from optim import CapturedAdam
                adam = CapturedAdam(o.optimizers, params, list(grads[:-1]), absent_is_zero=self.viol_dev is not None)
                go = torch.cuda.CUDAGraph()
                with torch.cuda.graph(go, pool=gf.pool(), stream=cap, capture_error_mode="thread_local"):
                    if self.cache is not None:
                        # the update reports whether it moved a geometry bit; the word travels to pinned memory and is cleared
                        adam.launch(watch=GeometryCache.WATCH, changed=self.cache.flag_dev, guard=self.viol_dev)
                        self.cache.flag_host.copy_(self.cache.flag_dev, non_blocking=True)
                        self.cache.flag_dev.zero_()
                    else:
                        adam.launch(guard=self.viol_dev)
                    if self.viol_dev is not None:
                        self.viol_host.copy_(self.viol_dev, non_blocking=True)  # cumulative: a violation stays visible
                    if o.post_update is not None:
                        with torch.no_grad():
                            o.post_update()
        finally:
            if was:
                gc.enable()
        self.gf, self.gb, self.go, self.adam, self.res = gf, gb, go, adam, res
        self.grads, self.vp_grad = list(grads[:-1]), grads[-1]
        self.key = self._key(g)
        self.recaptures += 1

    def _pack(self, cam):
        # (viewmatrix, projmatrix, campos) flattened into one tensor, built once per camera (training loops revisit them).
        # The cache entry keeps the source tensors alive: a key of (address, version) alone could be met again by a NEW
        # tensor that the allocator placed at a freed one's address.
        src = tuple(cam[k] for k in self.CAM_TENSORS)
        key = tuple((t.data_ptr(), t._version) for t in src)
        hit = self._packs.get(key)
        if hit is None or any(a is not b for a, b in zip(hit[1], src)):
            if len(self._packs) >= 4096:
                self._packs.clear()
            hit = (torch.cat([t.detach().reshape(-1).float() for t in src]), src)
            self._packs[key] = hit
        return hit[0]

    def _fill(self, name, static, src):
        # per-view constants come back every n_views steps and gt_image is often one tensor: skip what is already there
        # (the reference to the last source keeps its address from being reused by a different tensor)
        last = self._src.get(name)
        if last is None or last[0] is not src or last[1] != src._version:
            static.copy_(src, non_blocking=True)
            self._src[name] = (src, src._version)

    def __call__(self, cam, g, gt_image, view_dirs, vkey=None):
        """Returns None (frozen-geometry variants only) when the step must be repeated as a recording: the previous update
        moved geometry, which the host can only know once this step's forward has run."""
        if (int(cam["image_height"]), int(cam["image_width"]), float(cam["tanfovx"]), float(cam["tanfovy"])) != self.cfg:
            raise ValueError("WholeStepGraph: image size / field of view differ from the captured ones")
        cache = self.cache
        for _ in range(4):
            if cache is not None and self.mode == "record":
                cache.holds = None  # this forward overwrites the slot, whether or not its view ends up stored
            if cache is not None and self.mode != "record":
                if self.capacity <= 0:
                    self.capacity = cache.capacity  # a fresh variant adopts the chunks' capacity
                if cache.capacity != self.capacity or vkey not in cache.entries:
                    return None  # the recording variant re-sized the chunks / dropped the entry: record again
                cache.load(vkey)  # the view's tile lists and occlusion plane into the slot (a no-op for the same view)
            if self.gf is None or self.key != self._key(g):
                self._capture(cam, g, gt_image, view_dirs)
            self._fill("camera", self.s_pack, self._pack(cam))
            self._fill("view_dirs", self.s_vd, view_dirs)
            self._fill("gt_image", self.s_gt, gt_image)
            # gradient accumulation over views: a parameter that still holds the buffer handed out by the previous step
            # (the caller did not clear it) would see that buffer overwritten by this replay -- keep the old values in a
            # tensor of its own, the hand-out below then adds to it
            for p, gr in zip(self._params(g), self.grads):
                if gr is not None and p.grad is not None and p.grad.data_ptr() == gr.data_ptr():
                    p.grad = p.grad.clone()
            self.gf.replay()
            self.fwd_done.record()
            self.gb.replay()
            self.fwd_done.synchronize()
            self.check_declared()  # this forward has ended, so every earlier update's count has arrived
            r, over = int(self.bin.host[0]), int(self.bin.host[1])
            if over:
                self.capacity = -(-int(1.5 * over) // 65536) * 65536
                if cache is not None:
                    cache.invalidate()
                    cache.capacity = self.capacity
                self._drop_graphs()  # waits for the backward replay that is still running, then releases the execs
                continue
            if cache is not None:
                # this forward has finished, so the previous update has too: did it move geometry?
                moved = cache.geometry_moved()
                if moved:
                    cache.flag_host.zero_()
                    cache.invalidate()
                    if self.mode != "record":
                        cache.stats["repeated"] += 1
                        return None  # the replayed lists were stale: nothing of this step has been consumed, repeat it
                if self.mode == "record":
                    if not moved:  # while geometry still drifts every entry would be dropped at the next step anyway
                        cache.store(vkey)
                        cache.recorded_R[vkey] = r
                else:
                    if self.mode == "replay_rec":
                        cache.store_hits(vkey)
                    cache.stats["replayed"] += 1
                    r = cache.recorded_R.get(vkey, r)
            params = self._params(g)
            if self.go is not None:
                # complete iteration: all-reduce (multi-GPU) -> Adam + clamp from the third graph; the gradients are
                # consumed inside the step, the parameters keep .grad = None (zero_grad(set_to_none=True), train.py:518)
                if self.owner.before_update is not None:
                    self.owner.before_update()
                self.adam.advance()
                self.go.replay()
            else:
                for p, gr in zip(params, self.grads):
                    if gr is None:
                        continue
                    if p.grad is None:
                        p.grad = gr  # the graph's static output itself (or the all-reduce slab's view under grad_sink)
                    else:
                        p.grad.add_(gr)
            vp = self.res["viewspace_points"]
            vp.grad = self.vp_grad
            out = dict(self.res)
            out["num_rendered"] = r
            return out
        raise RuntimeError("WholeStepGraph: the binning capacity kept overflowing")


def view_dirs_for(cam: Dict, rays: torch.Tensor, device) -> torch.Tensor:
    """train.py:299-308."""
    H, W = cam["image_height"], cam["image_width"]
    vm = cam["viewmatrix"] if isinstance(cam["viewmatrix"], torch.Tensor) else torch.as_tensor(cam["viewmatrix"], device=device)
    c2w = torch.inverse(vm.T)
    return -((F.normalize(rays[:, None, :], p=2, dim=-1) * c2w[None, :3, :3]).sum(dim=-1).reshape(H, W, 3))


class Stage2Front(torch.nn.Module):
    """train.py:293-379 between the rasterizer and Gaussian_SSR as ONE tensor function: G-buffer
    post-processing, roughness remap, build_mips, pbr_shading, F0 and the linearised direct light.
    All shapes are fixed by the resolution, so the module can be captured into a hipGraph."""

    def __init__(self, light, brdf_lut: torch.Tensor, metallic: bool = True, indirect: bool = True,
                 tone: bool = False, gamma: bool = False):
        super().__init__()
        self.light = light
        self.register_buffer("brdf_lut", brdf_lut, persistent=False)
        self.metallic, self.indirect, self.tone, self.gamma = metallic, indirect, tone, gamma

    def forward(self, normal_map_from_depth, normal_map_raw, out_normal_view_raw, albedo_map, roughness_raw,
                metallic_map, occlusion_map, viewmatrix, view_dirs):
        from pbr import pbr_shading  # HIP-backed drop-in of the reference's pbr package
        (_, _, normals_view, normal_mask, out_normal_view) = gbuffer_post(
            normal_map_from_depth, normal_map_raw, out_normal_view_raw, viewmatrix)
        roughness_map = roughness_raw * (1.0 - 0.04) + 0.04  # train.py:293-295
        occlusion = (occlusion_map if self.indirect else torch.ones_like(roughness_map)).permute(1, 2, 0)
        self.light.build_mips()
        pbr_result = pbr_shading(light=self.light, normals=normals_view.permute(1, 2, 0).detach(), view_dirs=view_dirs,
                                 mask=normal_mask.permute(1, 2, 0), albedo=albedo_map.permute(1, 2, 0),
                                 roughness=roughness_map.permute(1, 2, 0),
                                 metallic=metallic_map.permute(1, 2, 0) if self.metallic else None, tone=self.tone,
                                 gamma=self.gamma, occlusion=occlusion.detach(), brdf_lut=self.brdf_lut)
        render_direct = pbr_result["render_rgb"].permute(2, 0, 1)
        render_direct = torch.where(normal_mask, render_direct, torch.zeros_like(render_direct))  # black background
        if self.metallic:
            F0 = (1.0 - metallic_map) * 0.04 + albedo_map * metallic_map
            metallic_out = metallic_map
        else:
            F0 = torch.ones_like(albedo_map) * 0.04
            metallic_out = torch.zeros_like(roughness_map)
        linear_rgb = srgb_to_linear(render_direct).detach()
        return (render_direct, roughness_map, metallic_out, F0, linear_rgb, out_normal_view.detach(),
                normal_mask.to(roughness_map.dtype))


def stage2_loss(render_direct, IRR_linear, gt_image, normal_mask_f, roughness_map, metallic_map):
    """train.py:382-402: IRR to sRGB, 3x3 median, L1 against the ground truth, 'lamb' regulariser."""
    IRR = linear_to_srgb(IRR_linear)
    IRR = filters.median_blur(IRR[None, ...], (3, 3))[0]
    render_rgb = render_direct + IRR
    loss = torch.abs(render_rgb - gt_image).mean()  # utils/loss_utils.l1_loss
    # train.py:401-402 writes (1 - roughness_map[normal_mask]).mean() + metallic_map[normal_mask].mean();
    # boolean indexing synchronises with the host (it needs the element count), the masked sums do not
    cnt = normal_mask_f.sum()
    lamb_loss = ((1.0 - roughness_map) * normal_mask_f).sum() / cnt + (metallic_map * normal_mask_f).sum() / cnt
    return loss + lamb_loss * 0.001, render_rgb.detach()


class Stage2Step:
    """One stage-2 iteration of train.py (:266-422) up to and including loss.backward().

    graphs=True captures the two launch-bound glue segments (front: ~120 small kernels + build_mips
    + shade, and the loss) forward AND backward into hipGraphs with torch.cuda.make_graphed_callables;
    the rasterizer (data-dependent sizes, one host read) and SSR stay eager so their kernels are timed
    individually.  Results are identical to graphs=False (same kernels, same order)."""

    def __init__(self, light, brdf_lut, gi: Dict, sh_degree: int, metallic: bool = True, indirect: bool = True,
                 gamma: bool = False, tone: bool = False, graphs: bool = False, fused: bool = False,
                 prepare=None, regularizer=None, optimizers=None, post_update=None, before_update=None,
                 geometry_cache: bool = False, materials_only: bool = False):
        """The last five arguments turn the step into a COMPLETE training iteration (train_iteration.Stage2Trainer):
        `prepare(raw) -> g` maps the optimizer's tensors to the rasterizer's inputs (the GaussianModel getters;
        `__call__` then takes the raw dictionary), `regularizer(maps) -> scalar` adds the BRDF / envmap terms of
        train.py:387-420 (maps: normal_map, albedo_map, roughness_map, metallic_map -- the rasterizer's planes --
        and gt_image), `optimizers` (FusedAdam) are stepped after the backward, `before_update()` runs between the two
        (the multi-GPU gradient all-reduce) and `post_update()` after (cubemap.clamp_, train.py:522).  All of it is
        part of the captured step when graphs=True (pipeline.WholeStepGraph)."""
        self.prepare, self.regularizer, self.optimizers = prepare, regularizer, list(optimizers or [])
        self.post_update, self.before_update = post_update, before_update
        self.gi, self.sh_degree, self.metallic = gi, sh_degree, metallic
        self.fused, self.light, self.brdf_lut = fused, light, brdf_lut
        self.flags = dict(metallic=metallic, indirect=indirect, gamma=gamma, tone=tone)
        self.back = self.mips = self.side = self.step_begin = self.blend_begin = self.graster = None
        # fused + graphs: the rasterizer's planes live at fixed addresses, the graph reads them in place
        self.pool = OutputPool() if (fused and graphs and os.environ.get("GIGS_OUTPUT_POOL", "1") == "1") else None
        self.front = Stage2Front(light, brdf_lut, metallic=metallic, indirect=indirect, tone=tone, gamma=gamma)
        self.loss_fn = stage2_loss
        self.graphs = graphs
        self._captured = False
        self._defer_backward = False  # WholeStepGraph's inner step: return the attached loss, the caller differentiates
        self.whole, self._wholes = None, {}
        # frozen-geometry reuse (GeometryCache; graph path only, off by default: the headline step never uses it)
        self.geometry_cache, self.geom_cache = bool(geometry_cache), None
        self.materials_only = bool(materials_only)

    def _capture(self, front_args, loss_args):
        def clone(args):
            return tuple(a.detach().clone().requires_grad_(a.requires_grad) for a in args)
        self.front = graphed(self.front, clone(front_args))
        self.loss_fn = graphed(self.loss_fn, clone(loss_args))
        self._captured = True

    def __call__(self, cam: Dict, g: Dict[str, torch.Tensor], gt_image: torch.Tensor, view_dirs: torch.Tensor,
                 extra_loss=None):
        """extra_loss(normal_map, albedo_map, roughness_map, metallic_map) -> scalar added to the loss before
        backward (the BRDF / envmap regularisers of train.py:387-420; see gi-gs_amd/losses.py)."""
        raw = g
        if (self.fused and self.graphs and extra_loss is None and os.environ.get("GIGS_STEP_GRAPH", "1") == "1"
                and os.environ.get("GIGS_RASTER_GRAPH", "0") != "1" and not getattr(self, "_dense", False)):
            try:
                # one capture per (image size, field of view): datasets with per-camera intrinsics keep a few of them
                cfg = (int(cam["image_height"]), int(cam["image_width"]), float(cam["tanfovx"]), float(cam["tanfovy"]))
                if self.geometry_cache:
                    out = self._cached_geometry_step(cfg, cam, raw, gt_image, view_dirs)
                    if out is not None:
                        return out
                elif self.whole is None or self.whole.cfg != cfg:
                    self.whole = self._wholes.get(cfg)
                    if self.whole is None and len(self._wholes) < 4:
                        self.whole = self._wholes[cfg] = WholeStepGraph(self, cam, g)
                if self.whole is not None and not self.geometry_cache:
                    return self.whole(cam, raw, gt_image, view_dirs)
                # more than four distinct camera models: the fifth onwards takes the piecewise path below
            except DenseScene:
                self._dense = True  # synchronous binning with the global radix sort: the rasterizer stays eager
                for w in self._wholes.values():
                    w.close()
                self.whole = None
                self._wholes.clear()
        regen = (lambda: self.prepare(raw)) if self.prepare is not None else None
        if regen is not None:
            g = regen()
        res = self._step(cam, g, gt_image, view_dirs, extra_loss, regen)
        if self.optimizers:  # complete iteration, eager formulation: train.py:517-522
            if self.before_update is not None:
                self.before_update()
            for o in self.optimizers:
                o.step()
            for leaf in self._leaves(raw):
                leaf.grad = None  # zero_grad(set_to_none=True)
            if self.post_update is not None:
                with torch.no_grad():
                    self.post_update()
        return res

    def _step(self, cam, g, gt_image, view_dirs, extra_loss=None, regen=None):
        """Everything but WholeStepGraph: the chained-graph formulation (GIGS_RASTER_GRAPH=1) and the eager ones.
        `regen()` rebuilds the rasterizer's inputs from the raw parameters for a repeated step (binning overflow)."""
        dev = g["means3D"].device
        if self.fused and self.graphs and os.environ.get("GIGS_RASTER_GRAPH", "0") == "1" and not getattr(self, "_dense", False):
            try:
                return self._graphed_step(cam, g, gt_image, view_dirs, extra_loss)
            except DenseScene:
                self._dense = True  # keep the rasterizer eager (synchronous binning, global radix sort); the rest stays graphed
                self.step_begin = None
        # train.py:263-264: black background for PBR (WholeStepGraph's inner step keeps it, and means2D, outside the graph)
        background = getattr(self, "_static_bg", None)
        if background is None:
            background = torch.zeros(3, device=dev)
        m2d = getattr(self, "_static_m2d", None)
        if self.fused:
            if self.step_begin is None:
                self.step_begin, self.blend_begin = torch.cuda.Event(), torch.cuda.Event()
                self.blend_begin.record()  # creates the underlying hipEvent; the library re-records it in the forward
            self.step_begin.record()
        lights = []
        hook = after_blend((lambda: lights.extend(self._fused_begin())) if self.fused else None)
        # eager rasterizer without the host read-back: asynchronous binning into a fixed-capacity buffer (sized from one
        # synchronous probe), the overflow flag checked after the backward has been queued (GIGS_RASTER_ASYNC=0: read back)
        abin = None
        if self.fused and self.graphs and os.environ.get("GIGS_RASTER_ASYNC", "1") == "1" and not getattr(self, "_dense", False):
            abin = self._eager_async(cam, g, background)
        pre_grads = _snapshot_grads(self._leaves(g)) if abin is not None else None
        # the blend-begin event belongs to THIS step's forward: it is part of the library context the forward runs with
        # (gigs_ctx_set_blend_begin_event), not of the process
        with hook, (abin if abin is not None else _NULLCTX):
            ev_ctx = gigs_lib.use(gigs_lib.current().derive(blend_event=self.blend_begin)) if self.fused else _NULLCTX
            with ev_ctx:
                if self.pool is not None:
                    with self.pool:
                        out = rasterize(cam, g, self.sh_degree, background, self.gi, means2D=m2d)
                else:
                    out = rasterize(cam, g, self.sh_degree, background, self.gi, means2D=m2d)
        if abin is not None:
            abin.snapshot()
        ((_, radii, _, _, normal_map_from_depth, normal_map, occlusion_map, albedo_map, roughness_map, metallic_map,
          out_normal_view, depth_pos), screenspace_points, st) = out
        H, W = cam["image_height"], cam["image_width"]
        gi = self.gi
        if self.fused:
            res = self._fused_step(cam, gt_image, view_dirs, st, radii, screenspace_points, normal_map, out_normal_view,
                                   albedo_map, roughness_map, metallic_map, occlusion_map, depth_pos, lights,
                                   extra_loss)
            if abin is not None:
                try:
                    res["num_rendered"] = abin.check()
                except BinningOverflow as ex:
                    self._abin = AsyncBinning(-(-int(1.5 * ex.needed) // 65536) * 65536, dev)
                    _restore_grads(pre_grads)  # the overflowed step's contribution is dropped, earlier ones are kept
                    return self._step(cam, regen() if regen is not None else g, gt_image, view_dirs, extra_loss, regen)
            return res
        front_args = (normal_map_from_depth, normal_map, out_normal_view, albedo_map, roughness_map, metallic_map,
                      occlusion_map.detach(), st.viewmatrix, view_dirs)
        if self.graphs and not self._captured:
            # capture needs representative loss inputs: run the front eagerly once
            with torch.no_grad():
                o = self.front(*front_args)
            la = (o[0].clone().requires_grad_(True), torch.zeros_like(o[0]).requires_grad_(True), gt_image,
                  o[6], o[1].clone().requires_grad_(True), o[2].clone().requires_grad_(True))
            self._capture(front_args, la)
        (render_direct, roughness_f, metallic_f, F0, linear_rgb, onv, normal_mask_f) = self.front(*front_args)
        ssr = Gaussian_SSR(cam["tanfovx"], cam["tanfovy"], W, H, gi["radius"], gi["bias"], gi["thick"], gi["delta"],
                           gi["step"], gi["start"])
        (IRR, _) = ssr(onv, depth_pos.detach(), linear_rgb, albedo_map, roughness_f, metallic_f, F0)
        loss, render_rgb = self.loss_fn(render_direct, IRR, gt_image, normal_mask_f, roughness_f, metallic_f)
        if extra_loss is not None:
            loss = loss + extra_loss(normal_map, albedo_map, roughness_map, metallic_map)
        if self.regularizer is not None:
            loss = loss + self.regularizer(dict(normal_map=normal_map, albedo_map=albedo_map, roughness_map=roughness_map,
                                                metallic_map=metallic_map, gt_image=gt_image))
        loss.backward()
        return dict(loss=loss.detach(), render_rgb=render_rgb, render_direct=render_direct.detach(),
                    IRR=IRR.detach(), viewspace_points=screenspace_points, radii=radii)


def _grad_leaves(t: torch.Tensor):
    """The leaf tensors whose .grad a backward through `t` accumulates into (t itself if it is a leaf)."""
    if t.grad_fn is None:
        return [t] if t.requires_grad else []
    out, seen, todo = [], set(), [t.grad_fn]
    while todo:
        fn = todo.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        if hasattr(fn, "variable"):
            out.append(fn.variable)
        todo.extend(nf for nf, _ in fn.next_functions)
    return out


def _snapshot_grads(leaves):
    """[(leaf, None | copy of its .grad)]: what a step that may have to be repeated (binning overflow) must restore.
    Costs nothing in the usual case (gradients cleared before the step); one clone per tensor when accumulating."""
    return [(p, None if p.grad is None else p.grad.clone()) for p in leaves]


def _restore_grads(snapshot):
    for p, gr in snapshot or ():
        p.grad = gr


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULLCTX = _NullCtx()


def _eager_async(self, cam, g, background):
    if getattr(self, "_abin", None) is None:
        probe = GraphedRaster(cam, g, self.gi, self.sh_degree)._probe(cam, g, background)
        tiles = ((int(cam["image_height"]) + 15) // 16) * ((int(cam["image_width"]) + 15) // 16)
        if _declined_as_dense(probe, tiles):
            self._dense = True
            return None
        self._abin = AsyncBinning(max(65536, -(-2 * probe // 65536) * 65536), g["means3D"].device)
    return self._abin


def _graphed_step(self, cam, g, gt_image, view_dirs, extra_loss=None):
    """fused + graphs: the WHOLE step replays from hipGraphs -- the rasterizer with its in-op filters and SSAO too
    (GraphedRaster: asynchronous binning, no host read-back), the light filter on the side stream from the start of the
    step, the fused stage-2 node -- six graph launches per iteration.  The binning overflow flag of this step is
    checked after loss.backward() has been queued (the host then waits for the forward only); on overflow the
    capacity grows, the graphs are re-captured and the step is repeated on cleared gradients."""
    from types import SimpleNamespace
    dev = g["means3D"].device
    if self.step_begin is None:
        self.step_begin = torch.cuda.Event()
        self._bg = torch.zeros(3, device=dev)  # train.py:263-264: black background for PBR
    pre_grads = _snapshot_grads(self._leaves(g))
    for attempt in range(4):
        self.step_begin.record()
        lights = list(self._fused_begin(self.step_begin))
        if self.graster is None:
            self.graster = GraphedRaster(cam, g, self.gi, self.sh_degree)
        if getattr(self, "_m2d", None) is None or self._m2d.shape != g["means3D"].shape:
            self._m2d = torch.zeros_like(g["means3D"], requires_grad=True)  # persistent: the graph's static input
        screenspace_points = self._m2d
        screenspace_points.grad = None
        out = self.graster(cam, g, screenspace_points, self._bg)
        self._static_storages = set(t.untyped_storage().data_ptr() for t in out)
        (_, radii, _, _, _, normal_map, occlusion_map, albedo_map, roughness_map, metallic_map, out_normal_view, depth_pos) = out
        res = self._fused_step(cam, gt_image, view_dirs, SimpleNamespace(viewmatrix=cam["viewmatrix"]), radii,
                               screenspace_points, normal_map, out_normal_view, albedo_map, roughness_map, metallic_map,
                               occlusion_map, depth_pos, lights, extra_loss)
        if os.environ.get("GIGS_ASYNC_CHECK", "step") == "lazy":
            # diagnostic: do not wait for this step's forward here; an overflow of step k is then only noticed at step k + 1
            res["num_rendered"] = -1
            return res
        try:
            res["num_rendered"] = self.graster.check()
            return res
        except BinningOverflow:
            _restore_grads(pre_grads)
    raise RuntimeError("Stage2Step: the binning capacity kept overflowing")


def _fused_begin(self, start_event=None):
    """Starts light.build_mips() on a side stream, called from inside the rasterizer's forward as soon as its
    kernels up to the blend are queued (diff_gaussian_rasterization.after_blend; the host call returns after
    the binning read-back, while the GPU is still sorting): the GGX pre-filter is independent of the G-buffer
    and overlaps the latency-bound sort / blend kernels.  autograd runs a node's backward on the stream of its forward, so the light's
    backward likewise overlaps the rasterizer's backward."""
    from stage2_fused import LightMips
    main = torch.cuda.current_stream()
    if self.mips is None:
        # light.base's AccumulateGrad lives on the stream the parameter was created on, its gradient is produced on
        # the side stream: intended here (autograd inserts the event wait), so the advisory warning is switched off
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        self.side = torch.cuda.Stream()
        self.mips = LightMips(self.light)
        self.dummy = torch.zeros(1, device=self.light.base.device)
        if self.graphs:
            with torch.no_grad():
                self.mips(self.dummy)  # builds the cached filter tables outside the capture
            self.mips = graphed(self.mips, (self.dummy,))
    # Start when the blend kernel starts (an event the library records right before launching it): that kernel is
    # a few long serial walks with most CUs idle.  Measured alternatives: starting at once (next to the sort passes,
    # which then take 0.3-0.5 instead of 0.22 ms) and starting after the blend kernel (next to the VALU-bound SSAO
    # march only: 4 % slower overall).  GIGS_LIGHT_START=step selects the former.
    if start_event is not None:
        self.side.wait_event(start_event)
    elif os.environ.get("GIGS_LIGHT_START", "blend") == "step":
        self.side.wait_event(self.step_begin)
    else:
        self.side.wait_event(self.blend_begin)
    import pbr.light as pbr_light
    # this stepper schedules the light itself: the filters are built HERE, on its side stream (not on the light's own
    # stream, which serves the op-by-op caller: pbr/light.py), so that their backward runs there too
    with torch.cuda.stream(self.side), pbr_light.build_on_current_stream():
        out = self.mips(self.dummy)
        # what the shade waits for: an event of its own, so that work queued on this stream later (the regularisers,
        # _fused_step) is not waited for with it
        if getattr(self, "lights_ready", None) is None:
            self.lights_ready = torch.cuda.Event()
        self.lights_ready.record()
    return out


def _fused_step(self, cam, gt_image, view_dirs, st, radii, screenspace_points, normal_map, out_normal_view, albedo_map,
                roughness_map, metallic_map, occlusion_map, depth_pos, lights, extra_loss=None):
    """fused=True: everything after the rasterizer is stage2_fused._Stage2Fused (6 kernels instead of ~250)."""
    from stage2_fused import Stage2FusedBack
    H, W = cam["image_height"], cam["image_width"]
    main = torch.cuda.current_stream()
    # The regularisers (BRDF TV, envmap TV: ~15 small kernels forward, as many backward) read the rasterizer's planes and
    # the light only -- not the shade, the marches or the loss -- so they go to the light's stream, idle by now, beside the
    # shade and the SSR march; autograd runs their backward there too, beside the shade backward (GIGS_REG_SIDE=0: on the
    # caller's stream behind the loss, where they sit on the critical path on both ways: 0.2 ms per iteration at C2).
    reg_side = self.regularizer is not None and os.environ.get("GIGS_REG_SIDE", "1") == "1"
    if reg_side:
        if getattr(self, "raster_done", None) is None:
            self.raster_done = torch.cuda.Event()
        self.raster_done.record()  # the rasterizer's planes exist: all the regularisers wait for
    if getattr(self, "lights_ready", None) is not None:
        main.wait_event(self.lights_ready)
    else:
        main.wait_stream(self.side)
    for t in lights:
        t.record_stream(main)
    args = (normal_map.detach(), out_normal_view.detach(), albedo_map, roughness_map, metallic_map,
            occlusion_map.detach(), depth_pos.detach(), st.viewmatrix, view_dirs, gt_image, *lights)
    if self.back is None:
        # light_stream: the fused node's backward issues its light-texture gradient scatter there (stage2_fused.py) -- the
        # stream on which _fused_begin built the light's filters, hence on which autograd runs their backward, behind that
        # scatter.  Not under make_graphed_callables, whose per-callable capture cannot leave a forked stream unjoined
        cfg = dict(H=H, W=W, gi=self.gi, focal_x=W / (2.0 * cam["tanfovx"]), focal_y=H / (2.0 * cam["tanfovy"]),
                   light_stream=None if self.graphs else self.side, **self.flags)
        self.back = Stage2FusedBack(self.brdf_lut, cfg)
        if self.graphs:
            # pooled rasterizer planes (fixed addresses) become the graph's static inputs themselves: no staging copies
            pooled = set(t.untyped_storage().data_ptr() for t in self.pool.buffers.values()) if self.pool else set()
            pooled |= getattr(self, "_static_storages", set())  # outputs of the graphed rasterizer: fixed addresses too
            # ... and so are the filtered light levels (static outputs of the light's own graph): read in place, not staged
            pooled |= set(t.untyped_storage().data_ptr() for t in lights)
            sample = tuple(a.detach().requires_grad_(a.requires_grad) if a.untyped_storage().data_ptr() in pooled
                           else a.detach().clone().requires_grad_(a.requires_grad) for a in args)
            self.back = graphed(self.back, sample)
    loss, render_rgb, render_direct, IRR = self.back(*args)
    if extra_loss is not None:
        loss = loss + extra_loss(normal_map, albedo_map, roughness_map, metallic_map)
    if reg_side:
        # queued (host order) behind the fused node so that autograd, which walks later nodes first, starts their backward
        # before the shade backward; the stream waits only for the rasterizer's planes
        maps = dict(normal_map=normal_map, albedo_map=albedo_map, roughness_map=roughness_map, metallic_map=metallic_map,
                    gt_image=gt_image)
        self.side.wait_event(self.raster_done)
        with torch.cuda.stream(self.side):
            reg = self.regularizer(maps)
        for t in maps.values():
            t.record_stream(self.side)
        main.wait_stream(self.side)
        reg.record_stream(main)
        loss = loss + reg
    elif self.regularizer is not None:
        loss = loss + self.regularizer(dict(normal_map=normal_map, albedo_map=albedo_map, roughness_map=roughness_map,
                                            metallic_map=metallic_map, gt_image=gt_image))
    res = dict(loss=loss.detach(), render_rgb=render_rgb, render_direct=render_direct, IRR=IRR,
               viewspace_points=screenspace_points, radii=radii)
    if self._defer_backward:
        res["_loss"] = loss
        return res
    loss.backward()
    return res



def _close_stepper(self):
    """Deterministic teardown of everything that owns hipGraphs (WholeStepGraphs, graphed callables, the graphed
    rasterizer): synchronise, release, synchronise.  The stepper captures again if it is called afterwards.  Steppers and
    trainers are context managers (`with Stage2Trainer(...) as tr:`) that close on exit."""
    # the WholeStepGraph objects stay (they remember the binning capacity and count their captures); their graphs go
    for w in list(getattr(self, "_wholes", {}).values()):
        w.close()
    if getattr(self, "whole", None) is not None:
        self.whole.close()
    if getattr(self, "geom_cache", None) is not None:
        self.geom_cache.invalidate()  # the per-view entries (tile lists, occlusion planes, hit lists): device memory
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if getattr(self, "_captured", False):  # piecewise path: the two graphed callables
        self.front = Stage2Front(self.light, self.brdf_lut, **self.flags)
        self.loss_fn = stage2_loss
        self._captured = False
    if getattr(self, "graster", None) is not None:
        self.graster.close()
    for name in ("graster", "back", "mips"):
        if getattr(self, name, None) is not None:
            setattr(self, name, None)
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def _enter(self):
    return self


def _exit(self, *exc):
    self.close()
    return False

def _cached_geometry_step(self, cfg, cam, raw, gt_image, view_dirs):
    """One step through the frozen-geometry variants of the whole-step graphs: replayed from the view's entry when there
    is one, recorded otherwise; a replay that turns out stale (the previous update moved geometry) is repeated as a
    recording.  Returns None when this camera model has no slot left (more than four models): the caller's other paths."""
    if self.geom_cache is None:
        self.geom_cache = GeometryCache(next(iter(raw.values())).device)
    cache = self.geom_cache
    vkey = cache.view_key(cam)
    mode = cache.mode(vkey, raw)
    for _ in range(3):
        key = (cfg, mode)
        whole = self._wholes.get(key)
        if whole is None:
            if len(self._wholes) >= 8:
                return None
            whole = self._wholes[key] = WholeStepGraph(self, cam, raw, cache=cache, mode=mode)
        self.whole = whole
        out = whole(cam, raw, gt_image, view_dirs, vkey=vkey)
        if out is not None:
            return out
        mode = "record"
    raise RuntimeError("Stage2Step: the frozen-geometry step kept being invalidated")


Stage2Step._cached_geometry_step = _cached_geometry_step


def _make_inner(self):
    return Stage2Step(self.light, self.brdf_lut, self.gi, self.sh_degree, graphs=False, fused=True, regularizer=self.regularizer,
                      **self.flags)


Stage2Step._make_inner = _make_inner
Stage2Step.close = _close_stepper
Stage2Step.__enter__ = _enter
Stage2Step.__exit__ = _exit


def _leaves(self, g):
    out, seen = [], set()
    for t in list(g.values()) + list(self.light.parameters()):
        for leaf in _grad_leaves(t):
            if id(leaf) not in seen:
                seen.add(id(leaf))
                out.append(leaf)
    return out


Stage2Step._leaves = _leaves
Stage2Step._eager_async = _eager_async
Stage2Step._graphed_step = _graphed_step
Stage2Step._fused_begin = _fused_begin
Stage2Step._fused_step = _fused_step


class _Stage1Inner:
    """One stage-1 iteration of train.py (:266-331) up to the loss: render -> fused G-buffer post-processing ->
    0.8 L1 + 0.2 D-SSIM + masked normal L1 + normal TV (losses.stage1_loss: one autograd node)."""

    def __init__(self, gi: Dict, sh_degree: int, lambda_dssim: float, normal_loss_weight: float, normal_tv_weight: float):
        self.gi, self.sh_degree = gi, sh_degree
        self.w = (float(lambda_dssim), float(normal_loss_weight), float(normal_tv_weight))
        self._defer_backward = False
        self._static_bg = self._static_m2d = None

    def __call__(self, cam, g, gt_image, view_dirs=None):
        import losses
        dev = g["means3D"].device
        bg = self._static_bg if self._static_bg is not None else torch.zeros(3, device=dev)
        out, screenspace_points, st = rasterize(cam, g, self.sh_degree, bg, self.gi, means2D=self._static_m2d)
        (image, radii, _, _, normal_map_from_depth, normal_map, _, _, _, _, out_normal_view, _) = out
        nfd, nfd_mask, normals_view, _, _ = gbuffer_post_fused(normal_map_from_depth, normal_map, out_normal_view, st.viewmatrix)
        loss, ll1, normal_loss = losses.stage1_loss(image, gt_image, normals_view, nfd, nfd_mask, *self.w)
        res = dict(loss=loss.detach(), Ll1=ll1.detach(), normal_loss=normal_loss.detach(), render=image.detach(),
                   viewspace_points=screenspace_points, radii=radii)
        if self._defer_backward:
            res["_loss"] = loss
            return res
        loss.backward()
        return res


class Stage1Step:
    """One stage-1 iteration (`iteration <= pbr_iteration`, train.py:266-331, 517-520) with Stage2Step's interface and
    formulations: eager, or -- graphs=True -- the whole iteration from hipGraphs (WholeStepGraph: forward, backward and,
    with `optimizers`, the update), the rasterizer under asynchronous binning.  No light, no shade: the loss reaches
    colour and normals, i.e. every Gaussian group."""

    light = None
    regularizer = None
    post_update = None

    def __init__(self, gi: Dict, sh_degree: int, lambda_dssim: float = 0.2, normal_loss_weight: float = 1.0,
                 normal_tv_weight: float = 1.0, graphs: bool = False, prepare=None, optimizers=None, before_update=None):
        self.gi, self.sh_degree, self.graphs = gi, sh_degree, graphs
        self.weights = (lambda_dssim, normal_loss_weight, normal_tv_weight)
        self.prepare, self.optimizers, self.before_update = prepare, list(optimizers or []), before_update
        self.whole, self._wholes = None, {}
        self._eager = self._make_inner()

    def _make_inner(self):
        return _Stage1Inner(self.gi, self.sh_degree, *self.weights)

    def __call__(self, cam: Dict, g: Dict[str, torch.Tensor], gt_image: torch.Tensor, view_dirs=None):
        raw = g
        if self.graphs and os.environ.get("GIGS_STEP_GRAPH", "1") == "1" and not getattr(self, "_dense", False):
            try:
                cfg = (int(cam["image_height"]), int(cam["image_width"]), float(cam["tanfovx"]), float(cam["tanfovy"]))
                if self.whole is None or self.whole.cfg != cfg:
                    self.whole = self._wholes.get(cfg)
                    if self.whole is None and len(self._wholes) < 4:
                        self.whole = self._wholes[cfg] = WholeStepGraph(self, cam, g)
                if self.whole is not None:
                    if getattr(self, "_no_vd", None) is None:
                        self._no_vd = torch.zeros(1, device=gt_image.device)  # the graph's (unused) view_dirs input
                    return self.whole(cam, raw, gt_image, view_dirs if view_dirs is not None else self._no_vd)
            except DenseScene:
                self._dense = True
                for w in self._wholes.values():
                    w.close()
                self.whole = None
                self._wholes.clear()
        if self.prepare is not None:
            g = self.prepare(raw)
        res = self._eager(cam, g, gt_image)
        if self.optimizers:
            if self.before_update is not None:
                self.before_update()
            for o in self.optimizers:
                o.step()
            for leaf in raw.values():
                leaf.grad = None
        return res


Stage1Step.close = _close_stepper
Stage1Step.__enter__ = _enter
Stage1Step.__exit__ = _exit


def stage2_step(cam: Dict, g: Dict[str, torch.Tensor], sh_degree: int, gi: Dict, light, brdf_lut: torch.Tensor,
                gt_image: torch.Tensor, rays: torch.Tensor, metallic: bool = True, indirect: bool = True,
                gamma: bool = False, tone: bool = False, view_dirs: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Eager convenience wrapper around Stage2Step (one-off use; bench.py keeps a Stage2Step)."""
    if view_dirs is None:  # a per-camera constant (train.py:299-308); callers may cache it with the camera
        view_dirs = view_dirs_for(cam, rays, g["means3D"].device)
    step = Stage2Step(light, brdf_lut, gi, sh_degree, metallic=metallic, indirect=indirect, gamma=gamma, tone=tone)
    return step(cam, g, gt_image, view_dirs)
