"""Stage-2 iteration between the rasterizer and loss.backward() as ONE autograd node.

The reference expresses this stretch of train.py (:293-402) as ~120 torch ops forward and as many
backward; `pipeline.Stage2Front` / `pipeline.stage2_loss` restate it op by op.  Here the same
arithmetic runs as five kernels forward (gigs_gbuffer_post, gigs_shade_fwd_ex, gigs_ssr,
gigs_stage2_loss_fwd_grad + its 1-thread finish) and ONE backward (gigs_shade_bwd_ex: the loss's
gradient planes are written by the forward pass for a unit upstream gradient and scaled there),
all reading and writing the rasterizer's [C,H,W] planes directly.

Gradients leave through exactly the tensors train.py differentiates: albedo_map, roughness_map,
metallic_map (-> the rasterizer's backward) and light.diffuse / light.specular (-> build_mips'
backward -> light.base).  normal maps, occlusion, depth_pos and linear_rgb are detached there as here.
tests/test_gpu_pbr.py::test_stage2_fused_matches_unfused compares the two formulations.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict

import torch

import gigs_lib
from diff_gaussian_rasterization import _C as _ops
from pbr.shade import _ptr_array

_lib = gigs_lib.lib()


def _p(t):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _Stage2Fused(torch.autograd.Function):
    """(cfg, normal_map, out_normal_view, albedo_map, roughness_map, metallic_map, occlusion_map, depth_pos,
    viewmatrix, view_dirs, gt_image, lut, diffuse, *specular) -> (loss, render_rgb, render_direct, IRR)."""

    @staticmethod
    def forward(ctx, cfg, normal_map, out_normal_view, albedo_map, roughness_map, metallic_map, occlusion_map,
                depth_pos, viewmatrix, view_dirs, gt_image, lut, diffuse, *specular):
        dev = albedo_map.device
        if not albedo_map.is_cuda:
            raise RuntimeError("stage2_fused needs CUDA/HIP tensors: gigs-hip has no CPU path")
        ctx.set_materialize_grads(False)  # the three image outputs carry no gradient: no zero tensors for them (3 fills / step)
        H, W = int(cfg["H"]), int(cfg["W"])
        f = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        normal_map, out_normal_view, albedo_map = f(normal_map), f(out_normal_view), f(albedo_map)
        roughness_map, metallic_map, depth_pos = f(roughness_map), f(metallic_map), f(depth_pos)
        occlusion = f(occlusion_map) if cfg["indirect"] else None
        viewmatrix, view_dirs, gt_image, lut, diffuse = f(viewmatrix), f(view_dirs), f(gt_image), f(lut), f(diffuse)
        specular = [f(s) for s in specular]
        use_metallic = bool(cfg["metallic"])

        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        normals_view, onv = new(3, H, W), new(3, H, W)
        mask_u8 = torch.empty((H, W), dtype=torch.uint8, device=dev)
        mask_f = new(1, H, W)
        render_direct, F0, linear_rgb, rough_f = new(3, H, W), new(3, H, W), new(3, H, W), new(1, H, W)
        render_rgb, acc4, loss = new(3, H, W), new(4 + 4 * 256), new(1)  # GIGS_STAGE2_ACC_FLOATS
        spec_ptr = _ptr_array(specular)
        spec_res = (C.c_int * len(specular))(*[int(s.shape[1]) for s in specular])
        ext = gigs_lib.ShadeExt(planar=1, rough_scale=1.0 - 0.04, rough_bias=0.04, out_F0=_p(F0), out_linear=_p(linear_rgb),
                                out_roughness=_p(rough_f))
        gi = cfg["gi"]
        with torch.cuda.device(dev):
            s = _stream()
            gigs_lib.check(_lib.gigs_gbuffer_post(H, W, _p(normal_map), _p(out_normal_view), _p(viewmatrix),
                                                  _p(normals_view), _p(mask_u8), _p(mask_f), _p(onv), s), "gbuffer_post")
            gigs_lib.check(_lib.gigs_shade_fwd_ex(
                gigs_lib.ctx_ptr(), H, W, _p(normals_view), _p(view_dirs), _p(albedo_map), _p(roughness_map), _p(mask_u8), _p(occlusion),
                _p(metallic_map) if use_metallic else None, None, _p(diffuse), int(diffuse.shape[1]), len(specular),
                spec_ptr, spec_res, _p(lut), int(lut.shape[-2]), int(lut.shape[-3]), int(bool(cfg["tone"])),
                int(bool(cfg["gamma"])), _p(render_direct), None, None, None, C.addressof(ext), s), "shade_fwd_ex")
            metallic_f = metallic_map if use_metallic else torch.zeros_like(rough_f)
            # Gaussian_SSR (train.py:370-379); its backward is closed-form (grad_albedo = grad * abd)
            IRR, abd = _ops.SSR(W, H, cfg["focal_x"], cfg["focal_y"], gi["radius"], gi["bias"], gi["thick"], gi["delta"],
                                gi["step"], gi["start"], onv, depth_pos, linear_rgb, albedo_map, rough_f, metallic_f, F0)
            # the loss and, in the same pass over the image, its gradients w.r.t. render_direct / IRR for a unit upstream
            # gradient (the backward then has no loss kernel: gigs_shade_bwd_ex scales them and forms the lamb terms)
            d_direct_u, d_irr_u = new(3, H, W), new(3, H, W)
            gigs_lib.check(_lib.gigs_stage2_loss_fwd_grad(H, W, _p(render_direct), _p(IRR), _p(gt_image), _p(mask_f),
                                                          _p(rough_f), _p(metallic_f), _p(render_rgb), _p(acc4), _p(loss),
                                                          _p(d_direct_u), _p(d_irr_u), s), "stage2_loss_fwd_grad")
        ctx.save_for_backward(normals_view, view_dirs, albedo_map, roughness_map, mask_u8, mask_f, occlusion,
                              metallic_map if use_metallic else None, lut, diffuse, d_direct_u, d_irr_u, abd,
                              acc4, *specular)
        ctx.cfg = cfg
        ctx.lib_ctx = gigs_lib.current()
        ctx.need_light = (ctx.needs_input_grad[12], [ctx.needs_input_grad[13 + i] for i in range(len(specular))])
        loss = loss.reshape(())
        ctx.mark_non_differentiable(render_rgb, render_direct, IRR)
        return loss, render_rgb, render_direct, IRR

    @staticmethod
    @gigs_lib.with_forward_context
    def backward(ctx, g_loss, *_unused):
        (normals_view, view_dirs, albedo_map, roughness_map, mask_u8, mask_f, occlusion, metallic_map, lut, diffuse,
         d_direct_u, d_irr_u, abd, acc4, *specular) = ctx.saved_tensors
        cfg = ctx.cfg
        dev = albedo_map.device
        H, W = int(cfg["H"]), int(cfg["W"])
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        if g_loss is None:
            g_loss = torch.zeros((), dtype=torch.float32, device=dev)
        g_loss = g_loss.contiguous().float()
        d_albedo, d_rough = new(3, H, W), new(1, H, W)
        d_metal = new(1, H, W) if metallic_map is not None else None
        need_d, need_s = ctx.need_light
        # one zero-filled slab for every light-gradient texture (they are accumulated with atomics)
        sizes = [diffuse.numel() if need_d else 0] + [s.numel() if n else 0 for s, n in zip(specular, need_s)]
        slab = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        views, off = [], 0
        for n, ref in zip(sizes, [diffuse] + list(specular)):
            views.append(slab[off:off + n].view(ref.shape) if n else None)
            off += n
        d_diffuse, d_spec = views[0], views[1:]
        spec_ptr, dspec_ptr = _ptr_array(specular), _ptr_array(d_spec)
        spec_res = (C.c_int * len(specular))(*[int(s.shape[1]) for s in specular])
        # unit-gradient planes from the forward, scaled by g_loss inside the kernel; lamb terms formed there too
        def launch(part, stream):
            ext = gigs_lib.ShadeExt(planar=1, rough_scale=1.0 - 0.04, rough_bias=0.04, g_albedo_mul_a=_p(d_irr_u),
                                    g_albedo_mul_b=_p(abd), g_scale=_p(g_loss), lamb_mask=_p(mask_f), lamb_acc4=_p(acc4), part=part)
            gigs_lib.check(_lib.gigs_shade_bwd_ex(
                gigs_lib.ctx_ptr(), H, W, _p(normals_view), _p(view_dirs), _p(albedo_map), _p(roughness_map), _p(mask_u8), _p(occlusion),
                _p(metallic_map), _p(diffuse), int(diffuse.shape[1]), len(specular), spec_ptr, spec_res, _p(lut),
                int(lut.shape[-2]), int(lut.shape[-3]), int(bool(cfg["tone"])), int(bool(cfg["gamma"])), _p(d_direct_u),
                None, None, None, _p(d_albedo), _p(d_rough), _p(d_metal), _p(d_diffuse), dspec_ptr, C.addressof(ext), stream),
                "shade_bwd_ex")

        light_stream = cfg.get("light_stream")
        any_light = need_d or any(need_s)
        with torch.cuda.device(dev):
            if light_stream is None or not any_light or os.environ.get("GIGS_SHADE_BWD_SPLIT", "0") != "1":
                launch(0, _stream())
            else:
                # Two launches of the same kernel.  The material gradients (to the rasterizer's backward) are one cheap pass;
                # the light-texture gradients are the float-atomic scatter that dominates this node (0.14 of 0.21 ms at C2) and
                # feed the light filters' backward, which runs on the light's stream anyway: issued there, they leave the
                # step's critical path (shade backward -> blend backward -> preprocess backward) and run beside it.
                main = torch.cuda.current_stream()
                light_stream.wait_stream(main)  # the inputs, the zero-filled gradient slab
                launch(1, main.cuda_stream)
                with torch.cuda.stream(light_stream):
                    head = int(1e3 * float(os.environ.get("GIGS_SHADE_LIGHT_HEAD_START_US", "0")))
                    if head > 0:
                        gigs_lib.check(_lib.gigs_stream_delay(head, light_stream.cuda_stream), "stream_delay")
                    launch(2, light_stream.cuda_stream)
                slab.record_stream(light_stream)
                for t in (normals_view, view_dirs, albedo_map, roughness_map, mask_u8, occlusion, metallic_map, lut, diffuse,
                          d_direct_u, d_irr_u, abd, acc4, g_loss, mask_f, *specular):
                    if t is not None:
                        t.record_stream(light_stream)
                # no join here: the only consumer of the light gradients is the light filters' backward, which autograd
                # runs on the stream of its forward -- light_stream itself (pipeline._fused_begin) -- i.e. behind part 2 in
                # stream order; the engine joins that stream with the caller's when the backward ends
        return (None, None, None, d_albedo, d_rough, d_metal, None, None, None, None, None, None, d_diffuse, *d_spec)


class LightMips(torch.nn.Module):
    """light.build_mips() as a tensor function of light.base: (diffuse, *specular).  The dummy argument only
    gives make_graphed_callables a tensor input."""

    def __init__(self, light):
        super().__init__()
        self.light = light

    def forward(self, _dummy):
        self.light.build_mips()
        return (self.light.diffuse, *self.light.specular)


class Stage2FusedBack(torch.nn.Module):
    """The fused node as a static-shape tensor function of the G-buffer planes and the filtered light
    (hipGraph-capturable)."""

    def __init__(self, brdf_lut: torch.Tensor, cfg: Dict):
        super().__init__()
        self.register_buffer("brdf_lut", brdf_lut, persistent=False)
        self.cfg = cfg

    def forward(self, normal_map, out_normal_view, albedo_map, roughness_map, metallic_map, occlusion_map, depth_pos,
                viewmatrix, view_dirs, gt_image, diffuse, *specular):
        return _Stage2Fused.apply(self.cfg, normal_map, out_normal_view, albedo_map, roughness_map, metallic_map,
                                  occlusion_map, depth_pos, viewmatrix, view_dirs, gt_image, self.brdf_lut, diffuse,
                                  *specular)
