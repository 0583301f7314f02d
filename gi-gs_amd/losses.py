"""The training loop's image losses on libgigs_hip.

Same names, arguments and results as the reference's helpers, each one HIP pass forward and one backward
instead of a chain of torch ops:

    l1_loss(network_output, gt)                          utils/loss_utils.py:19-20
    ssim(img1, img2, window_size=11, size_average=True)  utils/loss_utils.py:55-98
    l1_ssim_loss(image, gt, lambda_dssim)                train.py:318-320  (1-l)*L1 + l*(1-ssim), one kernel
    get_tv_loss(gt_image, prediction, pad=1, step=1)     train.py:83-113
    get_masked_tv_loss(mask, gt_image, prediction)       train.py:116-142
    masked_l1_loss(a, b, mask)                           train.py:327  F.l1_loss(a[:, mask], b[:, mask])
    stage1_loss(...)                                     train.py:318-331, the whole stage-1 objective
    get_envmap_dirs(res), env_tv_loss(base, dirs)        train.py:145-157, :405-424

Images are [C,H,W] fp32 on the GPU.  There is no CPU path: without libgigs_hip.so importing this module fails,
and CPU tensors raise.
"""
from __future__ import annotations

import torch

import gigs_lib

_lib = gigs_lib.lib()


def _p(t):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chw(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: needs CUDA/HIP tensors, gigs-hip has no CPU path")
    if t.dim() != 3:
        raise ValueError(f"{what}: expected [C,H,W], got {tuple(t.shape)}")
    return t.contiguous().float()


def _scratch(C: int, H: int, W: int, dev) -> torch.Tensor:
    return torch.empty(int(_lib.gigs_loss_scratch_floats(C, H, W)), dtype=torch.float32, device=dev)


class _L1Ssim(torch.autograd.Function):
    """(image, gt, lambda) -> [loss, mean|image-gt|, mean ssim]; gradient to image through loss only."""

    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        image, gt = _chw(image, "l1_ssim"), _chw(gt, "l1_ssim")
        if image.shape != gt.shape:
            raise ValueError("l1_ssim: image and gt differ in shape")
        C, H, W = image.shape
        need = bool(ctx.needs_input_grad[0])
        d = torch.empty((3, C, H, W), dtype=torch.float32, device=image.device) if need else None
        out = torch.empty(3, dtype=torch.float32, device=image.device)
        with torch.cuda.device(image.device):
            gigs_lib.check(_lib.gigs_l1_ssim_fwd(C, H, W, _p(image), _p(gt), float(lambda_dssim),
                                                 _p(d[0]) if need else None, _p(d[1]) if need else None,
                                                 _p(d[2]) if need else None, _p(_scratch(C, H, W, image.device)),
                                                 _p(out), _stream()), "l1_ssim_fwd")
        ctx.lam = float(lambda_dssim)
        ctx.save_for_backward(image, gt, d if need else torch.empty(0, device=image.device))
        return out

    @staticmethod
    def backward(ctx, g_out):
        image, gt, d = ctx.saved_tensors
        C, H, W = image.shape
        g_image = torch.empty_like(image)
        # only out[0] (the loss) carries gradient here; out[1:] are reporting values
        g = g_out[0:1].contiguous().float()
        with torch.cuda.device(image.device):
            gigs_lib.check(_lib.gigs_l1_ssim_bwd(C, H, W, _p(image), _p(gt), ctx.lam, _p(d[0]), _p(d[1]), _p(d[2]),
                                                 _p(g), _p(g_image), _stream()), "l1_ssim_bwd")
        return g_image, None, None


def l1_ssim_loss(image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = 0.2) -> torch.Tensor:
    """train.py:320: (1 - lambda) * l1 + lambda * (1 - ssim(image, gt)), one fused pass."""
    return _L1Ssim.apply(image, gt, lambda_dssim)[0]


def l1_loss(network_output: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """utils/loss_utils.py:19-20."""
    return _L1Ssim.apply(network_output, gt, 0.0)[0]


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, size_average: bool = True) -> torch.Tensor:
    """utils/loss_utils.py:55-98 for one [C,H,W] image, window 11 and the scalar mean (what train.py uses)."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("ssim: only window_size=11, size_average=True (the reference's only use)")
    # loss = 1 - ssim at lambda = 1
    return 1.0 - _L1Ssim.apply(img1, img2, 1.0)[0]


class _Tv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gt, prediction, mask_f, step):
        prediction = _chw(prediction, "tv_loss")
        gt = None if gt is None else _chw(gt, "tv_loss")
        C, H, W = prediction.shape
        if gt is not None and gt.shape != (3, H, W):
            raise ValueError("tv_loss: gt_image must be [3,H,W] matching prediction")
        if mask_f is not None:
            mask_f = mask_f.reshape(H, W).contiguous().float()
        dev = prediction.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            gigs_lib.check(_lib.gigs_tv_loss_fwd(C, H, W, int(step), _p(gt), _p(prediction), _p(mask_f),
                                                 _p(_scratch(C, H, W, dev)), _p(loss), _stream()), "tv_loss_fwd")
        ctx.step = int(step)
        ctx.has_mask, ctx.has_gt = mask_f is not None, gt is not None
        none = torch.empty(0, device=dev)
        ctx.save_for_backward(gt if gt is not None else none, prediction, mask_f if mask_f is not None else none)
        return loss[0]

    @staticmethod
    def backward(ctx, g_loss):
        gt, prediction, mask_f = ctx.saved_tensors
        C, H, W = prediction.shape
        g_pred = torch.empty_like(prediction)
        g = g_loss.reshape(1).contiguous().float()
        with torch.cuda.device(prediction.device):
            gigs_lib.check(_lib.gigs_tv_loss_bwd(C, H, W, ctx.step, _p(gt) if ctx.has_gt else None, _p(prediction),
                                                 _p(mask_f) if ctx.has_mask else None, _p(g), _p(g_pred), _stream()),
                           "tv_loss_bwd")
        return None, g_pred, None, None


def get_tv_loss(gt_image: torch.Tensor, prediction: torch.Tensor, pad: int = 1, step: int = 1) -> torch.Tensor:
    """train.py:83-113 (pad > 1 average-pools both inputs first, as there)."""
    if pad > 1:
        gt_image = torch.nn.functional.avg_pool2d(gt_image, pad, pad)
        prediction = torch.nn.functional.avg_pool2d(prediction, pad, pad)
    return _Tv.apply(gt_image, prediction, None, step)


def nonzero_mask(planes: torch.Tensor) -> torch.Tensor:
    """(planes != 0).all(0, keepdim=True).float() of a [3,H,W] tensor in one launch (gaussian_renderer/__init__.py:158)."""
    x = planes.detach()
    if not x.is_cuda or x.dim() != 3 or x.shape[0] != 3:
        return (x != 0).all(0, keepdim=True).float()
    x = x.contiguous().float()
    out = torch.empty((1,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        gigs_lib.check(_lib.gigs_nonzero_mask(int(x.shape[1]), int(x.shape[2]), _p(x), _p(out), _stream()), "nonzero_mask")
    return out


def get_masked_tv_loss(mask: torch.Tensor, gt_image: torch.Tensor, prediction: torch.Tensor,
                       erosion: bool = False) -> torch.Tensor:
    """train.py:116-142; `erosion` (a 7x7 kornia morphology pass the reference never enables) is not provided."""
    if erosion:
        raise NotImplementedError("get_masked_tv_loss: erosion=True is unused by the reference and not implemented")
    return _Tv.apply(gt_image, prediction, mask.float(), 1)


class _MaskedL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mask):
        a, b = _chw(a, "masked_l1"), _chw(b, "masked_l1")
        C, H, W = a.shape
        if b.shape != a.shape or mask.numel() != H * W:
            raise ValueError("masked_l1: shapes differ")
        mask_u8 = mask.reshape(H, W).contiguous().to(torch.uint8)
        out = torch.empty(2, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            gigs_lib.check(_lib.gigs_masked_l1_fwd(C, H, W, _p(a), _p(b), _p(mask_u8), _p(_scratch(C, H, W, a.device)),
                                                   _p(out), _stream()), "masked_l1_fwd")
        ctx.save_for_backward(a, b, mask_u8, out)
        return out[0]

    @staticmethod
    def backward(ctx, g_loss):
        a, b, mask_u8, out = ctx.saved_tensors
        C, H, W = a.shape
        g_a = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        g_b = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if g_a is None and g_b is None:
            return None, None, None
        g = g_loss.reshape(1).contiguous().float()
        with torch.cuda.device(a.device):
            gigs_lib.check(_lib.gigs_masked_l1_bwd(C, H, W, _p(a), _p(b), _p(mask_u8), _p(out), _p(g), _p(g_a), _p(g_b),
                                                   _stream()), "masked_l1_bwd")
        return g_a, g_b, None


def masked_l1_loss(a: torch.Tensor, b: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """train.py:327: F.l1_loss(a[:, mask], b[:, mask]) with a boolean [H,W] mask (NaN for an empty mask, as torch)."""
    return _MaskedL1.apply(a, b, mask)


class _Stage1Loss(torch.autograd.Function):
    """The three stage-1 terms as ONE autograd node: six kernels forward (three passes + their finish kernels), three
    backward, instead of three nodes plus the scalar arithmetic between them (the Python / autograd hops cost more than
    the kernels at 800x800).  Gradients: image <- colour term, normal_map <- masked L1 + TV."""

    @staticmethod
    def forward(ctx, image, gt, normal_map, nfd, mask, lam, w_normal, w_tv):
        image, gt = _chw(image, "stage1_loss"), _chw(gt, "stage1_loss")
        nm, nfd = _chw(normal_map, "stage1_loss"), _chw(nfd, "stage1_loss")
        C, H, W = image.shape
        if gt.shape != image.shape or nm.shape != (3, H, W) or nfd.shape != (3, H, W) or mask.numel() != H * W:
            raise ValueError("stage1_loss: shapes differ")
        dev = image.device
        mask_u8 = mask.reshape(H, W).contiguous().to(torch.uint8)
        need_img, need_nm = bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[2])
        d = torch.empty((3, C, H, W), dtype=torch.float32, device=dev) if need_img else None
        out = torch.empty(6, dtype=torch.float32, device=dev)  # {colour loss, l1, ssim, normal l1, count, tv}
        with torch.cuda.device(dev):
            s = _stream()
            gigs_lib.check(_lib.gigs_l1_ssim_fwd(C, H, W, _p(image), _p(gt), float(lam), _p(d[0]) if need_img else None,
                                                 _p(d[1]) if need_img else None, _p(d[2]) if need_img else None,
                                                 _p(_scratch(C, H, W, dev)), _p(out), s), "l1_ssim_fwd")
            gigs_lib.check(_lib.gigs_masked_l1_fwd(3, H, W, _p(nm), _p(nfd), _p(mask_u8), _p(_scratch(3, H, W, dev)),
                                                   _p(out[3:5]), s), "masked_l1_fwd")
            gigs_lib.check(_lib.gigs_tv_loss_fwd(3, H, W, 1, _p(gt), _p(nm), None, _p(_scratch(3, H, W, dev)), _p(out[5:6]),
                                                 s), "tv_loss_fwd")
        loss = out[0] + float(w_normal) * out[3] + float(w_tv) * out[5]
        ctx.lam, ctx.w_normal, ctx.w_tv = float(lam), float(w_normal), float(w_tv)
        ctx.need_img, ctx.need_nm = need_img, need_nm
        ctx.save_for_backward(image, gt, nm, nfd, mask_u8, out, d if need_img else torch.empty(0, device=dev))
        ll1, nl = out[1].clone(), out[3].clone()
        ctx.mark_non_differentiable(ll1, nl)
        return loss, ll1, nl

    @staticmethod
    def backward(ctx, g_loss, _g_l1, _g_nl):
        image, gt, nm, nfd, mask_u8, out, d = ctx.saved_tensors
        C, H, W = image.shape
        g = g_loss.reshape(1).contiguous().float()
        g_image = g_nm = None
        with torch.cuda.device(image.device):
            s = _stream()
            if ctx.need_img:
                g_image = torch.empty_like(image)
                gigs_lib.check(_lib.gigs_l1_ssim_bwd(C, H, W, _p(image), _p(gt), ctx.lam, _p(d[0]), _p(d[1]), _p(d[2]), _p(g),
                                                     _p(g_image), s), "l1_ssim_bwd")
            if ctx.need_nm:
                g_nm, g_tv = torch.empty_like(nm), torch.empty_like(nm)
                gn, gt_ = g * ctx.w_normal, g * ctx.w_tv
                gigs_lib.check(_lib.gigs_masked_l1_bwd(3, H, W, _p(nm), _p(nfd), _p(mask_u8), _p(out[3:5]), _p(gn), _p(g_nm),
                                                       None, s), "masked_l1_bwd")
                gigs_lib.check(_lib.gigs_tv_loss_bwd(3, H, W, 1, _p(gt), _p(nm), None, _p(gt_), _p(g_tv), s), "tv_loss_bwd")
                g_nm += g_tv
        return g_image, None, g_nm, None, None, None, None, None


def stage1_loss(image, gt_image, normal_map, normal_map_from_depth, normal_from_depth_mask, lambda_dssim=0.2,
                normal_loss_weight=1.0, normal_tv_weight=1.0):
    """The stage-1 objective of train.py:318-331 -> (loss, Ll1, normal_loss), one autograd node."""
    return _Stage1Loss.apply(image, gt_image, normal_map, normal_map_from_depth, normal_from_depth_mask, lambda_dssim,
                             normal_loss_weight, normal_tv_weight)


def get_envmap_dirs(res=(512, 1024), device="cuda") -> torch.Tensor:
    """train.py:145-157: directions of an equirectangular [H,W] panorama."""
    import math
    gy, gx = torch.meshgrid(torch.linspace(0.0 + 1.0 / res[0], 1.0 - 1.0 / res[0], res[0], device=device),
                            torch.linspace(-1.0 + 1.0 / res[1], 1.0 - 1.0 / res[1], res[1], device=device),
                            indexing="ij")
    sintheta, costheta = torch.sin(gy * math.pi), torch.cos(gy * math.pi)
    sinphi, cosphi = torch.sin(gx * math.pi), torch.cos(gx * math.pi)
    return torch.stack((sintheta * sinphi, costheta, -sintheta * cosphi), dim=-1)  # [H, W, 3]


def env_tv_loss(cubemap_base: torch.Tensor, envmap_dirs: torch.Tensor) -> torch.Tensor:
    """train.py:405-424: TV smoothness of the panorama sampled from the light's base cubemap,
    mean((E[1:]-E[:-1])^2) + mean((E[:,1:]-E[:,:-1])^2); two kernels forward (lookup into planes, TV), two back."""
    from pbr.texture import cube_texture
    env = cube_texture(cubemap_base, envmap_dirs, planar=True)  # [3, H, W]
    return _Tv.apply(None, env, None, 1)
