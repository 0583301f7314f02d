"""CPU restatement of the training-loop glue.  TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product (gi-gs_amd/losses.py and optim.py call libgigs_hip and have no CPU path).

Pinning:
  * ssim_ref / l1_ref      <- tests/golden/ref_loss.npz, produced by the reference's own utils/loss_utils.py
                              (importable on CPU) on seeded images, values and autograd gradients
                              (tests/golden/make_loss_golden.py).
  * adam_ref               <- torch.optim.Adam itself, which is what the reference instantiates
                              (scene/gaussian_model.py:346); tests compare against it directly.
  * tv_ref / masked_tv_ref / masked_l1_ref  <- restated from train.py:83-142, :327.  train.py is not importable here
                              (kornia, nvdiffrast, cv2 are absent) and the reference holds no fixture for them:
                              PARITY UNPINNED beyond the definition; a hand-computed case is in tests/test_losses_cpu.py.
"""
import math

import torch
import torch.nn.functional as F


def gaussian_taps(size: int = 11, sigma: float = 1.5) -> torch.Tensor:
    """utils/loss_utils.py:42-46: per-tap exp in double, stored fp32, normalised by the fp32 sum."""
    half = size // 2
    taps = torch.tensor([math.exp(-((i - half) ** 2) / float(2 * sigma ** 2)) for i in range(size)], dtype=torch.float32)
    return taps / taps.sum()


def ssim_map_ref(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """utils/loss_utils.py:72-93 for [C,H,W] inputs: depthwise, zero-padded 11x11 window (outer product of the
    taps, :49-53), C1 = 0.01^2, C2 = 0.03^2."""
    C = x.shape[0]
    t = gaussian_taps().to(x.dtype)
    win = torch.outer(t, t)[None, None].expand(C, 1, 11, 11).contiguous()

    def blur(img):
        return F.conv2d(img[None], win, padding=5, groups=C)[0]

    mx, my = blur(x), blur(y)
    vx = blur(x * x) - mx * mx
    vy = blur(y * y) - my * my
    cxy = blur(x * y) - mx * my
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mx * my + c1) * (2 * cxy + c2)) / ((mx * mx + my * my + c1) * (vx + vy + c2))


def ssim_ref(x, y):
    return ssim_map_ref(x, y).mean()  # loss_utils.py:95-96


def l1_ref(x, y):
    return (x - y).abs().mean()  # loss_utils.py:19-20


def l1_ssim_ref(x, y, lam):
    return (1.0 - lam) * l1_ref(x, y) + lam * (1.0 - ssim_ref(x, y))  # train.py:320


def _edge_weights(gt, s):
    wh = torch.exp(-(gt[:, s:, :] - gt[:, :-s, :]).abs().mean(dim=0, keepdim=True))
    ww = torch.exp(-(gt[:, :, s:] - gt[:, :, :-s]).abs().mean(dim=0, keepdim=True))
    return wh, ww


def tv_ref(gt, pred, step: int = 1, mask=None):
    """train.py:83-113 (pad = 1) and, with `mask` [1,H,W], :116-142 (no erosion): for every offset s in 1..step,
    mean over [C,H-s,W] of (pred(y+s)-pred(y))^2 * exp(-mean_c|gt(y+s)-gt(y)|) [* mask(y+s)*mask(y)], plus the same
    along x."""
    total = pred.new_zeros(())
    for s in range(1, step + 1):
        wh, ww = _edge_weights(gt, s)
        if mask is not None:
            m = mask.to(pred.dtype)
            wh = wh * (m[:, s:, :] * m[:, :-s, :])
            ww = ww * (m[:, :, s:] * m[:, :, :-s])
        dh = (pred[:, s:, :] - pred[:, :-s, :]) ** 2
        dw = (pred[:, :, s:] - pred[:, :, :-s]) ** 2
        total = total + (dh * wh).mean() + (dw * ww).mean()
    return total


def masked_l1_ref(a, b, mask):
    """train.py:327."""
    return F.l1_loss(a[:, mask], b[:, mask])


def adam_ref(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-15):
    """One torch.optim.Adam update (torch/optim/adam.py, single-tensor path, no weight decay / amsgrad) written out;
    returns new (p, m, v).  `step` is the 1-based update count."""
    m = m + (g - m) * (1.0 - beta1)
    v = v * beta2 + (g * g) * (1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * (m / denom), m, v


def envmap_dirs_ref(res=(512, 1024)):
    """train.py:145-157."""
    gy, gx = torch.meshgrid(torch.linspace(0.0 + 1.0 / res[0], 1.0 - 1.0 / res[0], res[0]),
                            torch.linspace(-1.0 + 1.0 / res[1], 1.0 - 1.0 / res[1], res[1]), indexing="ij")
    st, ct = torch.sin(gy * math.pi), torch.cos(gy * math.pi)
    sp, cp = torch.sin(gx * math.pi), torch.cos(gx * math.pi)
    return torch.stack((st * sp, ct, -st * cp), dim=-1)


def env_tv_ref(base, dirs):
    """train.py:405-424 with the cube lookup of oracle/torch_pbr_ref.py (nvdiffrast: parity unpinned)."""
    from oracle import torch_pbr_ref as pr
    env = pr.cube_sample(base.to(pr.DT), dirs.to(pr.DT))  # [H, W, 3]
    return ((env[1:] - env[:-1]) ** 2).mean() + ((env[:, 1:] - env[:, :-1]) ** 2).mean()
