"""Times the training-loop glue kernels on the GPU against the torch-op formulation the
reference uses for the same arithmetic, and prices them against the HBM roofline.

    python tools/glue_bench.py [--size 800] [--P 300000] [--M 9] [--iters 50]

Prints one JSON line: per kernel {ms, algorithmic GB/s, fraction of 8 TB/s, torch_ms}.  Algorithmic bytes:
l1_ssim_fwd 20 B and l1_ssim_bwd 24 B per pixel-channel (image, gt, three derivative planes, gradient);
tv_fwd 4(3+C) B and tv_bwd 4(3+2C) B per pixel; masked_l1 fwd (8C+1) B, bwd (12C+1) B per pixel;
adam 28 B per parameter (read p, g, m, v; write p, m, v).
"""
import argparse
import json
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gi-gs_amd"))

import gigs_lib  # noqa: E402
import losses  # noqa: E402
import optim  # noqa: E402

HBM_PEAK = 8.0e12


def torch_ssim(x, y, win):
    C = x.shape[0]
    blur = lambda t: F.conv2d(t[None], win, padding=5, groups=C)[0]  # noqa: E731
    mx, my = blur(x), blur(y)
    vx, vy, cxy = blur(x * x) - mx * mx, blur(y * y) - my * my, blur(x * y) - mx * my
    return (((2 * mx * my + 1e-4) * (2 * cxy + 9e-4)) / ((mx * mx + my * my + 1e-4) * (vx + vy + 9e-4))).mean()


def torch_tv(gt, pred, mask=None):
    wh = torch.exp(-(gt[:, 1:, :] - gt[:, :-1, :]).abs().mean(dim=0, keepdim=True))
    ww = torch.exp(-(gt[:, :, 1:] - gt[:, :, :-1]).abs().mean(dim=0, keepdim=True))
    th = torch.pow(pred[:, 1:, :] - pred[:, :-1, :], 2)
    tw = torch.pow(pred[:, :, 1:] - pred[:, :, :-1], 2)
    if mask is not None:
        wh = wh * mask[:, 1:, :] * mask[:, :-1, :]
        ww = ww * mask[:, :, 1:] * mask[:, :, :-1]
    return (th * wh).mean() + (tw * ww).mean()


def time_ms(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def kernel_ms(fn, iters, warmup=5):
    """Per-stage kernel time from the library's own HIP events (gigs_profile_*)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    with gigs_lib.profile() as prof:
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
    return {k: v[0] / v[1] for k, v in prof.stages.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--P", type=int, default=300000)
    ap.add_argument("--M", type=int, default=9)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    H = W = a.size
    N = H * W
    g = torch.Generator().manual_seed(0)
    gt = torch.rand(3, H, W, generator=g).to(dev)
    img = (gt + 0.05 * torch.randn(3, H, W, generator=g).to(dev)).clamp(0, 1)
    brdf = torch.rand(5, H, W, generator=g).to(dev)
    nrm = F.normalize(torch.randn(3, H, W, generator=g), dim=0).to(dev)
    nfd = F.normalize(torch.randn(3, H, W, generator=g), dim=0).to(dev)
    mask = (torch.rand(1, H, W, generator=g) > 0.2).to(dev)
    taps = torch.tensor([math.exp(-((i - 5) ** 2) / 4.5) for i in range(11)])
    taps = taps / taps.sum()
    win = torch.outer(taps, taps)[None, None].expand(3, 1, 11, 11).contiguous().to(dev)
    res = {}

    def hip_stage1():
        x = img.detach().requires_grad_(True)
        n = nrm.detach().requires_grad_(True)
        loss, _, _ = losses.stage1_loss(x, gt, n, nfd, mask[0], 0.2)
        loss.backward()

    def torch_stage1():
        x = img.detach().requires_grad_(True)
        n = nrm.detach().requires_grad_(True)
        loss = 0.8 * (x - gt).abs().mean() + 0.2 * (1.0 - torch_ssim(x, gt, win))
        loss = loss + F.l1_loss(n[:, mask[0]], nfd[:, mask[0]]) + torch_tv(gt, n)
        loss.backward()

    def hip_brdf_tv():
        p = brdf.detach().requires_grad_(True)
        losses.get_masked_tv_loss(mask, gt, p).backward()

    def torch_brdf_tv():
        p = brdf.detach().requires_grad_(True)
        torch_tv(gt, p, mask.float()).backward()

    k = kernel_ms(hip_stage1, a.iters)
    k.update({f"brdf_{n}": v for n, v in kernel_ms(hip_brdf_tv, a.iters).items()})
    bytes_ = {"l1_ssim_fwd": 20 * 3 * N, "l1_ssim_bwd": 24 * 3 * N, "tv_loss_fwd": 4 * (3 + 3) * N,
              "tv_loss_bwd": 4 * (3 + 6) * N, "masked_l1": (8 * 3 + 1 + 12 * 3 + 1) * N / 2,
              "brdf_tv_loss_fwd": 4 * (3 + 5) * N + 4 * N, "brdf_tv_loss_bwd": 4 * (3 + 10) * N + 4 * N}
    for name, ms in k.items():
        gbs = bytes_[name] / (ms * 1e-3) / 1e9 if name in bytes_ else None
        res[name] = {"ms": round(ms, 4), "GBps": None if gbs is None else round(gbs, 1),
                     "hbm_frac": None if gbs is None else round(gbs * 1e9 / HBM_PEAK, 4)}
    res["stage1_loss_fwd_bwd"] = {"hip_wall_ms": round(time_ms(hip_stage1, a.iters), 4),
                                  "torch_wall_ms": round(time_ms(torch_stage1, a.iters), 4)}
    res["brdf_masked_tv_fwd_bwd"] = {"hip_wall_ms": round(time_ms(hip_brdf_tv, a.iters), 4),
                                     "torch_wall_ms": round(time_ms(torch_brdf_tv, a.iters), 4)}

    # Adam over the reference's ten groups (scene/gaussian_model.py:325-344)
    P, M = a.P, a.M
    shapes = [(P, 3), (P, 1, 3), (P, M - 1, 3), (P, 1), (P, 3), (P, 3), (P, 1), (P, 1), (P, 3), (P, 4)]
    n_param = sum(math.prod(s) for s in shapes)

    def make(opt_cls, **kw):
        ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
        for p in ps:
            p.grad = torch.randn_like(p)
        return ps, opt_cls([{"params": [p], "lr": 1e-3} for p in ps], lr=0.0, eps=1e-15, **kw)

    ps, fo = make(optim.FusedAdam)
    k = kernel_ms(lambda: fo.step(), a.iters)
    ms = k["adam_step"]
    res["adam_step"] = {"ms": round(ms, 4), "params": n_param, "GBps": round(28 * n_param / (ms * 1e-3) / 1e9, 1),
                        "hbm_frac": round(28 * n_param / (ms * 1e-3) / HBM_PEAK, 4),
                        "hip_wall_ms": round(time_ms(lambda: fo.step(), a.iters), 4)}
    res["adam_step+zero_grad"] = {"hip_wall_ms": round(time_ms(lambda: fo.step(zero_grad=True), a.iters), 4)}
    del ps, fo
    for label, kw in (("torch_adam_foreach", {}), ("torch_adam_fused", {"fused": True}),
                      ("torch_adam_single_tensor", {"foreach": False})):
        try:
            ps, to = make(torch.optim.Adam, **kw)

            def step():
                to.step()
                to.zero_grad(set_to_none=False)

            res[label + "+zero_grad"] = {"wall_ms": round(time_ms(step, a.iters), 4)}
            del ps, to
        except Exception as e:  # noqa: BLE001
            res[label] = {"error": str(e)[:80]}
    print(json.dumps({"size": a.size, "P": P, "M": M, "glue": res}))


if __name__ == "__main__":
    main()
