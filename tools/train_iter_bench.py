"""A complete stage-2 training iteration of train.py (:247-523) on the C2 workload, optimizer included:

    activations (scene/gaussian_model.py getters) -> rasterizer + SSAO -> shade -> SSR -> L1
      + BRDF TV (train.py:387-402) + 0.001 lamb + 0.01 envmap TV (:405-420) -> backward
      -> Adam over the ten Gaussian groups + the light (train.py:517-523)

It reports iterations/s for (a) bench.py's metric step (post-activation leaves, no regularisers, no optimizer),
(b) the full iteration with the HIP glue (gi-gs_amd/losses.py, optim.py) and (c) the full iteration with the same
glue written as the reference writes it (torch op chains, torch.optim.Adam).  Dataset loading, densification and
logging are outside all three.

    python tools/train_iter_bench.py [--steps 60] [--warmup 10]
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gi-gs_amd"))

import activations  # noqa: E402
import bench  # noqa: E402  (workload construction only)
import losses  # noqa: E402
import optim  # noqa: E402
import pipeline  # noqa: E402
import scenes  # noqa: E402
from pbr.texture import cube_texture  # noqa: E402


def logit(x):
    x = np.clip(x, 1e-6, 1 - 1e-6)
    return np.log(x / (1 - x))


def torch_masked_tv(mask, gt, pred):
    wh = torch.exp(-(gt[:, 1:, :] - gt[:, :-1, :]).abs().mean(dim=0, keepdim=True))
    ww = torch.exp(-(gt[:, :, 1:] - gt[:, :, :-1]).abs().mean(dim=0, keepdim=True))
    th = torch.pow(pred[:, 1:, :] - pred[:, :-1, :], 2)
    tw = torch.pow(pred[:, :, 1:] - pred[:, :, :-1], 2)
    m = mask.float()
    return (th * wh * (m[:, 1:, :] * m[:, :-1, :])).mean() + (tw * ww * (m[:, :, 1:] * m[:, :, :-1])).mean()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gaussians", type=int, default=300000)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--sh-degree", type=int, default=2)
    ap.add_argument("--start", type=int, default=8, help="GI march start (>= 16 = empty march)")
    ap.add_argument("--only", default="", help="stage1_hip | stage2_hip: run one variant (for profiling)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gi = dict(scenes.GI_DEFAULTS, start=a.start)
    sc = scenes.surface_scene(P=a.gaussians, sh_degree=a.sh_degree, seed=0)
    n_views = 64
    cams = [scenes.orbit_camera(i, n_views, a.res, a.res, radius=3.5) for i in range(n_views)]
    cams_t = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    H = W = a.res
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, W, device=dev), indexing="ij")
    gt_image = torch.stack([0.5 + 0.3 * torch.sin(6 * xx), 0.5 + 0.3 * torch.cos(5 * yy), 0.4 + 0.2 * xx * yy])
    rays = pipeline.canonical_rays(cams[0], dev)
    view_dirs = [pipeline.view_dirs_for(c, rays, dev) for c in cams_t]
    envmap_dirs = losses.get_envmap_dirs(device=dev)

    def raw_params():
        t = lambda x: torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev))  # noqa: E731
        return dict(xyz=t(sc["means3D"]), f_dc=t(sc["shs"][:, :1]), f_rest=t(sc["shs"][:, 1:]),
                    opacity=t(logit(sc["opacities"])), normal=t(sc["normal"]), albedo=t(logit(sc["albedo"])),
                    roughness=t(logit(sc["roughness"])), metallic=t(logit(sc["metallic"])),
                    scaling=t(np.log(sc["scales"])), rotation=t(sc["rotations"]))

    def activate(r):  # scene/gaussian_model.py:178-263
        return dict(means3D=r["xyz"], shs=torch.cat((r["f_dc"], r["f_rest"]), dim=1), opacities=torch.sigmoid(r["opacity"]),
                    normal=F.normalize(r["normal"], dim=-1), albedo=torch.sigmoid(r["albedo"]),
                    roughness=torch.sigmoid(r["roughness"]), metallic=torch.sigmoid(r["metallic"]),
                    scales=torch.exp(r["scaling"]), rotations=F.normalize(r["rotation"]))

    lrs = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=2.5e-3 / 20, opacity=0.05, normal=0.05, albedo=0.05, roughness=0.05,
               metallic=0.05, scaling=5e-3, rotation=1e-3)  # arguments/__init__.py defaults

    def run(mode):
        light, lut = bench.make_light(dev, "hip")
        stepper = pipeline.Stage2Step(light, lut, gi, a.sh_degree, graphs=True, fused=True)
        if mode == "metric":
            g = {k: torch.from_numpy(sc[k]).to(dev).requires_grad_(True) for k in bench.PARAM_KEYS}
            leaves = list(g.values()) + list(light.parameters())

            def it(i):
                for p in leaves:
                    p.grad = None
                stepper(cams_t[i % n_views], g, gt_image, view_dirs[i % n_views])
        else:
            r = raw_params()
            groups = [{"params": [r[k]], "lr": lrs[k], "name": k} for k in lrs]
            hip = mode == "hip"
            Opt = optim.FusedAdam if hip else torch.optim.Adam
            opt = Opt(groups, lr=0.0, eps=1e-15)
            lopt = Opt([{"params": list(light.parameters()), "lr": 0.05}], lr=0.05)
            leaves = list(r.values()) + list(light.parameters())

            def reg(normal_map, albedo_map, roughness_map, metallic_map):
                mask = (normal_map.detach() != 0).all(0, keepdim=True)
                brdf = torch.cat([albedo_map, roughness_map * 0.96 + 0.04, metallic_map], dim=0)
                if hip:
                    return losses.get_masked_tv_loss(mask, gt_image, brdf) + 0.01 * losses.env_tv_loss(light.base, envmap_dirs)
                env = cube_texture(light.base, envmap_dirs)  # the lookup itself has no torch formulation here
                env_tv = torch.pow(env[1:] - env[:-1], 2).mean() + torch.pow(env[:, 1:] - env[:, :-1], 2).mean()
                return torch_masked_tv(mask, gt_image, brdf) + 0.01 * env_tv

            def it(i):
                stepper(cams_t[i % n_views], activations.activate(r) if hip else activate(r), gt_image,
                        view_dirs[i % n_views], extra_loss=reg)
                opt.step()
                lopt.step()
                for p in leaves:  # zero_grad(set_to_none=True), train.py:518-522
                    p.grad = None
                with torch.no_grad():  # train.py:427 wraps the optimizer section
                    light.clamp_(min=0.0)
        for i in range(a.warmup):
            it(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            it(a.warmup + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        return {"ms_per_iteration": round(dt * 1e3, 3), "iterations_per_s": round(1.0 / dt, 1)}

    def run_stage1(hip):
        """train.py:318-331 + :517-518: render -> L1 + D-SSIM + masked normal L1 + normal TV -> backward -> Adam."""
        r = raw_params()
        Opt = optim.FusedAdam if hip else torch.optim.Adam
        opt = Opt([{"params": [r[k]], "lr": lrs[k], "name": k} for k in lrs], lr=0.0, eps=1e-15)
        bg = torch.zeros(3, device=dev)
        taps = torch.tensor([math.exp(-((i - 5) ** 2) / 4.5) for i in range(11)])
        taps = taps / taps.sum()
        win = torch.outer(taps, taps)[None, None].expand(3, 1, 11, 11).contiguous().to(dev)

        def torch_ssim(x, y):
            blur = lambda t: F.conv2d(t[None], win, padding=5, groups=3)[0]  # noqa: E731
            mx, my = blur(x), blur(y)
            vx, vy, cxy = blur(x * x) - mx * mx, blur(y * y) - my * my, blur(x * y) - mx * my
            return (((2 * mx * my + 1e-4) * (2 * cxy + 9e-4)) / ((mx * mx + my * my + 1e-4) * (vx + vy + 9e-4))).mean()

        def it(i):
            res = pipeline.render(cams_t[i % n_views], activations.activate(r) if hip else activate(r), a.sh_degree, bg, gi,
                                  fused_post=hip)
            image, nm, nfd, mask = res["render"], res["normal_map"], res["normal_map_from_depth"], res["normal_from_depth_mask"]
            if hip:
                loss, _, _ = losses.stage1_loss(image, gt_image, nm, nfd, mask, 0.2)
            else:
                loss = 0.8 * (image - gt_image).abs().mean() + 0.2 * (1.0 - torch_ssim(image, gt_image))
                loss = loss + F.l1_loss(nm[:, mask], nfd[:, mask])
                wh = torch.exp(-(gt_image[:, 1:, :] - gt_image[:, :-1, :]).abs().mean(dim=0, keepdim=True))
                ww = torch.exp(-(gt_image[:, :, 1:] - gt_image[:, :, :-1]).abs().mean(dim=0, keepdim=True))
                loss = loss + (torch.pow(nm[:, 1:, :] - nm[:, :-1, :], 2) * wh).mean() + (
                    torch.pow(nm[:, :, 1:] - nm[:, :, :-1], 2) * ww).mean()
            loss.backward()
            opt.step()
            for p in r.values():
                p.grad = None
        for i in range(a.warmup):
            it(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            it(a.warmup + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        return {"ms_per_iteration": round(dt * 1e3, 3), "iterations_per_s": round(1.0 / dt, 1)}

    out = {"workload": f"C2 P={a.gaussians} {a.res}x{a.res} sh{a.sh_degree} start={a.start}", "steps": a.steps}
    if a.only:
        out[a.only] = run_stage1(True) if a.only == "stage1_hip" else run("hip")
        print(json.dumps(out))
        return
    out["stage1_full_iteration_hip_glue"] = run_stage1(True)
    out["stage1_full_iteration_torch_glue"] = run_stage1(False)
    out["metric_step_no_optimizer"] = run("metric")
    out["full_iteration_hip_glue"] = run("hip")
    out["full_iteration_torch_glue"] = run("torch")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
