"""GPU: the pieces together.  A small scene is rendered to make multi-view targets, its parameters are perturbed, and a
stage-1 style optimisation (train.py:247-523 in miniature: activations -> GaussianRasterizer -> L1 + D-SSIM ->
backward -> densification statistics -> FusedAdam, with one densify_and_prune on the way) has to bring the renders
back towards the targets.  This is the "PSNR of the final image" half of the metric as a regression
test of the whole chain rather than of one kernel."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _psnr(a, b):
    return -10.0 * math.log10(float(((a - b) ** 2).mean()) + 1e-12)


def test_stage1_optimisation_recovers_a_perturbed_scene():
    import activations
    import densify
    import losses
    import optim
    import pipeline
    import scenes
    assert torch.cuda.is_available(), "these tests need the MI355X"
    dev = torch.device("cuda:0")
    H = W = 96
    sc = scenes.surface_scene(P=6000, sh_degree=1, seed=2, scale_mu=0.035)
    n_views = 6
    cams = [scenes.orbit_camera(i, n_views, W, H, radius=3.5) for i in range(n_views)]
    cams = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    gi = dict(scenes.GI_DEFAULTS, start=16)  # empty march (README setting): stage 1 does not use the occlusion
    bg = torch.zeros(3, device=dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    logit = lambda x: np.log(np.clip(x, 1e-6, 1 - 1e-6) / (1 - np.clip(x, 1e-6, 1 - 1e-6)))  # noqa: E731
    truth = dict(xyz=t(sc["means3D"]), f_dc=t(sc["shs"][:, :1]), f_rest=t(sc["shs"][:, 1:]), opacity=t(logit(sc["opacities"])),
                 normal=t(sc["normal"]), albedo=t(logit(sc["albedo"])), roughness=t(logit(sc["roughness"])),
                 metallic=t(logit(sc["metallic"])), scaling=t(np.log(sc["scales"])), rotation=t(sc["rotations"]))
    with torch.no_grad():
        targets = [pipeline.render(c, activations.activate(truth), 1, bg, gi)["render"].clamp(0, 1) for c in cams]
    g = torch.Generator(device=dev).manual_seed(0)
    raw = {k: torch.nn.Parameter(v.clone()) for k, v in truth.items()}
    with torch.no_grad():  # a wrong starting point: colours, positions, opacities and sizes off
        raw["f_dc"].add_(torch.randn(raw["f_dc"].shape, device=dev, generator=g) * 0.6)
        raw["f_rest"].zero_()
        raw["xyz"].add_(torch.randn(raw["xyz"].shape, device=dev, generator=g) * 0.01)
        raw["opacity"].add_(torch.randn(raw["opacity"].shape, device=dev, generator=g) * 0.5)
        raw["scaling"].add_(0.15)
    lrs = dict(xyz=1.6e-4, f_dc=2.5e-2, f_rest=2.5e-3, opacity=0.05, normal=0.0, albedo=0.0, roughness=0.0, metallic=0.0,
               scaling=5e-3, rotation=1e-3)
    opt = optim.FusedAdam([{"params": [raw[k]], "lr": lrs[k], "name": k} for k in densify.NAMES], lr=0.0, eps=1e-15)
    stats = densify.DensifyState(raw["xyz"].shape[0], dev)

    def evaluate():
        with torch.no_grad():
            return float(np.mean([_psnr(pipeline.render(c, activations.activate(raw), 1, bg, gi)["render"].clamp(0, 1), tg)
                                  for c, tg in zip(cams, targets)]))

    psnr0 = evaluate()
    first = last = None
    P_before = raw["xyz"].shape[0]
    for it in range(1, 241):
        v = it % n_views
        res = pipeline.render(cams[v], activations.activate(raw), 1, bg, gi)
        loss = losses.l1_ssim_loss(res["render"], targets[v], 0.2)
        loss.backward()
        with torch.no_grad():
            densify.add_densification_stats(stats, res["viewspace_points"].grad, res["radii"])
            if it == 120:  # one densification in the middle of the run (train.py:500-507)
                new, stats = densify.densify_and_prune(opt, stats, 2e-4, 0.005, 4.0, None, generator=g)
                raw = dict(new)
            else:
                opt.step()
            for p in raw.values():
                p.grad = None
        if it <= 6:
            first = (first or 0.0) + loss.item() / 6
        if it > 234:
            last = (last or 0.0) + loss.item() / 6
    psnr1 = evaluate()
    print(f"\nstage-1 miniature: loss {first:.4f} -> {last:.4f}, PSNR {psnr0:.2f} -> {psnr1:.2f} dB, P {P_before} -> {raw['xyz'].shape[0]}")
    assert raw["xyz"].shape[0] != P_before                      # the densification changed the set
    assert all(torch.isfinite(p).all() for p in raw.values())
    assert last < 0.6 * first, (first, last)                      # the objective went down ...
    assert psnr1 > psnr0 + 3.0, (psnr0, psnr1)                    # ... and the renders moved towards the targets
