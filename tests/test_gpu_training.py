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


def test_stage2_optimisation_recovers_materials_and_light():
    """Stage 2 end to end (train.py:330-420, 517-522 in miniature) on the fast path: targets are stage-2 images
    (direct + indirect) of a scene under a known light; albedo / roughness / metallic and the light are perturbed; the
    COMPLETE iteration -- activations, rasterizer + SSAO, build_mips, shade, SSR, L1 + lamb + masked BRDF TV + envmap TV,
    backward through shade / SSR closed form / GGX pre-filter, FusedAdam on the Gaussians and on the light, clamp --
    replayed from three hipGraphs has to bring the images back.  The first iterations are also run through the eager
    formulation of the same iteration: both must take the same trajectory."""
    import activations
    import pbr
    import pipeline
    import scenes
    import train_iteration as ti
    dev = torch.device("cuda:0")
    H = W = 112
    deg = 1
    sc = scenes.surface_scene(P=6000, sh_degree=deg, seed=4, scale_mu=0.035)
    n_views = 6
    cams = [scenes.orbit_camera(i, n_views, W, H, radius=3.5) for i in range(n_views)]
    cams = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    gi = scenes.GI_DEFAULTS
    lut = pbr.get_brdf_lut().to(dev)
    rays = pipeline.canonical_rays(cams[0], dev)
    vds = [pipeline.view_dirs_for(c, rays, dev) for c in cams]

    def make_light(seed, flat=None):
        torch.manual_seed(seed)
        light = pbr.CubemapLight(base_res=64, device=dev)
        with torch.no_grad():
            if flat is not None:
                light.base.fill_(flat)
            else:  # a smooth, coloured environment: bright "sky" on one side
                d = torch.stack(torch.meshgrid(torch.linspace(-1, 1, 64, device=dev), torch.linspace(-1, 1, 64, device=dev),
                                               indexing="ij"), -1)
                for f in range(6):
                    light.base[f] = (0.35 + 0.3 * torch.sin(2.0 * d[..., :1] + f) + 0.25 * torch.cos(1.5 * d[..., 1:] - f)
                                     ) * torch.tensor([1.0, 0.85, 0.6], device=dev) * (1.6 if f == 2 else 0.8)
                light.base.clamp_(min=0.02)
        return light

    truth_light = make_light(0)
    truth = {k: v.detach() for k, v in ti.raw_from_scene(sc, dev).items()}

    def images(raw, light):
        step = pipeline.Stage2Step(light, lut, gi, deg, fused=True, graphs=False)
        out = []
        for c, vd in zip(cams, vds):
            g = {k: v.detach().requires_grad_(True) for k, v in activations.activate(raw).items()}
            out.append(step(c, g, torch.zeros(3, H, W, device=dev), vd)["render_rgb"].detach().clone())
        light.base.grad = None
        return out

    targets = images(truth, truth_light)

    def perturbed():
        gen = torch.Generator(device=dev).manual_seed(1)
        raw = {k: torch.nn.Parameter(v.clone()) for k, v in truth.items()}
        with torch.no_grad():
            for k, s in (("albedo", 1.0), ("roughness", 1.0), ("metallic", 1.0)):
                raw[k].add_(torch.randn(raw[k].shape, device=dev, generator=gen) * s)
        return raw, make_light(0, flat=0.5)

    def psnr_of(raw, light):
        return float(np.mean([_psnr(a, b) for a, b in zip(images({k: v.detach() for k, v in raw.items()}, light), targets)]))

    # (1) the same five iterations, eager and from the graphs: one trajectory
    traj = {}
    for graphs in (False, True):
        raw, light = perturbed()
        tr = ti.Stage2Trainer(raw, light, lut, gi, deg, graphs=graphs)
        losses_ = []
        for it in range(5):
            losses_.append(float(tr.iteration(cams[it % n_views], targets[it % n_views], vds[it % n_views])["loss"]))
        torch.cuda.synchronize()
        assert all(p.grad is None for p in raw.values()) and light.base.grad is None  # zero_grad(set_to_none=True)
        traj[graphs] = (losses_, {k: v.detach().clone() for k, v in raw.items()}, light.base.detach().clone(), tr)
    assert traj[True][3].stepper.whole is not None and traj[True][3].stepper.whole.go is not None
    for a, b in zip(traj[False][0], traj[True][0]):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (traj[False][0], traj[True][0])
    for k in ("albedo", "roughness", "metallic"):
        d = (traj[False][1][k] - traj[True][1][k]).abs().max().item()
        assert d <= 2e-3, (k, d)  # five Adam steps of 0.05 each moved them by up to 0.25
    assert (traj[False][2] - traj[True][2]).abs().max().item() <= 2e-3
    for k in ("xyz", "scaling", "rotation", "opacity", "f_dc", "normal"):  # stage 2 reaches none of these (exact zeros)
        assert torch.equal(traj[True][1][k], truth[k]), k
    assert int(traj[True][3].optimizer.state[traj[True][3].raw["albedo"]]["step"]) == 5
    # ... and the same iteration written as the reference writes it (torch op chains for the getters and the glue between
    # the operators, torch.optim.Adam): the op-by-op formulation is the checker of the fused one
    raw, light = perturbed()
    tr = ti.Stage2Trainer(raw, light, lut, gi, deg, glue="torch")
    losses_t = [float(tr.iteration(cams[it % n_views], targets[it % n_views], vds[it % n_views])["loss"]) for it in range(5)]
    torch.cuda.synchronize()
    for a, b in zip(traj[False][0], losses_t):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), (traj[False][0], losses_t)
    for k in ("albedo", "roughness", "metallic"):
        d = (traj[False][1][k] - raw[k].detach()).abs().max().item()
        assert d <= 5e-3, ("torch glue", k, d)

    # (2) the optimisation itself, from the graphs
    raw, light = perturbed()
    psnr0 = psnr_of(raw, light)
    tr = ti.Stage2Trainer(raw, light, lut, gi, deg, graphs=True)
    first = last = 0.0
    n_it = 360
    for it in range(n_it):
        v = it % n_views
        loss = tr.iteration(cams[v], targets[v], vds[v])["loss"]
        if it < 6:
            first += float(loss) / 6
        if it >= n_it - 6:
            last += float(loss) / 6
    torch.cuda.synchronize()
    psnr1 = psnr_of(raw, light)
    print(f"\nstage-2 miniature: loss {first:.4f} -> {last:.4f}, PSNR {psnr0:.2f} -> {psnr1:.2f} dB, "
          f"light mean {float(light.base.detach().mean()):.3f} (truth {float(truth_light.base.detach().mean()):.3f})")
    assert tr.stepper.whole is not None and tr.stepper.whole.recaptures == 1
    assert all(torch.isfinite(p).all() for p in raw.values()) and torch.isfinite(light.base).all()
    assert float(light.base.min()) >= 0.0                      # cubemap.clamp_(min=0)
    assert last < 0.5 * first, (first, last)
    assert psnr1 > psnr0 + 4.0, (psnr0, psnr1)


def test_stage2_trainer_follows_pruning_and_learning_rate_changes():
    """The training loop's other moving parts around the captured iteration: a prune in the middle (densify.prune_points
    replaces every parameter tensor and its Adam moments, as the reference's prune_points does) -> one re-capture, the moments
    of the surviving rows carried over, training continues; a learning-rate change (update_learning_rate) reaches the
    captured Adam launch without a re-capture."""
    import densify
    import pbr
    import pipeline
    import scenes
    import train_iteration as ti
    dev = torch.device("cuda:0")
    H = W = 96
    sc = scenes.surface_scene(P=5000, sh_degree=1, seed=8, scale_mu=0.035)
    cams = [scenes.orbit_camera(i, 4, W, H, radius=3.5) for i in range(4)]
    cams = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    gi = scenes.GI_DEFAULTS
    lut = pbr.get_brdf_lut().to(dev)
    rays = pipeline.canonical_rays(cams[0], dev)
    vds = [pipeline.view_dirs_for(c, rays, dev) for c in cams]
    torch.manual_seed(1)
    gts = [torch.rand(3, H, W, device=dev) * 0.6 for _ in cams]
    light = pbr.CubemapLight(base_res=64, device=dev)
    raw = ti.raw_from_scene(sc, dev)
    tr = ti.Stage2Trainer(raw, light, lut, gi, 1, graphs=True)
    for it in range(4):
        tr.iteration(cams[it % 4], gts[it % 4], vds[it % 4])
    torch.cuda.synchronize()
    assert tr.stepper.whole.recaptures == 1
    # (1) learning rate: zero for albedo from now on -> albedo stops moving, no re-capture
    before = raw["albedo"].detach().clone()
    m_before = tr.optimizer.state[raw["albedo"]]["exp_avg"].clone()
    tr.set_lr("albedo", 0.0)
    tr.iteration(cams[0], gts[0], vds[0])
    torch.cuda.synchronize()
    assert torch.equal(raw["albedo"].detach(), before) and tr.stepper.whole.recaptures == 1
    assert not torch.equal(tr.optimizer.state[raw["albedo"]]["exp_avg"], m_before)  # the moments still follow the gradients
    tr.set_lr("albedo", 0.05)
    # (2) prune every third Gaussian
    P0 = raw["xyz"].shape[0]
    mask = torch.zeros(P0, dtype=torch.bool, device=dev)
    mask[::3] = True
    keep_rows = raw["roughness"].detach()[~mask].clone()
    keep_m = tr.optimizer.state[raw["roughness"]]["exp_avg"][~mask].clone()
    steps_before = int(tr.optimizer.state[raw["roughness"]]["step"])
    new, _ = densify.prune_points(tr.optimizer, densify.DensifyState(P0, dev), mask)
    tr.replace_parameters(dict(new))
    P1 = tr.raw["xyz"].shape[0]
    assert P1 == P0 - int(mask.sum()) and torch.equal(tr.raw["roughness"].detach(), keep_rows)
    assert torch.equal(tr.optimizer.state[tr.raw["roughness"]]["exp_avg"], keep_m)
    losses_ = []
    for it in range(6):
        losses_.append(float(tr.iteration(cams[it % 4], gts[it % 4], vds[it % 4])["loss"]))
    torch.cuda.synchronize()
    assert tr.stepper.whole.recaptures == 2                      # one re-capture for the new tensors, then replays
    assert int(tr.optimizer.state[tr.raw["roughness"]]["step"]) == steps_before + 6
    assert all(np.isfinite(losses_)) and all(torch.isfinite(p).all() for p in tr.raw.values())
    assert not torch.equal(tr.raw["roughness"].detach(), keep_rows)  # the pruned model keeps training
    with pytest.raises(ValueError):
        tr.replace_parameters({k: v.detach().clone() for k, v in tr.raw.items()})  # not the optimizer's tensors


def test_stage1_iterations_from_the_graphs_follow_the_eager_trajectory():
    """train_iteration.Stage1Trainer: the complete stage-1 iteration (activations, rasterizer + filters, L1 + D-SSIM + masked
    normal L1 + normal TV, backward, FusedAdam) replayed from three hipGraphs takes the eager formulation's trajectory, and the
    objective goes down."""
    import scenes
    import train_iteration as ti
    dev = torch.device("cuda:0")
    H = W = 96
    sc = scenes.surface_scene(P=5000, sh_degree=1, seed=12, scale_mu=0.035)
    cams = [scenes.orbit_camera(i, 4, W, H, radius=3.5) for i in range(4)]
    cams = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    gi = dict(scenes.GI_DEFAULTS, start=16)  # empty march: stage 1 does not read the occlusion
    torch.manual_seed(2)
    gts = [torch.rand(3, H, W, device=dev) * 0.7 for _ in cams]
    runs = {}
    for graphs in (False, True):
        raw = ti.raw_from_scene(sc, dev)
        tr = ti.Stage1Trainer(raw, gi, 1, graphs=graphs)
        losses_ = [float(tr.iteration(cams[it % 4], gts[it % 4])["loss"]) for it in range(12)]
        torch.cuda.synchronize()
        assert all(p.grad is None for p in raw.values())
        runs[graphs] = (losses_, {k: v.detach().clone() for k, v in raw.items()}, tr)
    assert runs[True][2].stepper.whole is not None and runs[True][2].stepper.whole.go is not None
    assert runs[True][2].stepper.whole.recaptures == 1
    for a, b in zip(runs[False][0], runs[True][0]):
        assert abs(a - b) <= 5e-5 * max(1.0, abs(a)), (runs[False][0], runs[True][0])
    for k, lim in (("f_dc", 5e-4), ("opacity", 5e-3), ("xyz", 5e-5), ("normal", 5e-3), ("scaling", 1e-3)):
        d = (runs[False][1][k] - runs[True][1][k]).abs().max().item()
        assert d <= lim, (k, d)
    first, last = np.mean(runs[True][0][:4]), np.mean(runs[True][0][-4:])
    assert last < first, (first, last)
    # compute_occlusion=False (no SSAO march although the GI settings ask for one: stage 1 does not read it): the same updates
    raw = ti.raw_from_scene(sc, dev)
    tr = ti.Stage1Trainer(raw, scenes.GI_DEFAULTS, 1, graphs=True, compute_occlusion=False)
    lean = [float(tr.iteration(cams[it % 4], gts[it % 4])["loss"]) for it in range(12)]
    torch.cuda.synchronize()
    for a, b in zip(lean, runs[True][0]):  # float atomics in the backward: equal to rounding, not bit for bit
        assert abs(a - b) <= 5e-5 * max(1.0, abs(a)), (lean, runs[True][0])
    for k, lim in (("f_dc", 5e-4), ("opacity", 5e-3), ("xyz", 5e-5), ("normal", 5e-3), ("scaling", 1e-3)):
        assert (raw[k].detach() - runs[True][1][k]).abs().max().item() <= lim, k


@pytest.mark.parametrize("stage", [2, 1])
def test_trainer_with_the_gradient_slab_follows_the_plain_trajectory(stage):
    """data_parallel() moves the raw gradients into ONE dp.GradSlab -- born there through the gradient sinks of the
    activations / the rasterizer, the light's copied in by a node of the backward graph -- and makes the captured Adam
    launch read the slab's views.  With one rank the collective is skipped, so the trainer must take exactly
    the updates of a trainer without the slab: same losses, bit-identical parameters after six iterations from the graphs,
    the stage-2 stretch adjacent at the slab's end; a prune in between re-carves the slab around the new tensors."""
    import densify
    import pbr
    import pipeline
    import scenes
    import train_iteration as ti
    dev = torch.device("cuda:0")
    H = W = 96
    sc = scenes.surface_scene(P=5000, sh_degree=1, seed=8, scale_mu=0.035)
    cams = [scenes.orbit_camera(i, 4, W, H, radius=3.5) for i in range(4)]
    cams = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    gi = scenes.GI_DEFAULTS
    lut = pbr.get_brdf_lut().to(dev)
    rays = pipeline.canonical_rays(cams[0], dev)
    vds = [pipeline.view_dirs_for(c, rays, dev) for c in cams]
    torch.manual_seed(1)
    gts = [torch.rand(3, H, W, device=dev) * 0.6 for _ in cams]

    def run(with_slab):
        torch.manual_seed(3)
        raw = ti.raw_from_scene(sc, dev)
        if stage == 2:
            light = pbr.CubemapLight(base_res=64, device=dev)
            tr = ti.Stage2Trainer(raw, light, lut, gi, 1, graphs=True)
        else:
            light = None
            tr = ti.Stage1Trainer(raw, gi, 1, graphs=True)
        slab = tr.data_parallel() if with_slab else None
        it_ = (lambda i: tr.iteration(cams[i % 4], gts[i % 4], vds[i % 4])) if stage == 2 else (lambda i: tr.iteration(cams[i % 4], gts[i % 4]))
        losses_ = [float(it_(i)["loss"]) for i in range(3)]
        if with_slab:
            wsg = tr.stepper.whole
            assert wsg is not None and wsg.go is not None
            views = {v.data_ptr() for v in slab.views}
            assert all(gr is None or gr.data_ptr() in views for gr in wsg.grads), "a captured gradient lives outside the slab"
            if stage == 2:
                assert tr.slab_order[-4:] == ["albedo", "roughness", "metallic", "cubemap"]
        # a prune: new tensors, new slab, one re-capture
        P0 = tr.raw["xyz"].shape[0]
        mask = torch.zeros(P0, dtype=torch.bool, device=dev)
        mask[::4] = True
        new, _ = densify.prune_points(tr.optimizer, densify.DensifyState(P0, dev), mask)
        tr.replace_parameters(dict(new))
        if with_slab:
            assert tr.slab is not slab and tr.slab.flat.numel() < slab.flat.numel()
        losses_ += [float(it_(i)["loss"]) for i in range(3, 6)]
        torch.cuda.synchronize()
        out = ({k: v.detach().clone() for k, v in tr.raw.items()}, None if light is None else light.base.detach().clone(), losses_)
        tr.close()
        return out

    plain, slabbed = run(False), run(True)
    for a, b in zip(plain[2], slabbed[2]):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (plain[2], slabbed[2])
    # the gradients are sums of float atomics (blend / shade backward): two runs agree to rounding, not bit for bit -- the
    # bar of the eager-vs-graphs comparison above (six Adam steps move a parameter by up to 6 lr)
    for k in plain[0]:
        d = (plain[0][k] - slabbed[0][k]).abs().max().item()
        assert d <= 2e-3, (k, d)
    if stage == 2:
        assert (plain[1] - slabbed[1]).abs().max().item() <= 2e-3
        for k in ("xyz", "scaling", "rotation", "opacity", "f_dc", "normal"):  # exact zeros either way: stage 2 reaches none
            assert torch.equal(plain[0][k], slabbed[0][k]), k


def test_frozen_geometry_cache_replays_the_tile_lists_and_follows_the_plain_trainer():
    """pipeline.GeometryCache (a secondary figure): in stage 2 the loss reaches materials and light only,
    so once Adam's residual momentum has died out the geometry stops changing bit for bit and, per view, the tile lists and
    the occlusion plane are reused (gigs_ctx_set_reuse_binning; no binning, no SSAO march).  Here the optimizer starts WITH
    momentum on positions, scales and opacities (as after stage 1): geometry drifts for a few dozen iterations -- every one
    of them must be recorded, none replayed from a stale entry -- and then freezes.  The cached trainer takes the updates of
    the plain one (losses and parameters to the rounding of the float-atomic gradient sums), geometry bit for bit; a replayed
    forward equals the recorded one bit for bit; a learning-rate kick that moves geometry again drops every entry."""
    import pbr
    import pipeline
    import scenes
    import train_iteration as ti
    dev = torch.device("cuda:0")
    H = W = 96
    sc = scenes.surface_scene(P=5000, sh_degree=1, seed=8, scale_mu=0.035)
    n_views = 4
    cams = [scenes.orbit_camera(i, n_views, W, H, radius=3.5) for i in range(n_views)]
    cams = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    gi = scenes.GI_DEFAULTS
    lut = pbr.get_brdf_lut().to(dev)
    rays = pipeline.canonical_rays(cams[0], dev)
    vds = [pipeline.view_dirs_for(c, rays, dev) for c in cams]
    torch.manual_seed(1)
    gts = [torch.rand(3, H, W, device=dev) * 0.6 for _ in cams]

    def make(cached):
        torch.manual_seed(3)
        raw = ti.raw_from_scene(sc, dev)
        light = pbr.CubemapLight(base_res=64, device=dev)
        tr = ti.Stage2Trainer(raw, light, lut, gi, 1, graphs=True, geometry_cache=cached)
        gen = torch.Generator(device=dev).manual_seed(5)
        for name, scale in (("xyz", 1e-3), ("scaling", 1e-3), ("opacity", 1e-2)):  # residual momentum of an earlier stage
            p = raw[name]
            tr.optimizer.state[p] = {"step": torch.tensor(200.0), "exp_avg": torch.randn(p.shape, device=dev, generator=gen) * scale,
                                     "exp_avg_sq": torch.full_like(p, scale * scale)}
        return tr, raw, light

    def run(tr, n, first=0):
        return [float(tr.iteration(cams[i % n_views], gts[i % n_views], vds[i % n_views])["loss"]) for i in range(first, first + n)]

    plain, raw_p, light_p = make(False)
    cached, raw_c, light_c = make(True)
    xyz0 = raw_c["xyz"].detach().clone()
    lp, lc = run(plain, 300), run(cached, 300)
    torch.cuda.synchronize()
    cache = cached.stepper.geom_cache
    st = dict(cache.stats)
    print("geometry cache:", st)
    assert not torch.equal(raw_c["xyz"].detach(), xyz0), "the test's momentum did not move the geometry at all"
    assert st["replayed"] >= 50 and st["hit_lists"] >= n_views, st   # the frozen phase replays; every view got its hit list
    assert st["recorded"] >= n_views and st["replayed"] < 300 - 20, st  # the drifting phase (>= 20 iterations) is never replayed
    # The gradients are sums of float atomics (blend / shade backward) and Adam runs with eps = 1e-15, so two PLAIN runs
    # already drift apart over hundreds of iterations; a second plain trainer is the yardstick for the late iterations, the
    # early ones (before rounding differences have been amplified) are held tight
    plain2, raw_q, _ = make(False)
    lq = run(plain2, 300)
    torch.cuda.synchronize()
    plain2.close()
    own = max(abs(a - b) for a, b in zip(lp, lq))
    print("max loss difference: cached vs plain %.2e, plain vs plain %.2e" % (max(abs(a - b) for a, b in zip(lp, lc)), own))
    for i, (a, b) in enumerate(zip(lp, lc)):
        assert abs(a - b) <= (2e-5 if i < 40 else max(2e-4, 4.0 * own)) * max(1.0, abs(a)), (i, a, b)
    for k in ("xyz", "scaling", "rotation", "opacity", "normal", "f_dc", "f_rest"):  # zero gradients: deterministic drift
        assert torch.equal(raw_p[k].detach(), raw_c[k].detach()), k
    for k in ("albedo", "roughness", "metallic"):
        bar = max(5e-3, 4.0 * (raw_p[k] - raw_q[k]).abs().max().item())
        assert (raw_p[k] - raw_c[k]).abs().max().item() <= bar, (k, bar)
    # a replayed forward equals a recorded one bit for bit (same parameters: learning rates zero, two visits of one view)
    for g_ in cached.optimizer.param_groups + cached.light_optimizer.param_groups:
        g_["lr"] = 0.0
    before = dict(cache.stats)
    a = cached.iteration(cams[1], gts[1], vds[1])
    a = {k: a[k].detach().clone() for k in ("render_rgb", "render_direct", "IRR")} | {"loss": float(a["loss"])}
    assert cache.stats["replayed"] == before["replayed"] + 1 and cache.stats["hit_lists"] == before["hit_lists"]  # a pure gather
    cache.invalidate()
    b = cached.iteration(cams[1], gts[1], vds[1])
    assert cache.stats["recorded"] == before["recorded"] + 1
    assert float(b["loss"]) == a["loss"]
    for k in ("render_rgb", "render_direct", "IRR"):
        assert torch.equal(a[k].view(torch.int32), b[k].detach().view(torch.int32)), k
    # geometry moves again (a position learning rate and fresh momentum): the next replay is found stale and repeated
    cached.iteration(cams[2], gts[2], vds[2])  # view 2 has an entry again
    p = raw_c["xyz"]
    cached.optimizer.state[p]["exp_avg"].normal_(0.0, 1e-3)
    cached.set_lr("xyz", 1e-3)
    r0 = cache.stats["repeated"]
    cached.iteration(cams[3], gts[3], vds[3])   # this update moves xyz
    cached.iteration(cams[2], gts[2], vds[2])   # optimistic replay of view 2 -> stale -> repeated as a recording
    torch.cuda.synchronize()
    assert cache.stats["repeated"] == r0 + 1 and cache.stats["invalidated"] >= 2, cache.stats
    plain.close()
    cached.close()


def test_declared_stage2_gradient_set_gives_the_same_updates_and_is_checked_on_the_device():
    """Stage2Trainer(materials_only=True), the default: the rasterizer's backward writes dL/d(albedo, roughness, metallic)
    only (gigs_ctx_set_materials_only), the activations' backward and Adam never see the 59 other floats per Gaussian --
    Adam updates those groups with g = 0.  (1) Same training as with every zero materialised: the zero-gradient groups --
    started WITH momentum, so that they do move -- bit for bit, materials / light / losses to the rounding of the float-atomic
    gradient sums.  (2) The premise is checked by every backward: a loss term that reaches the rasterized normals (outside
    the declared set) raises at the next iteration instead of training on zeros that are not zeros."""
    import pbr
    import pipeline
    import scenes
    import train_iteration as ti
    dev = torch.device("cuda:0")
    H = W = 96
    sc = scenes.surface_scene(P=5000, sh_degree=2, seed=8, scale_mu=0.035)
    n_views = 4
    cams = [scenes.orbit_camera(i, n_views, W, H, radius=3.5) for i in range(n_views)]
    cams = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    gi = scenes.GI_DEFAULTS
    lut = pbr.get_brdf_lut().to(dev)
    rays = pipeline.canonical_rays(cams[0], dev)
    vds = [pipeline.view_dirs_for(c, rays, dev) for c in cams]
    torch.manual_seed(1)
    gts = [torch.rand(3, H, W, device=dev) * 0.6 for _ in cams]
    zero_groups = ("xyz", "f_dc", "f_rest", "opacity", "normal", "scaling", "rotation")

    def make(declared):
        torch.manual_seed(3)
        raw = ti.raw_from_scene(sc, dev)
        light = pbr.CubemapLight(base_res=64, device=dev)
        tr = ti.Stage2Trainer(raw, light, lut, gi, 2, graphs=True, materials_only=declared)
        gen = torch.Generator(device=dev).manual_seed(5)
        for name in zero_groups:  # residual momentum of stage 1: these groups keep moving although their gradient is zero
            p = raw[name]
            tr.optimizer.state[p] = {"step": torch.tensor(200.0), "exp_avg": torch.randn(p.shape, device=dev, generator=gen) * 1e-3,
                                     "exp_avg_sq": torch.full_like(p, 1e-6)}
        return tr, raw, light

    def run(tr, n):
        return [float(tr.iteration(cams[i % n_views], gts[i % n_views], vds[i % n_views])["loss"]) for i in range(n)]

    full, raw_f, light_f = make(False)
    start = {k: raw_f[k].detach().clone() for k in zero_groups}
    lf = run(full, 10)
    decl, raw_d, light_d = make(True)
    ld = run(decl, 10)
    torch.cuda.synchronize()
    w = decl.stepper.whole
    assert w.viol_dev is not None and int(w.viol_dev.item()) == 0 and full.stepper.whole.viol_dev is None
    assert sum(g is None for g in w.grads) == len(zero_groups)  # nothing materialised for them
    for k in zero_groups:
        assert not torch.equal(raw_f[k].detach(), start[k]), k       # momentum did move the group
        assert torch.equal(raw_d[k].detach(), raw_f[k].detach()), k  # g = 0 either way: the same bits
        for s in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(decl.optimizer.state[raw_d[k]][s], full.optimizer.state[raw_f[k]][s]), (k, s)
        assert int(decl.optimizer.state[raw_d[k]]["step"]) == 210
    for a, b in zip(lf, ld):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (lf, ld)
    for k in ("albedo", "roughness", "metallic"):
        torch.testing.assert_close(raw_d[k].detach(), raw_f[k].detach(), rtol=0, atol=2e-3)
        assert not torch.equal(raw_d[k].detach(), ti.raw_from_scene(sc, dev)[k].detach())
    torch.testing.assert_close(light_d.base.detach(), light_f.base.detach(), rtol=0, atol=2e-3)
    full.close()
    decl.close()

    # (2) a regulariser outside the declared set
    class Bad(ti.Stage2Regularizer):
        def __call__(self, maps):
            return super().__call__(maps) + 0.1 * maps["normal_map"].square().mean()

    tr, raw, light = make(True)
    tr.stepper.regularizer = Bad(light)
    tr.stepper.close()  # nothing captured yet; the next iteration captures with the new regulariser
    before = {k: v.detach().clone() for k, v in raw.items()}
    light_before = light.base.detach().clone()
    tr.iteration(cams[0], gts[0], vds[0])
    torch.cuda.synchronize()
    # the violating step's update is guarded on the device (gigs_adam_step_guarded): nothing was changed by it
    assert all(torch.equal(raw[k].detach(), before[k]) for k in raw) and torch.equal(light.base.detach(), light_before)
    with pytest.raises(RuntimeError, match="outside the declared stage-2 set"):
        tr.iteration(cams[1], gts[1], vds[1])
        tr.close()
    tr.stepper.whole.viol_host.zero_()
    tr.close()
