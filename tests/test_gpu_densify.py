"""GPU: densification bookkeeping and the one-launch rebuild (gi-gs_amd/densify.py -> gigs_densify_stats,
gigs_gather_rows) against the step-by-step CPU restatement of scene/gaussian_model.py (oracle/densify_ref.py; parity
with the reference itself is unpinned, see that file).  Copied attributes must be bit-identical; re-sampled positions
agree to fp32 rounding (same draws fed to both sides)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES = ["xyz", "f_dc", "f_rest", "opacity", "normal", "albedo", "roughness", "metallic", "scaling", "rotation"]
SHAPES = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (8, 3), "opacity": (1,), "normal": (3,), "albedo": (3,),
          "roughness": (1,), "metallic": (1,), "scaling": (3,), "rotation": (4,)}


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _model(P, seed, extent=4.0):
    g = torch.Generator().manual_seed(seed)
    params = {n: torch.randn((P,) + SHAPES[n], generator=g) for n in NAMES}
    # scales straddling percent_dense * extent = 0.04, opacities straddling 0.05 after sigmoid
    params["scaling"] = torch.log(torch.rand(P, 3, generator=g) * 0.08 + 1e-3)
    params["opacity"] = torch.randn(P, 1, generator=g) * 2.5
    m = {n: torch.randn((P,) + SHAPES[n], generator=g) * 0.1 for n in NAMES}
    v = {n: torch.rand((P,) + SHAPES[n], generator=g) * 0.01 for n in NAMES}
    return dict(params=params, exp_avg=m, exp_avg_sq=v, stats=None), g


def _stats_rounds(P, g, rounds=6):
    out = []
    for _ in range(rounds):
        grad = torch.randn(P, 3, generator=g) * 3e-4
        radii = (torch.rand(P, generator=g) * 40 - 8).to(torch.int32)  # ~20 % invisible
        out.append((grad, radii))
    return out


def _make_optimizer(model, dev, cls):
    ps = {n: torch.nn.Parameter(model["params"][n].clone().to(dev)) for n in NAMES}
    opt = cls([{"params": [ps[n]], "lr": 1e-3, "name": n} for n in NAMES], lr=0.0, eps=1e-15)
    for n in NAMES:
        opt.state[ps[n]] = {"step": torch.tensor(5.0), "exp_avg": model["exp_avg"][n].clone().to(dev),
                            "exp_avg_sq": model["exp_avg_sq"][n].clone().to(dev)}
    return opt


def test_densify_stats_match_restatement():
    import densify
    from oracle import densify_ref as ref
    dev = _dev()
    P = 5003
    _, g = _model(8, 1)
    st = densify.DensifyState(P, dev)
    rs = dict(accum=torch.zeros(P, 1), accum_abs=torch.zeros(P, 1), accum_abs_max=torch.zeros(P, 1),
              denom=torch.zeros(P, 1), max_radii2D=torch.zeros(P))
    for grad, radii in _stats_rounds(P, g):
        ref.add_stats(rs, grad, radii)
        densify.add_densification_stats(st, grad.to(dev), radii.to(dev))
    assert torch.equal(st.denom.cpu(), rs["denom"]) and torch.equal(st.max_radii2D.cpu(), rs["max_radii2D"])
    assert torch.allclose(st.xyz_gradient_accum.cpu(), rs["accum"], rtol=1e-6, atol=1e-10)
    assert torch.allclose(st.xyz_gradient_accum_abs.cpu(), rs["accum_abs"], rtol=1e-6, atol=1e-10)
    assert torch.allclose(st.xyz_gradient_accum_abs_max.cpu(), rs["accum_abs_max"], rtol=1e-6, atol=1e-10)


@pytest.mark.parametrize("screen", [None, 20])
def test_densify_and_prune_matches_stepwise_restatement(screen):
    import densify
    import optim
    from oracle import densify_ref as ref
    dev = _dev()
    P = 4001
    model, g = _model(P, 7)
    model["stats"] = dict(accum=torch.zeros(P, 1), accum_abs=torch.zeros(P, 1), accum_abs_max=torch.zeros(P, 1),
                          denom=torch.zeros(P, 1), max_radii2D=torch.zeros(P))
    opt = _make_optimizer(model, dev, optim.FusedAdam)
    st = densify.DensifyState(P, dev)
    for grad, radii in _stats_rounds(P, g):
        ref.add_stats(model["stats"], grad, radii)
        densify.add_densification_stats(st, grad.to(dev), radii.to(dev))
    zc, zs = torch.randn(P, 3, generator=g), torch.randn(2 * P, 3, generator=g)
    ref.densify_and_prune(model, 2e-4, 0.05, 4.0, screen, zc, zs)
    noise = {"clone": zc, "split": zs}
    new, st2 = densify.densify_and_prune(opt, st, 2e-4, 0.05, 4.0, screen, _noise=lambda kind, n: noise[kind][:n])
    Pn = model["params"]["xyz"].shape[0]
    assert Pn != P and new["xyz"].shape[0] == Pn, (P, Pn, new["xyz"].shape)
    for n in NAMES:
        got, want = new[n].detach().cpu(), model["params"][n]
        if n == "xyz":
            assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)
        elif n == "scaling":
            assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)
        else:
            assert torch.equal(got, want), n
        s = opt.state[new[n]]
        assert float(s["step"]) == 5.0
        assert torch.equal(s["exp_avg"].cpu(), model["exp_avg"][n]), n
        assert torch.equal(s["exp_avg_sq"].cpu(), model["exp_avg_sq"][n]), n
        assert opt.param_groups[NAMES.index(n)]["params"][0] is new[n]
    assert st2.denom.shape == (Pn, 1) and float(st2.denom.abs().sum()) == 0.0 and st2.max_radii2D.shape == (Pn,)
    # the optimizer keeps working on the rebuilt tensors
    for n in NAMES:
        new[n].grad = torch.ones_like(new[n])
    before = new["xyz"].detach().clone()
    opt.step()
    assert not torch.equal(before, new["xyz"].detach())


def test_same_seed_gives_identical_decisions_and_samples():
    """View-parallel ranks seed the generator identically instead of broadcasting the new parameters."""
    import densify
    import optim
    dev = _dev()
    P = 3000
    outs = []
    for _ in range(2):
        model, g = _model(P, 11)
        opt = _make_optimizer(model, dev, optim.FusedAdam)
        st = densify.DensifyState(P, dev)
        for grad, radii in _stats_rounds(P, g):
            densify.add_densification_stats(st, grad.to(dev), radii.to(dev))
        gen = torch.Generator(device=dev).manual_seed(1234)
        new, _ = densify.densify_and_prune(opt, st, 2e-4, 0.05, 4.0, 20, generator=gen)
        outs.append(new)
    for n in NAMES:
        assert torch.equal(outs[0][n], outs[1][n]), n


def test_prune_points_and_reset_opacity():
    import densify
    from oracle import densify_ref as ref
    dev = _dev()
    P = 1500
    model, g = _model(P, 5)
    opt = _make_optimizer(model, dev, torch.optim.Adam)  # any optimizer with torch's state layout
    st = densify.DensifyState(P, dev)
    st.denom += 2.0
    st.max_radii2D += torch.arange(P, device=dev).float()
    mask = torch.rand(P, generator=g) > 0.7
    new, st2 = densify.prune_points(opt, st, mask.to(dev))
    keep = ~mask
    for n in NAMES:
        assert torch.equal(new[n].detach().cpu(), model["params"][n][keep])
        assert torch.equal(opt.state[new[n]]["exp_avg_sq"].cpu(), model["exp_avg_sq"][n][keep])
    assert torch.equal(st2.max_radii2D.cpu(), torch.arange(P).float()[keep]) and st2.denom.shape == (int(keep.sum()), 1)
    m2 = dict(params={"opacity": new["opacity"].detach().cpu().clone()}, exp_avg={}, exp_avg_sq={})
    ref.reset_opacity(m2)
    o = densify.reset_opacity(opt)
    assert torch.allclose(o.detach().cpu(), m2["params"]["opacity"], rtol=1e-6, atol=1e-6)
    assert float(opt.state[o]["exp_avg"].abs().sum()) == 0.0 and float(opt.state[o]["exp_avg_sq"].abs().sum()) == 0.0
    assert float(torch.sigmoid(o.detach()).max()) <= 0.01 + 1e-6


def test_checkpoint_resumes_into_fused_adam(tmp_path):
    """chkpntN.pth written from a torch.optim.Adam run (the reference's optimizer) continues under FusedAdam."""
    import os
    import types
    import optim
    import scene_io
    dev = _dev()
    model, g = _model(257, 3)
    p = {n: torch.nn.Parameter(model["params"][n].clone()) for n in NAMES}
    ref_opt = torch.optim.Adam([{"params": [p[n]], "lr": 1e-3 * (i + 1), "name": n} for i, n in enumerate(NAMES)], lr=0.0,
                               eps=1e-15)
    grads = [{n: torch.randn(p[n].shape, generator=g) for n in NAMES} for _ in range(3)]
    for k in range(2):
        for n in NAMES:
            p[n].grad = grads[k][n].clone()
        ref_opt.step()
    stats = types.SimpleNamespace(max_radii2D=torch.zeros(257), xyz_gradient_accum=torch.zeros(257, 1),
                                  xyz_gradient_accum_abs=torch.zeros(257, 1), xyz_gradient_accum_abs_max=torch.zeros(257, 1),
                                  denom=torch.zeros(257, 1))
    path = os.path.join(tmp_path, "chkpnt2.pth")
    scene_io.save_checkpoint(path, scene_io.capture(2, p, stats, ref_opt, 1.0), {}, {}, 2)
    _, params, _, opt_dict, _ = scene_io.restore(scene_io.load_checkpoint(path)["gaussians"])
    q = {n: torch.nn.Parameter(params[n].to(dev)) for n in NAMES}
    opt = optim.FusedAdam([{"params": [q[n]], "lr": 0.0, "name": n} for n in NAMES], lr=0.0, eps=1e-15)
    opt.load_state_dict(opt_dict)
    for n in NAMES:
        p[n].grad = grads[2][n].clone()
        q[n].grad = grads[2][n].clone().to(dev)
    ref_opt.step()
    opt.step()
    for n in NAMES:
        assert int(opt.state[q[n]]["step"]) == 3
        assert (q[n].detach().cpu() - p[n].detach()).abs().max().item() <= 2e-6 * p[n].detach().abs().max().item(), n
