"""GPU: the training-loop glue kernels (gi-gs_amd/losses.py, optim.py -> libgigs_hip) against the reference-pinned
fixture (tests/golden/ref_loss.npz), the CPU restatement (oracle/train_glue_ref.py) and torch.optim.Adam.

Tolerances (fp32; the HIP kernels sum the window separably and reduce per workgroup, torch sums the 121-tap window
and reduces pairwise): losses 2e-6 absolute, gradients 5e-5 of the gradient's largest magnitude."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_loss.npz"))


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _close_grad(got, want, rel=5e-5):
    want = np.asarray(want)
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got.detach().cpu().numpy() - want).max()
    assert err <= rel * scale, (err, scale)


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_l1_ssim_matches_reference_fixture(case):
    import losses
    dev = _dev()
    x = torch.from_numpy(GOLD[f"{case}_img"]).to(dev).requires_grad_(True)
    y = torch.from_numpy(GOLD[f"{case}_gt"]).to(dev)
    lam = float(GOLD["lambda"])
    loss = losses.l1_ssim_loss(x, y, lam)
    loss.backward()
    assert abs(loss.item() - float(GOLD[f"{case}_loss"])) < 2e-6
    _close_grad(x.grad, GOLD[f"{case}_grad"])
    assert abs(losses.l1_loss(x, y).item() - float(GOLD[f"{case}_l1"])) < 1e-6
    x2 = x.detach().clone().requires_grad_(True)
    s = losses.ssim(x2, y)
    s.backward()
    assert abs(s.item() - float(GOLD[f"{case}_ssim"])) < 2e-6
    _close_grad(x2.grad, GOLD[f"{case}_grad_ssim"])


def test_l1_ssim_full_size_matches_restatement_and_is_reproducible():
    import losses
    from oracle import train_glue_ref as ref
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    gt = torch.rand(3, 800, 800, generator=g)
    img = (gt + 0.05 * torch.randn(3, 800, 800, generator=g)).clamp(0, 1)
    xr = img.clone().requires_grad_(True)
    want = ref.l1_ssim_ref(xr, gt, 0.2)
    want.backward()
    x = img.to(dev).requires_grad_(True)
    got = losses.l1_ssim_loss(x, gt.to(dev), 0.2)
    got.backward()
    assert abs(got.item() - want.item()) < 2e-6
    _close_grad(x.grad, xr.grad.numpy())
    x2 = img.to(dev).requires_grad_(True)
    got2 = losses.l1_ssim_loss(x2, gt.to(dev), 0.2)
    got2.backward()
    assert got2.item() == got.item() and torch.equal(x2.grad, x.grad)  # fixed-order reductions
    # property: identical images -> ssim 1, l1 0
    assert abs(losses.ssim(gt.to(dev), gt.to(dev)).item() - 1.0) < 1e-6
    assert losses.l1_loss(gt.to(dev), gt.to(dev)).item() == 0.0


@pytest.mark.parametrize("C,H,W,step,masked", [(3, 37, 53, 1, False), (5, 64, 70, 1, True), (5, 33, 41, 3, False),
                                                (1, 2, 2, 1, True), (5, 800, 800, 1, True)])
def test_tv_losses_match_restatement(C, H, W, step, masked):
    import losses
    from oracle import train_glue_ref as ref
    dev = _dev()
    g = torch.Generator().manual_seed(C * 1000 + H)
    gt = torch.rand(3, H, W, generator=g)
    pred = torch.rand(C, H, W, generator=g)
    mask = (torch.rand(1, H, W, generator=g) > 0.3) if masked else None
    pr = pred.clone().requires_grad_(True)
    want = ref.tv_ref(gt, pr, step=step, mask=mask)
    want.backward()
    p = pred.to(dev).requires_grad_(True)
    if masked:
        got = losses.get_masked_tv_loss(mask.to(dev), gt.to(dev), p)
    else:
        got = losses.get_tv_loss(gt.to(dev), p, pad=1, step=step)
    (got * 3.0).backward()  # a non-unit upstream gradient
    assert abs(got.item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))
    _close_grad(p.grad, 3.0 * pr.grad.numpy())


def test_tv_loss_pad_pools_like_the_reference():
    import losses
    from oracle import train_glue_ref as ref
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    gt, pred = torch.rand(3, 64, 48, generator=g), torch.rand(5, 64, 48, generator=g)
    want = ref.tv_ref(torch.nn.functional.avg_pool2d(gt, 8, 8), torch.nn.functional.avg_pool2d(pred, 8, 8))
    got = losses.get_tv_loss(gt.to(dev), pred.to(dev), pad=8, step=1)
    assert abs(got.item() - want.item()) < 2e-6


def test_masked_l1_matches_torch_and_handles_empty_mask():
    import losses
    from oracle import train_glue_ref as ref
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    a, b = torch.randn(3, 75, 90, generator=g), torch.randn(3, 75, 90, generator=g)
    a[:, :5] = b[:, :5]  # ties
    mask = torch.rand(75, 90, generator=g) > 0.4
    ar = a.clone().requires_grad_(True)
    want = ref.masked_l1_ref(ar, b, mask)
    want.backward()
    ad = a.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    got = losses.masked_l1_loss(ad, bd, mask.to(dev))
    got.backward()
    assert abs(got.item() - want.item()) < 1e-6
    _close_grad(ad.grad, ar.grad.numpy())
    assert torch.equal(bd.grad, -ad.grad)
    empty = losses.masked_l1_loss(a.to(dev), b.to(dev), torch.zeros(75, 90, dtype=torch.bool, device=dev))
    assert torch.isnan(empty).item()


def test_stage1_loss_is_the_sum_of_its_parts():
    import losses
    from oracle import train_glue_ref as ref
    dev = _dev()
    g = torch.Generator().manual_seed(13)
    H, W = 96, 112
    gt, img = torch.rand(3, H, W, generator=g), torch.rand(3, H, W, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0)
    nfd = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0)
    mask = torch.rand(H, W, generator=g) > 0.2
    ir, nr = img.clone().requires_grad_(True), nrm.clone().requires_grad_(True)
    want = ref.l1_ssim_ref(ir, gt, 0.2) + ref.masked_l1_ref(nr, nfd, mask) + 0.5 * ref.tv_ref(gt, nr)
    want.backward()
    i, n = img.to(dev).requires_grad_(True), nrm.to(dev).requires_grad_(True)
    loss, ll1, nl = losses.stage1_loss(i, gt.to(dev), n, nfd.to(dev), mask.to(dev), lambda_dssim=0.2, normal_tv_weight=0.5)
    loss.backward()
    assert abs(loss.item() - want.item()) < 5e-6
    assert abs(ll1.item() - ref.l1_ref(img, gt).item()) < 1e-6
    _close_grad(i.grad, ir.grad.numpy())
    _close_grad(n.grad, nr.grad.numpy())


def test_fused_adam_follows_torch_adam():
    import optim
    dev = _dev()
    torch.manual_seed(3)
    # the reference's ten groups at P = 1237 (ragged sizes: tails, a group smaller than one float4, an empty one)
    shapes = [(1237, 3), (1237, 1, 3), (1237, 8, 3), (1237, 1), (1237, 3), (1237, 3), (1237, 1), (1237, 1), (1237, 3),
              (1237, 4), (3,), (0, 3), (6, 256, 256, 3)]
    init = [torch.randn(s) for s in shapes]
    ref_p = [torch.nn.Parameter(t.clone()) for t in init]
    got_p = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    mk = lambda ps: [{"params": [p], "lr": 1e-3 * (i + 1), "name": str(i)} for i, p in enumerate(ps)]  # noqa: E731
    ref_opt = torch.optim.Adam(mk(ref_p), lr=0.0, eps=1e-15)
    got_opt = optim.FusedAdam(mk(got_p), lr=0.0, eps=1e-15)
    for it in range(1, 8):
        for grp_r, grp_g in zip(ref_opt.param_groups, got_opt.param_groups):
            grp_r["lr"] = grp_g["lr"] = grp_r["lr"] * 0.9  # a scheduler changing lr every step (train.py:253)
        for pr, pg in zip(ref_p, got_p):
            g = torch.randn(pr.shape) * (10.0 ** (it % 4 - 2))
            if it == 3 and pr.dim() == 2:
                g[::2] = 0.0  # invisible Gaussians get exact zero gradients
            pr.grad = g.clone()
            pg.grad = g.clone().to(dev)
        ref_opt.step()
        got_opt.step(zero_grad=(it % 2 == 0))
        for pr, pg in zip(ref_p, got_p):
            if pr.numel() == 0:
                continue
            scale = pr.detach().abs().max().item()
            assert (pg.detach().cpu() - pr.detach()).abs().max().item() <= 2e-6 * scale, (it, tuple(pr.shape))
            if it % 2 == 0:
                assert pg.grad.abs().max().item() == 0.0
            sr, sg = ref_opt.state[pr], got_opt.state[pg]
            assert int(sg["step"]) == it
            for k in ("exp_avg", "exp_avg_sq"):
                assert (sg[k].cpu() - sr[k]).abs().max().item() <= 2e-6 * sr[k].abs().max().item()


def test_fused_adam_state_survives_densification_style_edits():
    """scene/gaussian_model.py:628-706 replaces a group's parameter and cats its state tensors in place."""
    import optim
    dev = _dev()
    torch.manual_seed(4)
    p = torch.nn.Parameter(torch.randn(100, 3, device=dev))
    ref_p = torch.nn.Parameter(p.detach().cpu().clone())
    opt = optim.FusedAdam([{"params": [p], "lr": 1e-2, "name": "xyz"}], lr=0.0, eps=1e-15)
    ref = torch.optim.Adam([{"params": [ref_p], "lr": 1e-2, "name": "xyz"}], lr=0.0, eps=1e-15)
    for o, q in ((opt, p), (ref, ref_p)):
        q.grad = torch.ones_like(q)
        o.step()
        grp = o.param_groups[0]
        st = o.state.get(grp["params"][0])
        ext = torch.zeros(7, 3, device=q.device)
        st["exp_avg"] = torch.cat((st["exp_avg"], ext), dim=0)
        st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], ext), dim=0)
        del o.state[grp["params"][0]]
        grp["params"][0] = torch.nn.Parameter(torch.cat((grp["params"][0].detach(), ext + 0.5), dim=0))
        o.state[grp["params"][0]] = st
        grp["params"][0].grad = torch.full_like(grp["params"][0], -2.0)
        o.step()
    a, b = opt.param_groups[0]["params"][0].detach().cpu(), ref.param_groups[0]["params"][0].detach()
    assert a.shape == (107, 3) and (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()


def test_cube_texture_matches_restated_lookup():
    from oracle import torch_pbr_ref as pr
    from pbr.texture import cube_texture
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    base = torch.rand(6, 16, 16, 3, generator=g)
    d = torch.randn(4000, 3, generator=g)
    d[:64] = torch.tensor([1.0, 1.0, 1.0]) * torch.sign(torch.randn(64, 3, generator=g))  # cube corners
    d[64:128, 2] = d[64:128, 0].abs()  # face edges
    br = base.to(pr.DT).requires_grad_(True)
    want = pr.cube_sample(br, d.to(pr.DT))
    w = torch.randn(4000, 3, generator=g)
    (want * w.to(pr.DT)).sum().backward()
    b = base.to(dev).requires_grad_(True)
    got = cube_texture(b, d.to(dev))
    (got * w.to(dev)).sum().backward()
    assert (got.detach().cpu().double() - want.detach()).abs().max().item() < 2e-6
    assert (b.grad.cpu().double() - br.grad).abs().max().item() < 1e-4 * br.grad.abs().max().item()
    planes = cube_texture(base.to(dev), d.to(dev).reshape(40, 100, 3), planar=True)
    assert planes.shape == (3, 40, 100) and torch.equal(planes.permute(1, 2, 0).reshape(4000, 3), got.detach())


@pytest.mark.parametrize("size", [(64, 128), (128, 256)])
def test_env_tv_loss_matches_restatement(size, monkeypatch):
    """(128, 256): 32 768 directions -- the lookup's backward runs as the cached gather (pbr/texture.py), whose pole texels
    hold hundreds of entries (the one-wave-per-texel part); compared with the restatement and with the atomic scatter."""
    import losses
    from oracle import train_glue_ref as ref
    dev = _dev()
    g = torch.Generator().manual_seed(22)
    base = (0.5 + 0.25 * torch.randn(6, 32, 32, 3, generator=g)).abs()
    dirs = ref.envmap_dirs_ref(size)
    got_dirs = losses.get_envmap_dirs(size, device=dev)
    assert (got_dirs.cpu() - dirs).abs().max().item() < 1e-6
    br = base.double().requires_grad_(True)
    want = ref.env_tv_ref(br, dirs)
    want.backward()
    b = base.to(dev).requires_grad_(True)
    got = losses.env_tv_loss(b, dirs.to(dev))
    got.backward()
    assert abs(got.item() - want.item()) <= 2e-5 * abs(want.item())
    assert (b.grad.cpu().double() - br.grad).abs().max().item() <= 1e-4 * br.grad.abs().max().item()
    if size[0] * size[1] >= 1 << 14:
        import pbr.texture as tex
        d_dev = dirs.to(dev)
        b1 = base.to(dev).requires_grad_(True)
        losses.env_tv_loss(b1, d_dev).backward()
        plan = tex._gather_plan(d_dev.contiguous().float(), 32, size[0] * size[1], build=False)
        assert plan is not None and plan["n_heavy"] > 0  # the gather ran, poles included
        b2 = base.to(dev).requires_grad_(True)
        losses.env_tv_loss(b2, d_dev).backward()
        assert torch.equal(b1.grad, b2.grad)  # reproducible
        monkeypatch.setenv("GIGS_CUBE_BWD_GATHER", "0")
        b3 = base.to(dev).requires_grad_(True)
        losses.env_tv_loss(b3, d_dev).backward()
        assert (b3.grad - b1.grad).abs().max().item() <= 1e-5 * b1.grad.abs().max().item()


@pytest.mark.parametrize("P,K,misalign", [(1003, 9, False), (1003, 9, True), (130, 16, False), (64, 4, False), (77, 1, False)])
def test_fused_activations_match_the_getters(P, K, misalign):
    """scene/gaussian_model.py:178-263 written with torch ops vs gigs_activate_fwd / _bwd (SH concatenation through 64-row
    LDS tiles: full and ragged tiles, every SH degree, and pointers off the 16-byte grid = views into a slab)."""
    import activations
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator().manual_seed(31)
    shapes = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (K - 1, 3), "opacity": (1,), "normal": (3,), "albedo": (3,),
              "roughness": (1,), "metallic": (1,), "scaling": (3,), "rotation": (4,)}
    base = {n: torch.randn((P,) + s, generator=g) * 2.0 for n, s in shapes.items()}
    base["normal"][:3] = 0.0  # F.normalize's eps branch
    want_in = {n: t.clone().requires_grad_(True) for n, t in base.items()}
    want = dict(shs=torch.cat((want_in["f_dc"], want_in["f_rest"]), dim=1), opacities=torch.sigmoid(want_in["opacity"]),
                normal=F.normalize(want_in["normal"], p=2, dim=-1), albedo=torch.sigmoid(want_in["albedo"]),
                roughness=torch.sigmoid(want_in["roughness"]), metallic=torch.sigmoid(want_in["metallic"]),
                scales=torch.exp(want_in["scaling"]), rotations=F.normalize(want_in["rotation"]))
    def place(t):
        if not misalign:
            return t.clone().to(dev).requires_grad_(True)
        buf = torch.zeros(t.numel() + 3, device=dev)  # the parameter starts 4 bytes into an allocation
        buf[1:1 + t.numel()] = t.reshape(-1).to(dev)
        return buf[1:1 + t.numel()].view(t.shape).detach().requires_grad_(True)
    got_in = {n: place(t) for n, t in base.items()}
    got = activations.activate(got_in)
    assert got["means3D"] is got_in["xyz"]
    ws = {n: torch.randn(want[n].shape, generator=g) for n in want}
    sum((want[n] * ws[n]).sum() for n in want if n != "metallic").backward()   # metallic: no incoming gradient
    sum((got[n] * ws[n].to(dev)).sum() for n in want if n != "metallic").backward()
    for n in want:
        assert torch.allclose(got[n].detach().cpu(), want[n].detach(), rtol=2e-6, atol=1e-7), n
    for n in shapes:
        if n == "xyz" or base[n].numel() == 0:  # K = 1: f_rest is empty
            continue
        w = want_in[n].grad if want_in[n].grad is not None else torch.zeros_like(base[n])
        gg = got_in[n].grad.cpu()
        if n == "normal":  # rows 0-2 sit on the eps branch: gradient g / 1e-12
            assert torch.allclose(gg[:3], w[:3], rtol=1e-5)
            gg, w = gg[3:], w[3:]
        assert (gg - w).abs().max().item() <= 1e-5 * max(w.abs().max().item(), 1e-30), n


def test_captured_adam_in_a_hipgraph_equals_the_eager_launch():
    """optim.CapturedAdam (gigs_adam_step_dyn: per-step scalars read from a device table) replayed from a hipGraph gives the
    parameters, moments and step counts of FusedAdam.step() bit for bit, with a learning-rate schedule, two optimizers and
    a parameter that has no gradient buffer."""
    import optim
    dev = _dev()
    torch.manual_seed(5)
    shapes = [(1237, 3), (1237, 8, 3), (1237, 1), (3,), (6, 64, 64, 3), (77, 4)]
    init = [torch.randn(s) for s in shapes]

    def make():
        ps = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
        a = optim.FusedAdam([{"params": [p], "lr": 1e-3 * (i + 1), "name": str(i)} for i, p in enumerate(ps[:4])], lr=0.0, eps=1e-15)
        b = optim.FusedAdam([{"params": ps[4:], "lr": 0.05}], lr=0.05)
        return ps, a, b

    ref_p, ref_a, ref_b = make()
    got_p, got_a, got_b = make()
    bufs = [torch.zeros_like(p) for p in got_p]
    bufs[5] = None  # no gradient buffer: untouched, like a parameter whose .grad is None
    cap = optim.CapturedAdam([got_a, got_b], got_p, bufs)
    cap.warmup()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cap.launch()
    for it in range(1, 7):
        for ga, gb in zip(ref_a.param_groups, got_a.param_groups):
            ga["lr"] = gb["lr"] = ga["lr"] * 0.8
        for i, (pr, pg) in enumerate(zip(ref_p, got_p)):
            g = (torch.randn(pr.shape) * (10.0 ** (it % 3 - 1))).to(dev)
            pr.grad = None if i == 5 else g.clone()
            if bufs[i] is not None:
                bufs[i].copy_(g)
        ref_a.step()
        ref_b.step()
        cap.advance()
        graph.replay()
        torch.cuda.synchronize()
        for i, (pr, pg) in enumerate(zip(ref_p, got_p)):
            assert torch.equal(pr.detach(), pg.detach()), (it, i)
        for ro, go in ((ref_a, got_a), (ref_b, got_b)):
            for gr_, gg_ in zip(ro.param_groups, go.param_groups):
                for pr, pg in zip(gr_["params"], gg_["params"]):
                    sr, sg = ro.state.get(pr, {}), go.state.get(pg, {})
                    if not sr:
                        assert not sg or int(sg["step"]) == 0
                        continue
                    assert int(sr["step"]) == int(sg["step"]) == it
                    assert torch.equal(sr["exp_avg"], sg["exp_avg"]) and torch.equal(sr["exp_avg_sq"], sg["exp_avg_sq"])
    cap.retreat()
    assert int(got_a.state[got_p[0]]["step"]) == 5


def test_nonzero_mask_equals_the_torch_expression():
    """losses.nonzero_mask == (x != 0).all(0, keepdim=True).float() (gaussian_renderer/__init__.py:158), incl. -0.0, NaN, Inf
    and a ragged size."""
    import losses
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 37, 53, generator=g)
    x[torch.rand(3, 37, 53, generator=g) < 0.2] = 0.0
    x[0, 0, 0], x[1, 0, 1], x[2, 0, 2], x[0, 0, 3] = -0.0, float("nan"), float("inf"), float("nan")
    x[1, 0, 3] = 0.0
    want = (x != 0).all(0, keepdim=True).float()
    got = losses.nonzero_mask(x.to(dev)).cpu()
    assert got.shape == want.shape and torch.equal(got, want)
    assert 0.0 < want.mean().item() < 1.0
