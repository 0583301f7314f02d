"""GPU parity tests of the deferred shade and the cubemap light: HIP (through the C ABI via the
drop-in `pbr` package) against the C oracle (forward) and against autograd of the float64 torch
restatement (backward).  Tolerances: forward planes mean L1 <= 1e-4 (north_star); gradients
relative to their peak."""
import os

import numpy as np
import pytest
import torch

import scenes
from helpers import GAUSS_KEYS, focal, oracle_forward
from oracle import torch_pbr_ref as tp
from test_pbr_cpu import light_levels, make_gbuffer, make_light

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
L1_TOL = 1e-4


def tt(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.requires_grad_(True) if grad else t


def lut_np():
    path = os.path.join(os.path.dirname(__file__), "..", "gi-gs_amd", "pbr", "brdf_256_256.bin")
    return np.fromfile(path, dtype=np.float32).reshape(256, 256, 2)


def rel_peak(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-20)


@pytest.mark.parametrize("res,level_rough", [(16, 1.0), (32, 0.5), (64, 0.36), (128, 0.22), (256, 0.08)])
def test_cubemap_filters_match_oracle(orc, res, level_rough):
    """Each level of the 5-level chain bench.py runs (base_res = 256: 256^2 at GGX roughness 0.08 ... 16^2 at 1.0):
    window bounds bit-exact, forward and backward against the oracle."""
    from pbr.renderutils import ops
    rng = np.random.default_rng(res)
    cm = rng.uniform(0, 1, size=(6, res, res, 3)).astype(np.float32)
    cc = ops._ndf_cutoff(level_rough, 0.99)
    cos_cut, bounds = ops._ndf_bounds(res, level_rough, 0.99, torch.device(DEV))
    assert cos_cut == cc
    b_ref = orc.specular_bounds(res, cc)
    np.testing.assert_array_equal(bounds.cpu().numpy(), b_ref)  # integer AABBs: bit-exact
    x = tt(cm, grad=True)
    out = ops._specular_cubemap.apply(x, level_rough, cc, bounds)
    ref = orc.specular_cubemap_fwd(cm, b_ref, level_rough, cc)
    # per-term weights are bit-identical to the oracle's (same IEEE sequence incl. the fp64 NDF division);
    # only the summation order differs
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=2e-5, atol=2e-6)
    g4 = rng.normal(size=(6, res, res, 4)).astype(np.float32)
    (out * tt(g4)).sum().backward()
    # HIP backward is a gather over the symmetric window; the oracle scatters like the reference
    want = orc.specular_cubemap_bwd(b_ref, g4, level_rough, cc)
    assert rel_peak(x.grad.cpu().numpy(), want) < 2e-5
    # public API: rgb / w
    y = ops.specular_cubemap(tt(cm), level_rough).cpu().numpy()
    np.testing.assert_allclose(y, ref[..., :3] / ref[..., 3:], rtol=2e-5, atol=2e-6)
    if res == 16:
        x = tt(cm, grad=True)
        d = ops.diffuse_cubemap(x)
        np.testing.assert_allclose(d.detach().cpu().numpy(), orc.diffuse_cubemap_fwd(cm), rtol=2e-5, atol=2e-6)
        g = rng.normal(size=(6, res, res, 3)).astype(np.float32)
        (d * tt(g)).sum().backward()
        assert rel_peak(x.grad.cpu().numpy(), orc.diffuse_cubemap_bwd(g)) < 2e-5


def test_cubemap_mip_matches_oracle(orc):
    from pbr.light import cubemap_mip
    rng = np.random.default_rng(5)
    cm = rng.uniform(0, 1, size=(6, 64, 64, 3)).astype(np.float32)
    x = tt(cm, grad=True)
    m = cubemap_mip.apply(x)
    np.testing.assert_allclose(m.detach().cpu().numpy(), orc.cubemap_mip_fwd(cm), atol=1e-7)
    g = rng.normal(size=(6, 32, 32, 3)).astype(np.float32)
    (m * tt(g)).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), orc.cubemap_mip_bwd(g), atol=1e-6)


@pytest.mark.parametrize("tone,gamma,use_metal,base", [(False, False, True, 64), (True, True, True, 64), (False, True, False, 64),
                                                        (False, False, True, 256)])
def test_shade_forward_and_backward(orc, tone, gamma, use_metal, base):
    """base = 256 is the light bench.py times: five specular levels, the finest ones accumulate their gradient with
    global float atomics, the coarse ones in LDS."""
    import pbr
    rng = np.random.default_rng(7)
    H, W = 48, 80
    g = make_gbuffer(rng, H, W)
    diffuse, spec = light_levels(orc, make_light(rng, base))
    lut = lut_np()
    bg = rng.uniform(size=(H, W, 3)).astype(np.float32)
    ref = orc.shade_fwd(g["normals"], g["view_dirs"], g["albedo"], g["roughness"], g["mask"], g["occlusion"],
                        g["metallic"] if use_metal else None, bg, diffuse, spec, lut, tone=tone, gamma=gamma)

    class L:  # minimal stand-in for CubemapLight after build_mips()
        pass
    light = L()
    light.diffuse = tt(diffuse, grad=True)
    light.specular = [tt(s, grad=True) for s in spec]
    alb, rgh = tt(g["albedo"], grad=True), tt(g["roughness"], grad=True)
    met = tt(g["metallic"], grad=True) if use_metal else None
    res = pbr.pbr_shading(light, tt(g["normals"]), tt(g["view_dirs"]), alb, rgh, tt(g["mask"]), tone=tone, gamma=gamma,
                          occlusion=tt(g["occlusion"]), metallic=met, brdf_lut=tt(lut)[None], background=tt(bg))
    for k in ("render_rgb", "diffuse_rgb", "specular_rgb", "diffuse_light"):
        d = np.abs(res[k].detach().cpu().numpy() - ref[k])
        assert d.mean() <= L1_TOL and d.max() <= 5e-4, (k, d.mean(), d.max())
    # backward against autograd of the float64 restatement
    gr = {k: rng.normal(size=(H, W, 3)).astype(np.float32) for k in ("render_rgb", "diffuse_rgb", "specular_rgb", "diffuse_light")}
    sum(((res[k] * tt(gr[k])).sum() for k in gr)).backward()
    t64 = dict(albedo=tp.to64(g["albedo"]).requires_grad_(True), roughness=tp.to64(g["roughness"]).requires_grad_(True),
               diffuse=tp.to64(diffuse).requires_grad_(True), spec=[tp.to64(s).requires_grad_(True) for s in spec])
    met64 = tp.to64(g["metallic"]).requires_grad_(True) if use_metal else None
    outs = tp.shade(tp.to64(g["normals"]), tp.to64(g["view_dirs"]), t64["albedo"], t64["roughness"], torch.from_numpy(g["mask"]),
                    tp.to64(g["occlusion"]), met64, tp.to64(bg), t64["diffuse"], t64["spec"], tp.to64(lut), tone=tone, gamma=gamma)
    sum(((o * tp.to64(gr[k])).sum() for o, k in zip(outs, ("render_rgb", "diffuse_rgb", "specular_rgb", "diffuse_light")))).backward()
    checks = [("albedo", alb.grad, t64["albedo"].grad), ("roughness", rgh.grad, t64["roughness"].grad),
              ("diffuse", light.diffuse.grad, t64["diffuse"].grad)]
    checks += [(f"spec{i}", a.grad, b.grad) for i, (a, b) in enumerate(zip(light.specular, t64["spec"]))]
    if use_metal:
        checks.append(("metallic", met.grad, met64.grad))
    for name, a, b in checks:
        a, b = a.cpu().numpy().astype(np.float64), b.numpy()
        # fp32 threshold flips (clamp at 0/1, the sRGB knee) touch isolated pixels: judge by the bulk
        err = np.abs(a - b)
        scale = max(np.abs(b).max(), 1e-20)
        assert np.median(err) / scale < 1e-5 and (err / scale > 1e-3).mean() < 2e-3, (name, err.max() / scale)


def test_light_build_mips_and_stage2_step(orc):
    """CubemapLight.build_mips + the whole stage-2 step run through the HIP path; the direct
    shade of the composed pipeline is compared with the oracle composition on the same view."""
    import pbr
    import pipeline
    torch.manual_seed(0)
    light = pbr.CubemapLight(base_res=64, device=DEV)
    light.build_mips()
    assert [s.shape[1] for s in light.specular] == [64, 32, 16] and light.diffuse.shape == (6, 16, 16, 3)
    base = light.base.detach().cpu().numpy()
    diffuse, spec = light_levels(orc, base)
    np.testing.assert_allclose(light.diffuse.detach().cpu().numpy(), diffuse, rtol=5e-5, atol=5e-6)
    for a, b in zip(light.specular, spec):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b, rtol=5e-5, atol=5e-6)

    sc = scenes.surface_scene(P=12_000, sh_degree=2, seed=8, scale_mu=0.025)
    cam = scenes.orbit_camera(1, 6, 160, 128, radius=3.5)
    gi = scenes.GI_DEFAULTS
    g = {k: tt(sc[k], grad=True) for k in GAUSS_KEYS}
    H, W = 128, 160
    gt = torch.rand(3, H, W, device=DEV)
    lut = pbr.get_brdf_lut().to(DEV)
    rays = pipeline.canonical_rays(cam, DEV)
    camt = {k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    out = pipeline.stage2_step(camt, g, 2, gi, light, lut, gt, rays)
    torch.cuda.synchronize()
    assert torch.isfinite(out["loss"])
    for k in ("albedo", "roughness", "metallic"):
        assert g[k].grad is not None and torch.isfinite(g[k].grad).all() and float(g[k].grad.abs().sum()) > 0, k
    assert light.base.grad is not None and torch.isfinite(light.base.grad).all() and float(light.base.grad.abs().sum()) > 0
    # geometry receives no gradient in stage 2 (normals and GI inputs are detached)
    assert float(g["means3D"].grad.abs().sum()) == 0

    # oracle composition of the same step (forward only) -> render_direct
    import torch.nn.functional as F
    r, ref = oracle_forward(orc, sc, cam)
    fx, fy = focal(cam)
    nd, pos = orc.depth_to_normal(W, H, fx, fy, cam["viewmatrix"], orc.median3x3(ref["depth"]))
    posf = orc.median3x3(pos)
    occ = orc.ssao(W, H, fx, fy, gi["radius"], gi["bias"], gi["thick"], gi["delta"], gi["step"], gi["start"], ref["normal_view"], posf)
    nm = torch.from_numpy(ref["normal"])
    nm = torch.where(torch.norm(nm, dim=0, keepdim=True) > 0, F.normalize(nm, dim=0, p=2), nm)
    nm = torch.from_numpy(orc.median3x3(nm.numpy()))
    Rm = torch.from_numpy(cam["viewmatrix"][:3, :3])
    normals_view = -(nm.permute(1, 2, 0) @ Rm)
    vd = pipeline.view_dirs_for({k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()},
                                pipeline.canonical_rays(cam, "cpu"), "cpu").numpy()
    mask = (ref["normal"] != 0).all(0)[..., None]
    rough = ref["roughness"] * (1.0 - 0.04) + 0.04
    sh = orc.shade_fwd(normals_view.numpy(), vd, ref["albedo"].transpose(1, 2, 0), rough.transpose(1, 2, 0), mask,
                       occ.transpose(1, 2, 0), ref["metallic"].transpose(1, 2, 0), None, diffuse, spec, lut[0].cpu().numpy())
    want = np.where(mask, sh["render_rgb"], 0.0).transpose(2, 0, 1)
    d = np.abs(out["render_direct"].cpu().numpy() - want)
    assert d.mean() <= L1_TOL, d.mean()


def test_stage2_hipgraph_capture_matches_eager(orc):
    """The hipGraph-captured glue segments replay the same kernels: loss and gradients agree with the
    eager step (up to float-atomic ordering), also on a second view replayed from the same graphs."""
    import pbr
    import pipeline
    torch.manual_seed(1)
    sc = scenes.surface_scene(P=10_000, sh_degree=2, seed=9, scale_mu=0.025)
    gi = scenes.GI_DEFAULTS
    H, W = 176, 224  # 154 tiles: sparse enough per tile for the tile-bucketed binning the graphed modes need
    cams = [scenes.orbit_camera(i, 6, W, H, radius=3.5) for i in (1, 4)]
    camts = [{k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    gt = torch.rand(3, H, W, device=DEV)
    lut = pbr.get_brdf_lut().to(DEV)
    rays = pipeline.canonical_rays(cams[0], DEV)
    vds = [pipeline.view_dirs_for(c, rays, DEV) for c in camts]
    results = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(2)
        light = pbr.CubemapLight(base_res=64, device=DEV)
        g = {k: tt(sc[k], grad=True) for k in GAUSS_KEYS}
        step = pipeline.Stage2Step(light, lut, gi, 2, graphs=(mode == "graph"))
        outs = []
        for rep in range(2):  # graph mode: the second round replays both captured graphs
            for ci in (0, 1):
                for t in list(g.values()) + [light.base]:
                    t.grad = None
                o = step(camts[ci], g, gt, vds[ci])
                torch.cuda.synchronize()
                outs.append((float(o["loss"]), {k: g[k].grad.clone() for k in ("albedo", "roughness", "metallic")},
                             light.base.grad.clone(), o["render_rgb"].clone()))
        results[mode] = outs
    for (le, ge, be, re_), (lg, gg, bg_, rg) in zip(results["eager"], results["graph"]):
        assert abs(le - lg) <= 1e-6 * max(1.0, abs(le)), (le, lg)
        torch.testing.assert_close(re_, rg, rtol=0, atol=1e-6)
        for k in ge:
            assert rel_peak(gg[k].cpu().numpy(), ge[k].cpu().numpy()) < 1e-4, k
        assert rel_peak(bg_.cpu().numpy(), be.cpu().numpy()) < 1e-4
    # the two views give different results (the graphs are not replaying stale inputs)
    assert abs(results["graph"][0][0] - results["graph"][1][0]) > 1e-6


@pytest.mark.parametrize("metallic", [True, False])
def test_stage2_fused_matches_unfused(orc, metallic):
    """stage2_fused (gbuffer_post + shade_ex + SSR + loss as one autograd node, 6 kernels) against the
    op-by-op torch formulation of train.py:293-402 in pipeline.Stage2Front / stage2_loss: same loss, image
    and gradients, eagerly and replayed from a hipGraph on a second view."""
    import pbr
    import pipeline
    sc = scenes.surface_scene(P=10_000, sh_degree=2, seed=9, scale_mu=0.025)
    gi = scenes.GI_DEFAULTS
    H, W = 176, 224  # 154 tiles: sparse enough per tile for the tile-bucketed binning the graphed modes need
    cams = [scenes.orbit_camera(i, 6, W, H, radius=3.5) for i in (1, 4)]
    camts = [{k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    torch.manual_seed(1)
    gt = torch.rand(3, H, W, device=DEV) * 0.5
    lut = pbr.get_brdf_lut().to(DEV)
    rays = pipeline.canonical_rays(cams[0], DEV)
    vds = [pipeline.view_dirs_for(c, rays, DEV) for c in camts]
    results = {}
    for mode in ("unfused", "fused", "fused_graph", "fused_graph_raster", "fused_step_graph"):
        torch.manual_seed(2)
        light = pbr.CubemapLight(base_res=64, device=DEV)
        g = {k: tt(sc[k], grad=True) for k in GAUSS_KEYS}
        # fused_step_graph: the whole iteration as two hand-captured hipGraphs (pipeline.WholeStepGraph, the default);
        # fused_graph: glue replayed from hipGraphs, rasterizer launched eagerly with asynchronous binning;
        # fused_graph_raster: the rasterizer captured too, six make_graphed_callables pieces (GIGS_RASTER_GRAPH=1)
        os.environ["GIGS_RASTER_GRAPH"] = "1" if mode == "fused_graph_raster" else "0"
        os.environ["GIGS_STEP_GRAPH"] = "1" if mode == "fused_step_graph" else "0"
        step = pipeline.Stage2Step(light, lut, gi, 2, metallic=metallic, fused=mode != "unfused",
                                   graphs=mode in ("fused_graph", "fused_graph_raster", "fused_step_graph"))
        outs = []
        for ci in (0, 1, 0):
            for t in list(g.values()) + [light.base]:
                t.grad = None
            o = step(camts[ci], g, gt, vds[ci])
            torch.cuda.synchronize()
            outs.append((float(o["loss"]), {k: g[k].grad.clone() for k in ("albedo", "roughness", "metallic")},
                         light.base.grad.clone(), o["render_rgb"].clone(), o["IRR"].clone()))
        if mode == "fused_step_graph":
            assert step.whole is not None and step.whole.recaptures == 1  # three views, one capture
        results[mode] = outs
    os.environ.pop("GIGS_RASTER_GRAPH", None)
    os.environ.pop("GIGS_STEP_GRAPH", None)
    for mode in ("fused", "fused_graph", "fused_graph_raster", "fused_step_graph"):
        for (lu, gu, bu, ru, iu), (lf, gf, bf, rf, irf) in zip(results["unfused"], results[mode]):
            assert abs(lu - lf) <= 2e-6 * max(1.0, abs(lu)), (mode, lu, lf)
            torch.testing.assert_close(irf, iu, rtol=0, atol=2e-6)
            # a median tie / sRGB knee can move single pixels: compare the image in the mean and at the 99.9th percentile
            d = (rf - ru).abs().flatten()
            assert float(d.mean()) <= 1e-6 and float(torch.quantile(d, 0.999)) <= 1e-5, (mode, float(d.mean()), float(d.max()))
            for k in gu:
                assert rel_peak(gf[k].cpu().numpy(), gu[k].cpu().numpy()) < 2e-3, (mode, k)
            assert rel_peak(bf.cpu().numpy(), bu.cpu().numpy()) < 2e-3, mode
    assert abs(results["fused_graph"][0][0] - results["fused_graph"][1][0]) > 1e-6


@pytest.mark.parametrize("raster", ["step_graph", "graph", "eager_async"])
def test_graphed_step_survives_a_binning_overflow(raster, monkeypatch):
    """The whole-step hipGraph bins into a fixed-capacity buffer; a view with more instances than the capacity raises
    the device-side overflow flag, the capacity grows, the graph is re-captured and the step repeated: same loss,
    image and gradients as the eager step."""
    import pbr
    import pipeline
    from diff_gaussian_rasterization import AsyncBinning
    monkeypatch.setenv("GIGS_RASTER_GRAPH", "1" if raster == "graph" else "0")
    monkeypatch.setenv("GIGS_STEP_GRAPH", "1" if raster == "step_graph" else "0")
    sc = scenes.surface_scene(P=20_000, sh_degree=2, seed=4, scale_mu=0.03)
    gi = scenes.GI_DEFAULTS
    H, W = 160, 208
    cam = scenes.orbit_camera(2, 6, W, H, radius=3.5)
    camt = {k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    torch.manual_seed(3)
    gt = torch.rand(3, H, W, device=DEV) * 0.5
    lut = pbr.get_brdf_lut().to(DEV)
    vd = pipeline.view_dirs_for(camt, pipeline.canonical_rays(cam, DEV), DEV)
    res = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(5)
        light = pbr.CubemapLight(base_res=64, device=DEV)
        g = {k: tt(sc[k], grad=True) for k in GAUSS_KEYS}
        step = pipeline.Stage2Step(light, lut, gi, 2, fused=True, graphs=(mode == "graph"))
        if mode == "graph" and raster == "step_graph":
            step.whole = pipeline.WholeStepGraph(step, camt, g)
            step.whole.capacity = 65536  # far below this view's instance count
        elif mode == "graph" and raster == "graph":
            step.graster = pipeline.GraphedRaster(camt, g, gi, 2, capacity=65536)
        elif mode == "graph":
            step._abin = AsyncBinning(65536, DEV)  # the eager rasterizer's asynchronous binning, same protocol
        o = step(camt, g, gt, vd)
        torch.cuda.synchronize()
        if mode == "graph" and raster == "step_graph":
            assert step.whole.recaptures == 2 and step.whole.capacity > 65536
        elif mode == "graph" and raster == "graph":
            assert step.graster.recaptures == 2 and step.graster.capacity > 65536
        elif mode == "graph":
            assert step._abin.capacity > 65536
        if mode == "graph":
            assert o["num_rendered"] > 65536
        res[mode] = (float(o["loss"]), o["render_rgb"].clone(), {k: g[k].grad.clone() for k in ("albedo", "roughness", "metallic")},
                     light.base.grad.clone())
    (le, re_, ge, be), (lg, rg, gg, bg_) = res["eager"], res["graph"]
    assert abs(le - lg) <= 2e-6 * max(1.0, abs(le))
    torch.testing.assert_close(rg, re_, rtol=0, atol=2e-6)
    for k in ge:
        assert rel_peak(gg[k].cpu().numpy(), ge[k].cpu().numpy()) < 2e-3, k
    assert rel_peak(bg_.cpu().numpy(), be.cpu().numpy()) < 2e-3


@pytest.mark.gpu
def test_fused_gbuffer_post_matches_torch_chain_forward_and_backward():
    """pipeline.gbuffer_post (torch ops, gaussian_renderer/__init__.py:157-199) vs the fused kernels incl. the
    gradient w.r.t. normal_map that stage 1 needs."""
    import pipeline
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(17)
    H, W = 70, 93
    nm = torch.randn(3, H, W, generator=g)
    nm[:, 10:20, 30:60] = 0.0                      # empty pixels
    nm[:, 40, 5] = torch.tensor([0.0, 1.0, 0.0])   # a zero component: masked out, still normalised
    nm = (nm * 8).round() / 8                       # coarse values: ties inside the median windows
    nfd = torch.randn(3, H, W, generator=g)
    nfd[:, 50:, :20] = 0.0
    onv = torch.randn(3, H, W, generator=g)
    onv[:, :5] = 0.0
    vm = torch.linalg.qr(torch.randn(4, 4, generator=g))[0]
    w = torch.randn(3, H, W, generator=g)
    a = nm.clone().to(dev).requires_grad_(True)
    ref = pipeline.gbuffer_post(nfd.to(dev), a, onv.to(dev), vm.to(dev))
    (ref[2] * w.to(dev)).sum().backward()
    b = nm.clone().to(dev).requires_grad_(True)
    got = pipeline.gbuffer_post_fused(nfd.to(dev), b, onv.to(dev), vm.to(dev))
    (got[2] * w.to(dev)).sum().backward()
    assert torch.equal(got[1], ref[1]) and torch.equal(got[3], ref[3])
    for i in (0, 2, 4):
        assert torch.allclose(got[i], ref[i], rtol=1e-6, atol=1e-6), i
    assert (b.grad - a.grad).abs().max().item() <= 1e-5 * a.grad.abs().max().item()


@pytest.mark.gpu
def test_build_mips_fused_chain_and_prescaled_tables_match_op_by_op(monkeypatch):
    """build_mips with the one-node mip chain and the 1/wsum-folded backward tables (defaults) against the op-by-op
    formulation (GIGS_MIP_CHAIN=0, division pass + unscaled table): same light, same gradient of a random functional."""
    import pbr
    from pbr.renderutils import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    base0 = torch.rand(6, 64, 64, 3, device=dev) + 0.1
    ws = None
    results = []
    for chain, prescaled in (("0", "0"), ("1", "1")):
        monkeypatch.setenv("GIGS_MIP_CHAIN", chain)
        monkeypatch.setenv("GIGS_SPEC_PRESCALED", prescaled)
        ops._weightTables.clear()
        light = pbr.CubemapLight(base_res=64, device=dev)
        with torch.no_grad():
            light.base.copy_(base0)
        light.build_mips()
        outs = [light.diffuse] + list(light.specular)
        if ws is None:
            ws = [torch.randn_like(o) for o in outs]
        sum((o * w).sum() for o, w in zip(outs, ws)).backward()
        results.append(([o.detach().clone() for o in outs], light.base.grad.detach().clone()))
    ops._weightTables.clear()
    (o0, g0), (o1, g1) = results
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)  # the forward runs the same kernels
    assert (g0 - g1).abs().max().item() <= 2e-6 * g0.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("base", [64, 256])
def test_merged_level_filter_equals_per_level(base, monkeypatch):
    """build_mips filters all GGX levels in one launch each way (gigs_specular_cubemap_multi_w); the per-level calls
    (GIGS_SPEC_MULTI=0) must give the same levels and the same gradient of the base cubemap, bit for bit (same kernels'
    bodies, same per-texel summation order)."""
    import pbr
    res = {}
    for multi in ("1", "0"):
        monkeypatch.setenv("GIGS_SPEC_MULTI", multi)
        torch.manual_seed(11)
        light = pbr.CubemapLight(base_res=base, device=DEV)
        light.build_mips()
        g = torch.Generator(device="cpu").manual_seed(5)
        loss = sum((s * torch.randn(s.shape, generator=g).to(DEV)).sum() for s in light.specular) + light.diffuse.sum()
        loss.backward()
        torch.cuda.synchronize()
        res[multi] = ([s.detach().clone() for s in light.specular], light.base.grad.clone())
    for a, b in zip(res["1"][0], res["0"][0]):
        assert torch.equal(a, b)
    assert torch.equal(res["1"][1], res["0"][1])


def test_whole_step_graph_gradient_semantics_and_recapture(monkeypatch):
    """pipeline.WholeStepGraph hands its gradients out with loss.backward()'s semantics and follows the parameters:
    (1) `.grad is None` before the step -> the fresh gradient; (2) an existing `.grad` (gradient accumulation over views)
    -> old + new; (3) a parameter tensor replaced by another one (densification) -> one re-capture, results as eager."""
    import pbr
    import pipeline
    monkeypatch.setenv("GIGS_STEP_GRAPH", "1")
    monkeypatch.setenv("GIGS_RASTER_GRAPH", "0")
    sc = scenes.surface_scene(P=8000, sh_degree=2, seed=21, scale_mu=0.025)
    gi = scenes.GI_DEFAULTS
    H, W = 176, 224
    cams = [scenes.orbit_camera(i, 6, W, H, radius=3.5) for i in (0, 3)]
    camts = [{k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    torch.manual_seed(4)
    gt = torch.rand(3, H, W, device=DEV) * 0.5
    lut = pbr.get_brdf_lut().to(DEV)
    rays = pipeline.canonical_rays(cams[0], DEV)
    vds = [pipeline.view_dirs_for(c, rays, DEV) for c in camts]
    keys = ("albedo", "roughness", "metallic")

    def fresh():
        torch.manual_seed(6)
        light = pbr.CubemapLight(base_res=64, device=DEV)
        g = {k: tt(sc[k], grad=True) for k in GAUSS_KEYS}
        return light, g

    def grads(g, light):
        return {**{k: g[k].grad.clone() for k in keys}, "light": light.base.grad.clone()}

    # eager reference: the two views separately
    light, g = fresh()
    eager = pipeline.Stage2Step(light, lut, gi, 2, fused=True, graphs=False)
    ref = []
    for ci in (0, 1):
        for t in list(g.values()) + [light.base]:
            t.grad = None
        o = eager(camts[ci], g, gt, vds[ci])
        torch.cuda.synchronize()
        ref.append((float(o["loss"]), grads(g, light)))

    light, g = fresh()
    step = pipeline.Stage2Step(light, lut, gi, 2, fused=True, graphs=True)
    o = step(camts[0], g, gt, vds[0])  # (1)
    torch.cuda.synchronize()
    assert step.whole is not None and step.whole.recaptures == 1
    assert abs(float(o["loss"]) - ref[0][0]) <= 2e-6 * max(1.0, abs(ref[0][0]))
    got = grads(g, light)
    for k in got:
        assert rel_peak(got[k].cpu().numpy(), ref[0][1][k].cpu().numpy()) < 2e-3, k
    # (2) keep the gradients of view 0, step on view 1: the sum of both
    keep = {k: v.clone() for k, v in got.items()}
    for k in keys:
        g[k].grad = keep[k].clone()
    light.base.grad = keep["light"].clone()
    step(camts[1], g, gt, vds[1])
    torch.cuda.synchronize()
    assert step.whole.recaptures == 1
    acc = grads(g, light)
    for k in acc:
        want = (ref[0][1][k] + ref[1][1][k]).cpu().numpy()
        assert rel_peak(acc[k].cpu().numpy(), want) < 2e-3, k
    # (2b) accumulation WITHOUT re-assignment (ordinary gradient accumulation over views): the parameters still hold the
    # buffers the previous step handed out, which the next replay overwrites -- the sum must come out all the same
    for t in list(g.values()) + [light.base]:
        t.grad = None
    step(camts[0], g, gt, vds[0])
    step(camts[1], g, gt, vds[1])
    torch.cuda.synchronize()
    acc = grads(g, light)
    for k in acc:
        want = (ref[0][1][k] + ref[1][1][k]).cpu().numpy()
        assert rel_peak(acc[k].cpu().numpy(), want) < 2e-3, ("accumulate in place", k)
    # (3) a replaced parameter tensor (same values): one re-capture, same results
    g["albedo"] = g["albedo"].detach().clone().requires_grad_(True)
    for t in list(g.values()) + [light.base]:
        t.grad = None
    o = step(camts[1], g, gt, vds[1])
    torch.cuda.synchronize()
    assert step.whole.recaptures == 2
    assert abs(float(o["loss"]) - ref[1][0]) <= 2e-6 * max(1.0, abs(ref[1][0]))
    got = grads(g, light)
    for k in got:
        assert rel_peak(got[k].cpu().numpy(), ref[1][1][k].cpu().numpy()) < 2e-3, k
    # (4) another image size: a second capture beside the first, which stays valid
    H2, W2 = 144, 176
    cam2 = scenes.orbit_camera(2, 6, W2, H2, radius=3.5)
    camt2 = {k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in cam2.items()}
    gt2 = torch.rand(3, H2, W2, device=DEV) * 0.5
    vd2 = pipeline.view_dirs_for(camt2, pipeline.canonical_rays(cam2, DEV), DEV)
    for t in list(g.values()) + [light.base]:
        t.grad = None
    o2 = step(camt2, g, gt2, vd2)
    torch.cuda.synchronize()
    assert tuple(o2["render_rgb"].shape[-2:]) == (H2, W2) and len(step._wholes) == 2
    for t in list(g.values()) + [light.base]:
        t.grad = None
    o = step(camts[1], g, gt, vds[1])
    torch.cuda.synchronize()
    assert step.whole.recaptures == 2 and abs(float(o["loss"]) - ref[1][0]) <= 2e-6 * max(1.0, abs(ref[1][0]))


def test_loss_gradient_planes_from_the_forward_pass_equal_the_backward_kernel():
    """gigs_stage2_loss_fwd_grad writes d loss / d render_direct and d loss / d IRR for a unit upstream gradient in the
    forward pass; times g_loss they must equal what the stand-alone backward kernel (gigs_stage2_loss_bwd) computes, and
    loss / render_rgb / acc4 must equal gigs_stage2_loss_fwd's."""
    import gigs_lib
    lib = gigs_lib.lib()
    H, W = 83, 131
    g = torch.Generator(device="cpu").manual_seed(12)
    rnd = lambda *s: torch.rand(*s, generator=g).to(DEV)  # noqa: E731
    direct, irr, gt = rnd(3, H, W), rnd(3, H, W) * 0.3, rnd(3, H, W)
    irr[:, 10:14, 20:30] = float("nan")  # NaN windows drop their gradient
    irr[:, 40:60, 50:90] = 0.001          # the linear segment of the sRGB curve, and median ties
    mask = (rnd(1, H, W) > 0.3).float()
    rough, metal = rnd(1, H, W), rnd(1, H, W)
    n_acc = 4 + 4 * 256
    s = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()  # noqa: E731
    rgb_a, acc_a, loss_a = torch.empty(3, H, W, device=DEV), torch.empty(n_acc, device=DEV), torch.empty(1, device=DEV)
    rgb_b, acc_b, loss_b = torch.empty(3, H, W, device=DEV), torch.empty(n_acc, device=DEV), torch.empty(1, device=DEV)
    du, iu = torch.empty(3, H, W, device=DEV), torch.empty(3, H, W, device=DEV)
    gigs_lib.check(lib.gigs_stage2_loss_fwd(H, W, p(direct), p(irr), p(gt), p(mask), p(rough), p(metal), p(rgb_a), p(acc_a),
                                            p(loss_a), s), "loss_fwd")
    gigs_lib.check(lib.gigs_stage2_loss_fwd_grad(H, W, p(direct), p(irr), p(gt), p(mask), p(rough), p(metal), p(rgb_b),
                                                 p(acc_b), p(loss_b), p(du), p(iu), s), "loss_fwd_grad")
    gl = torch.tensor([0.75], device=DEV)  # a power of two times 3: the scaling is exact up to one rounding
    dd, di = torch.empty(3, H, W, device=DEV), torch.empty(3, H, W, device=DEV)
    dr, dm = torch.empty(1, H, W, device=DEV), torch.empty(1, H, W, device=DEV)
    gigs_lib.check(lib.gigs_stage2_loss_bwd(H, W, p(direct), p(irr), p(gt), p(mask), p(acc_a), p(gl), p(dd), p(di), p(dr),
                                            p(dm), s), "loss_bwd")
    torch.cuda.synchronize()
    assert torch.equal(loss_a, loss_b) and torch.equal(acc_a[:4], acc_b[:4])
    assert torch.equal(rgb_a.nan_to_num(), rgb_b.nan_to_num())
    torch.testing.assert_close(du * 0.75, dd, rtol=2e-7, atol=0)
    # several atomics of either sign may land on one texel in any order: cancellation leaves ~1e-11 of a ~1e-4 term
    torch.testing.assert_close(iu * 0.75, di, rtol=1e-6, atol=1e-9)
    assert float(di.abs().sum()) > 0 and float((dd != 0).float().mean()) > 0.9


def test_whole_step_graph_writes_gradients_into_the_all_reduce_slab(monkeypatch):
    """Multi-GPU readiness on one GPU: with dp.GradSlab's sink active, the captured backward writes the rasterizer's
    gradients straight into the slab (the `.grad` tensors handed out ARE the slab's views, on every replay), the light's
    gradient is copied in once by the slab, and the numbers equal the eager step's."""
    import diff_gaussian_rasterization as dgr
    import dp
    import pbr
    import pipeline
    monkeypatch.setenv("GIGS_STEP_GRAPH", "1")
    monkeypatch.setenv("GIGS_RASTER_GRAPH", "0")
    sc = scenes.surface_scene(P=6000, sh_degree=2, seed=31, scale_mu=0.025)
    gi = scenes.GI_DEFAULTS
    H, W = 160, 208
    cams = [scenes.orbit_camera(i, 6, W, H, radius=3.5) for i in (0, 2)]
    camts = [{k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    torch.manual_seed(8)
    gt = torch.rand(3, H, W, device=DEV) * 0.5
    lut = pbr.get_brdf_lut().to(DEV)
    rays = pipeline.canonical_rays(cams[0], DEV)
    vds = [pipeline.view_dirs_for(c, rays, DEV) for c in camts]
    order = ["means3D", "opacities", "normal", "shs", "scales", "rotations", "albedo", "roughness", "metallic"]
    names = {"opacities": "opacity", "shs": "sh"}

    def run(graphs):
        torch.manual_seed(9)
        light = pbr.CubemapLight(base_res=64, device=DEV)
        g = {k: tt(sc[k], grad=True) for k in GAUSS_KEYS}
        params = [g[k] for k in order] + [light.base]
        slab = dp.GradSlab(params)
        sink = slab.sink([names.get(k, k) for k in order])
        step = pipeline.Stage2Step(light, lut, gi, 2, fused=True, graphs=graphs)
        outs = []
        for ci in (0, 1, 0):
            for p in params:
                p.grad = None
            with dgr.grad_sink(sink):
                step(camts[ci], g, gt, vds[ci])
            if graphs:
                for i, k in enumerate(order):
                    assert g[k].grad is not None and g[k].grad.data_ptr() == slab.views[i].data_ptr(), k
            slab._gather_stray()  # what allreduce_async does before the collective: strays (the light) are copied in
            torch.cuda.synchronize()
            assert light.base.grad.data_ptr() == slab.views[-1].data_ptr()
            outs.append(slab.flat.clone())
        return outs

    eager, graph = run(False), run(True)
    for a, b in zip(eager, graph):
        assert rel_peak(b.cpu().numpy(), a.cpu().numpy()) < 2e-3
    assert float((graph[0] - graph[1]).abs().max()) > 0  # two views, two different gradients in the same slab


def test_grad_slab_attach_with_sink_does_not_double_gradients():
    """dp.GradSlab.attach() + grad_sink on the EAGER path: a parameter whose .grad already aliases the sink tensor would
    be added to itself by AccumulateGrad (the rasterizer's backward overwrites the view and returns it).  attach(skip=sink)
    leaves sinked parameters at .grad = None; after _gather_stray the slab holds exactly the plain backward's gradients."""
    import diff_gaussian_rasterization as dgr
    import dp
    import pbr
    import pipeline
    sc = scenes.surface_scene(P=5000, sh_degree=2, seed=33, scale_mu=0.025)
    gi = scenes.GI_DEFAULTS
    H, W = 128, 160
    cam = scenes.orbit_camera(1, 6, W, H, radius=3.5)
    camt = {k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    torch.manual_seed(3)
    gt = torch.rand(3, H, W, device=DEV) * 0.5
    lut = pbr.get_brdf_lut().to(DEV)
    vd = pipeline.view_dirs_for(camt, pipeline.canonical_rays(cam, DEV), DEV)
    order = ["means3D", "opacities", "normal", "shs", "scales", "rotations", "albedo", "roughness", "metallic"]
    names = {"opacities": "opacity", "shs": "sh"}

    def run(mode):
        torch.manual_seed(9)
        light = pbr.CubemapLight(base_res=64, device=DEV)
        g = {k: tt(sc[k], grad=True) for k in GAUSS_KEYS}
        params = [g[k] for k in order] + [light.base]
        step = pipeline.Stage2Step(light, lut, gi, 2, fused=True, graphs=False)
        if mode == "plain":
            step(camt, g, gt, vd)
            torch.cuda.synchronize()
            return torch.cat([p.grad.reshape(-1) for p in params])
        slab = dp.GradSlab(params)
        sink = slab.sink([names.get(k, k) for k in order])
        slab.attach(skip=sink)
        assert all(g[k].grad is None for k in order) and light.base.grad is not None  # the light accumulates in place
        with dgr.grad_sink(sink):
            step(camt, g, gt, vd)
        slab._gather_stray()
        torch.cuda.synchronize()
        return slab.flat.clone()

    plain, slab = run("plain"), run("slab")
    assert rel_peak(slab.cpu().numpy(), plain.cpu().numpy()) < 2e-3
    assert float(plain.abs().max()) > 0


def test_shade_backward_in_two_launches_equals_one(monkeypatch):
    """GIGS_SHADE_BWD_SPLIT=1: the fused node's backward issues the material gradients on the main stream and the
    light-texture scatter on the light's stream (gigs_shade_ext.part = 1 / 2): same gradients as the single launch, eager and
    from the whole-step graphs."""
    import pbr
    import pipeline
    sc = scenes.surface_scene(P=6000, sh_degree=2, seed=41, scale_mu=0.025)
    gi = scenes.GI_DEFAULTS
    H, W = 144, 176
    cam = scenes.orbit_camera(1, 6, W, H, radius=3.5)
    camt = {k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    torch.manual_seed(2)
    gt = torch.rand(3, H, W, device=DEV) * 0.5
    lut = pbr.get_brdf_lut().to(DEV)
    vd = pipeline.view_dirs_for(camt, pipeline.canonical_rays(cam, DEV), DEV)

    def run(split, graphs):
        monkeypatch.setenv("GIGS_SHADE_BWD_SPLIT", split)
        torch.manual_seed(9)
        light = pbr.CubemapLight(base_res=64, device=DEV)
        g = {k: tt(sc[k], grad=True) for k in GAUSS_KEYS}
        step = pipeline.Stage2Step(light, lut, gi, 2, fused=True, graphs=graphs)
        for _ in range(2):
            for t in list(g.values()) + [light.base]:
                t.grad = None
            o = step(camt, g, gt, vd)
        torch.cuda.synchronize()
        return float(o["loss"]), {k: g[k].grad.clone() for k in ("albedo", "roughness", "metallic")}, light.base.grad.clone()

    ref = run("0", False)
    for graphs in (False, True):
        got = run("1", graphs)
        assert abs(got[0] - ref[0]) <= 2e-6 * max(1.0, abs(ref[0]))
        for k in ref[1]:
            assert torch.equal(got[1][k], ref[1][k]) or rel_peak(got[1][k].cpu().numpy(), ref[1][k].cpu().numpy()) < 1e-6, k
        assert rel_peak(got[2].cpu().numpy(), ref[2].cpu().numpy()) < 2e-3  # float atomics: order-dependent rounding


def test_hipgraph_lifetime_is_deterministic(monkeypatch):
    """The round-3 host segfault (hip::Graph::UpdateStreams at the first replay of a fresh exec) came from graph execs that
    a cyclic-GC pass destroyed at an arbitrary moment: WholeStepGraph <-> Stage2Step was a reference cycle.  Now the owner
    is held weakly and teardown is explicit: (a) dropping the last reference to a stepper destroys its graphs at once, with
    the collector switched OFF (no cycle); (b) close() leaves nothing captured and the stepper captures again on the next
    call; (c) the bisected sequence -- earlier captures, an eager full-size step, a new capture, its first replay -- runs
    with the collector enabled.  Run once; the autouse teardown of the GPU tests is not what keeps it alive."""
    import gc
    import weakref

    import pbr
    import pipeline
    import train_iteration
    monkeypatch.setenv("GIGS_STEP_GRAPH", "1")
    monkeypatch.setenv("GIGS_RASTER_GRAPH", "0")
    sc = scenes.surface_scene(P=6000, sh_degree=2, seed=21, scale_mu=0.03)
    gi = scenes.GI_DEFAULTS
    H, W = 112, 144
    cam = scenes.orbit_camera(0, 6, W, H, radius=3.5)
    camt = {k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    gt = torch.rand(3, H, W, device=DEV) * 0.5
    lut = pbr.get_brdf_lut().to(DEV)
    vd = pipeline.view_dirs_for(camt, pipeline.canonical_rays(cam, DEV), DEV)

    def make():
        torch.manual_seed(6)
        light = pbr.CubemapLight(base_res=64, device=DEV)
        g = {k: tt(sc[k], grad=True) for k in GAUSS_KEYS}
        return pipeline.Stage2Step(light, lut, gi, 2, fused=True, graphs=True), g

    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        # (a) no cycle: the graphs die with the last reference, collector off
        step, g = make()
        loss0 = float(step(camt, g, gt, vd)["loss"])
        whole = weakref.ref(step.whole)
        gf = weakref.ref(step.whole.gf)
        assert whole() is not None and gf() is not None
        del step
        assert whole() is None and gf() is None, "a reference cycle keeps the graph execs alive until the collector runs"
        # (b) close(): nothing captured afterwards, and the next call captures again with the same result
        step, g = make()
        step(camt, g, gt, vd)
        w = step.whole
        assert w.gf is not None
        step.close()
        assert w.gf is None and w.gb is None and w.res is None and w.inner is None
        assert abs(float(step(camt, g, gt, vd)["loss"]) - loss0) <= 1e-6 * max(1.0, abs(loss0))
        assert step.whole is w and w.gf is not None and w.recaptures == 2
        # a trainer (trainer <-> stepper IS a cycle through bound methods): close() is what releases its graphs
        raw = train_iteration.raw_from_scene(sc, DEV)
        with train_iteration.Stage2Trainer(raw, pbr.CubemapLight(base_res=64, device=DEV), lut, gi, 2, graphs=True) as tr:
            tr.iteration(camt, gt, vd)
            tw = tr.stepper.whole
            assert tw is not None and tw.gf is not None
        assert tw.gf is None and tw.gb is None and tw.go is None
    finally:
        if was:
            gc.enable()
    # (c) the bisected sequence, collector enabled: earlier captures (above, still alive in `step`), an eager full-size
    # step, a new capture, its first replay
    Hf, Wf = 800, 800
    camf = scenes.orbit_camera(5, 64, Wf, Hf, radius=3.5)
    camft = {k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in camf.items()}
    big = scenes.surface_scene(P=120_000, sh_degree=2, seed=0)
    gb = {k: tt(big[k], grad=True) for k in GAUSS_KEYS}
    gtf = torch.rand(3, Hf, Wf, device=DEV) * 0.5
    vdf = pipeline.view_dirs_for(camft, pipeline.canonical_rays(camf, DEV), DEV)
    lightf = pbr.CubemapLight(base_res=64, device=DEV)
    pipeline.Stage2Step(lightf, lut, gi, 2, fused=True, graphs=False)(camft, gb, gtf, vdf)  # eager, full size
    new = pipeline.Stage2Step(lightf, lut, gi, 2, fused=True, graphs=True)
    for t in list(gb.values()) + [lightf.base]:
        t.grad = None
    o1 = new(camft, gb, gtf, vdf)   # capture + first replay
    l1 = float(o1["loss"])
    for t in list(gb.values()) + [lightf.base]:
        t.grad = None
    o2 = new(camft, gb, gtf, vdf)   # second replay
    torch.cuda.synchronize()
    assert np.isfinite(l1) and abs(float(o2["loss"]) - l1) <= 1e-6 * max(1.0, abs(l1))
    new.close()
    step.close()


def test_drop_in_light_prefetch_is_adopted_and_changes_nothing(monkeypatch):
    """The op-by-op caller of train.py:330-402 (render -> cubemap.build_mips() -> pbr_shading -> Gaussian_SSR -> loss):
    from the second iteration on the rasterizer's forward starts the light's pre-filter on the light's side stream and
    build_mips() adopts it (pbr/light.py::CubemapLight.prefetch) -- same loss, image and gradients as with
    GIGS_LIGHT_PREFETCH=0, over optimizer steps that change the light between iterations; a caller that changes the light
    between the rasterizer and build_mips() gets a correct, freshly built filter and no further prefetches."""
    import pbr
    import pipeline
    sc = scenes.surface_scene(P=10_000, sh_degree=2, seed=9, scale_mu=0.025)
    gi = scenes.GI_DEFAULTS
    H, W = 176, 224
    cams = [scenes.orbit_camera(i, 6, W, H, radius=3.5) for i in (1, 4)]
    camts = [{k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    torch.manual_seed(1)
    gt = torch.rand(3, H, W, device=DEV) * 0.5
    lut = pbr.get_brdf_lut().to(DEV)
    rays = pipeline.canonical_rays(cams[0], DEV)
    vds = [pipeline.view_dirs_for(c, rays, DEV) for c in camts]

    def run(prefetch):
        monkeypatch.setenv("GIGS_LIGHT_PREFETCH", "1" if prefetch else "0")
        torch.manual_seed(2)
        light = pbr.CubemapLight(base_res=64, device=DEV)
        g = {k: tt(sc[k], grad=True) for k in GAUSS_KEYS}
        opt = torch.optim.SGD([light.base, g["albedo"]], lr=0.05)
        step = pipeline.Stage2Step(light, lut, gi, 2, metallic=True, fused=False, graphs=False)
        outs = []
        for it in range(5):
            opt.zero_grad(set_to_none=True)
            o = step(camts[it % 2], g, gt, vds[it % 2])
            torch.cuda.synchronize()
            outs.append((float(o["loss"]), light.base.grad.clone(), g["albedo"].grad.clone(), o["render_rgb"].clone()))
            opt.step()
            with torch.no_grad():
                light.clamp_(min=0.0)
        return outs, light

    ref, _ = run(False)
    got, light = run(True)
    assert light.prefetch_stats == dict(started=4, adopted=4, discarded=0), light.prefetch_stats
    for (lr_, br, ar, rr), (lg, bg, ag, rg) in zip(ref, got):
        assert abs(lr_ - lg) <= 2e-6 * max(1.0, abs(lr_))
        torch.testing.assert_close(rg, rr, rtol=0, atol=2e-6)
        assert rel_peak(bg.cpu().numpy(), br.cpu().numpy()) < 2e-3  # float-atomic sums: rounding differs from run to run
        assert rel_peak(ag.cpu().numpy(), ar.cpu().numpy()) < 2e-3
    assert abs(ref[0][0] - ref[2][0]) > 1e-7  # the optimizer did change the light / albedo between iterations

    # a caller that touches the light between the rasterizer's forward and build_mips(): the stale prefetch is dropped
    monkeypatch.setenv("GIGS_LIGHT_PREFETCH", "1")
    torch.manual_seed(2)
    light = pbr.CubemapLight(base_res=64, device=DEV)
    g = {k: tt(sc[k], grad=True) for k in GAUSS_KEYS}
    means2D = torch.zeros_like(g["means3D"], requires_grad=True)
    light.build_mips()  # inline (nothing to adopt): the light now expects to be asked again
    want = [t.detach().clone() for t in [light.diffuse, *light.specular]]
    pipeline.rasterize(camts[0], g, 2, torch.zeros(3, device=DEV), gi, means2D=means2D)
    assert light.prefetch_stats["started"] == 1
    with torch.no_grad():
        light.base.mul_(0.5)
    light.build_mips()
    assert light.prefetch_stats == dict(started=1, adopted=0, discarded=1)
    for a, b in zip([light.diffuse, *light.specular], want):
        torch.testing.assert_close(a, 0.5 * b, rtol=1e-5, atol=1e-7)  # the filter is linear in the base
    pipeline.rasterize(camts[0], g, 2, torch.zeros(3, device=DEV), gi, means2D=means2D)
    assert light.prefetch_stats["started"] == 1  # no further guesses for this light
    torch.cuda.synchronize()


def test_two_steppers_on_two_host_threads_match_serial_runs():
    """Two whole-step-graph steppers in one process, each driven by its own host thread on its own stream (own light, own
    parameter tensors): the scopes of diff_gaussian_rasterization / activations are per thread and the library state per
    context, so the concurrent runs reproduce the serial ones -- losses, images and gradients ('re-entrant per
    stream').  Captures happen one thread at a time (capture mode thread_local), replays concurrently."""
    import threading
    import pbr
    import pipeline
    sc = scenes.surface_scene(P=10_000, sh_degree=2, seed=9, scale_mu=0.025)
    gi = scenes.GI_DEFAULTS
    H, W = 176, 224
    cams = [scenes.orbit_camera(i, 6, W, H, radius=3.5) for i in (1, 4, 2)]
    camts = [{k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    torch.manual_seed(1)
    gt = torch.rand(3, H, W, device=DEV) * 0.5
    lut = pbr.get_brdf_lut().to(DEV)
    rays = pipeline.canonical_rays(cams[0], DEV)
    vds = [pipeline.view_dirs_for(c, rays, DEV) for c in camts]
    n_steps = 6

    class Worker:
        def __init__(self, k):
            self.k, self.stream, self.out, self.err = k, torch.cuda.Stream(), [], None
            torch.manual_seed(20 + k)
            self.light = pbr.CubemapLight(base_res=64, device=DEV)
            self.g = {key: tt(sc[key], grad=True) for key in GAUSS_KEYS}
            with torch.cuda.stream(self.stream):
                self.step = pipeline.Stage2Step(self.light, lut, gi, 2, metallic=True, fused=True, graphs=True)
                self.one(0)  # capture
                torch.cuda.synchronize()

        def one(self, i):
            for p in list(self.g.values()) + [self.light.base]:
                p.grad = None
            vi = (i + self.k) % len(camts)
            o = self.step(camts[vi], self.g, gt, vds[vi])
            self.stream.synchronize()
            return (float(o["loss"]), self.g["albedo"].grad.clone(), self.light.base.grad.clone(), o["render_rgb"].clone())

        def run(self, barrier=None):
            try:
                with torch.cuda.stream(self.stream):
                    if barrier is not None:
                        barrier.wait()
                    self.out = [self.one(i) for i in range(n_steps)]
            except BaseException as ex:  # noqa: BLE001 -- reported by the asserting thread
                self.err = ex
                if barrier is not None:
                    barrier.abort()

    workers = [Worker(0), Worker(1)]
    for w in workers:  # serial reference runs
        w.run()
        assert w.err is None, w.err
    serial = [w.out for w in workers]
    barrier = threading.Barrier(2)
    threads = [threading.Thread(target=w.run, args=(barrier,)) for w in workers]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    for w, ref in zip(workers, serial):
        assert w.err is None, w.err
        for (ls, as_, bs, rs), (lc, ac, bc, rc) in zip(ref, w.out):
            assert abs(ls - lc) <= 2e-6 * max(1.0, abs(ls))
            torch.testing.assert_close(rc, rs, rtol=0, atol=2e-6)
            assert rel_peak(ac.cpu().numpy(), as_.cpu().numpy()) < 2e-3
            assert rel_peak(bc.cpu().numpy(), bs.cpu().numpy()) < 2e-3
    assert abs(serial[0][0][0] - serial[1][0][0]) > 1e-7  # the two workers do render different things
    for w in workers:
        w.step.close()
