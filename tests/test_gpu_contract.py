"""Operator-contract sweep against the oracle: the parameters of the reference's operator API that the other parity
tests hold fixed.

    scale_modifier != 1      forward.cu:127-161 (computeCov3D scales by it), backward.cu:283-300 (dL_dscale carries it)
    derive_normal = False    R/diff_gaussian_rasterization/__init__.py:486-490 (zeros, then bilateral / median / SSAO)
    debug = True             CHECK_CUDA after every stage (auxiliary.h:178-185), snapshot dumps (__init__.py:115-139, 299-302)
    prefiltered = True       only a trap on an impossible state in the reference (auxiliary.h:167-171): same results
    delta = 0.03125          65 x 32 = 2 080 rays (forward.cu:679-681), exact and default march
"""
import os

import numpy as np
import pytest
import torch

import scenes
from helpers import GAUSS_KEYS, focal, oracle_forward
from oracle import stage2_ref
from test_gpu_parity import (DEV, L1_TOL, PLANES, _backward_pair, _check_grads, _dgr, check_forward, hip_planes,
                             hip_raw_forward, settings, tt)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scale_modifier", [0.6, 1.7])
def test_scale_modifier_forward_and_backward(orc, scale_modifier):
    """scale_modifier enters the 3D covariance (forward.cu:127-161) and the scale gradient (backward.cu:283-300): integer
    state and per-Gaussian state bit-exact, planes and every gradient within the usual bars -- and the footprints really
    change (radii differ from the scale_modifier = 1 run)."""
    dgr = _dgr()
    sc = scenes.surface_scene(P=12_000, sh_degree=2, seed=7, scale_mu=0.03)
    cam = scenes.orbit_camera(1, 6, 203, 165, radius=3.4)
    _, ref, res, _ = check_forward(orc, sc, cam, tag=f"scale_modifier {scale_modifier} ", scale_modifier=scale_modifier)
    base = hip_raw_forward(dgr, sc, cam)
    assert not np.array_equal(hip_planes(base)["radii"], hip_planes(res)["radii"]), "scale_modifier had no effect"
    assert (res[0] < base[0]) == (scale_modifier < 1.0)
    sc2 = scenes.random_scene(P=2500, sh_degree=3, seed=31, scale_mu=0.06)
    cam2 = scenes.orbit_camera(2, 5, 131, 117)
    got, want, _ = _backward_pair(orc, sc2, cam2, (0.2, 0.1, 0.4), seed=5, scale_modifier=scale_modifier)
    _check_grads(got, want, tag=f"scale_modifier {scale_modifier} ")
    one, _, _ = _backward_pair(orc, sc2, cam2, (0.2, 0.1, 0.4), seed=5)
    assert np.abs(got["scales"] - one["scales"]).max() > 1e-3 * np.abs(one["scales"]).max()
    # precomputed covariances ignore it (forward.cu:254-262): same bits as scale_modifier = 1
    r1, _ = oracle_forward(orc, sc2, cam2)
    cov = r1.state("cov3D").reshape(-1, 6)
    a = hip_planes(hip_raw_forward(dgr, sc2, cam2, cov3D_precomp=cov))
    b = hip_planes(hip_raw_forward(dgr, sc2, cam2, cov3D_precomp=cov, scale_modifier=scale_modifier))
    for k in PLANES + ["radii"]:
        np.testing.assert_array_equal(a[k].view(np.uint32), b[k].view(np.uint32), err_msg=k)


def _operator(dgr, sc, cam, gi, derive_normal=True, **skw):
    t = {k: tt(sc[k]) for k in GAUSS_KEYS}
    st = settings(dgr, cam, sc["sh_degree"], gi=gi, **skw)
    with torch.no_grad():
        out = dgr.GaussianRasterizer(st)(t["means3D"], torch.zeros_like(t["means3D"]), t["opacities"], t["normal"], t["albedo"],
                                         t["roughness"], t["metallic"], shs=t["shs"], scales=t["scales"],
                                         rotations=t["rotations"], derive_normal=derive_normal)
    torch.cuda.synchronize()
    names = ("render", "radii", "opacity_map", "depth_map", "normal_map_from_depth", "normal_map", "occlusion_map",
             "albedo_map", "roughness_map", "metallic_map", "out_normal_view", "depth_pos")
    return {k: v.cpu().numpy() for k, v in zip(names, out)}


def _planes_close(got, ref, names, tag, tol=L1_TOL):
    for k in names:
        a, b = got[k], ref[k]
        assert np.array_equal(np.isnan(a), np.isnan(b)), f"{tag}{k}: NaN pattern"
        d = np.abs(np.nan_to_num(a) - np.nan_to_num(b))
        assert d.mean() <= tol, f"{tag}{k}: mean L1 {d.mean():.3e}"


def test_derive_normal_false_matches_the_reference_sequence(orc):
    """derive_normal=False (R/.../__init__.py:486-490): normal_from_depth = bilateral(zeros) = zeros, depth_pos_filter =
    median(zeros) = zeros, and SSAO marches from position 0 with the raw view-space normal (every sample projects through
    z = 0 + ...: the occlusion plane is whatever the reference's arithmetic gives there).  Against the oracle's composition
    of the same sequence, with both marches; the rasterizer's own planes are those of derive_normal=True."""
    import gigs_lib
    dgr = _dgr()
    sc = scenes.surface_scene(P=9_000, sh_degree=1, seed=4, scale_mu=0.03)
    cam = scenes.orbit_camera(1, 5, 208, 160, radius=3.5)
    gi = scenes.GI_DEFAULTS
    ref = stage2_ref.operator_forward(orc, sc, cam, gi, 1, derive_normal=False)
    on = _operator(dgr, sc, cam, gi, derive_normal=True)
    for march in ("proj", "exact"):
        with gigs_lib.options(gi_march=march):
            got = _operator(dgr, sc, cam, gi, derive_normal=False)
        assert np.array_equal(got["radii"], ref["radii"])
        assert not got["normal_map_from_depth"].any() and not got["depth_pos"].any()
        assert not ref["normal_map_from_depth"].any() and not ref["depth_pos"].any()
        _planes_close(got, ref, ("render", "opacity_map", "depth_map", "normal_map", "albedo_map", "roughness_map",
                                 "metallic_map", "out_normal_view", "occlusion_map"), f"derive_normal=False {march} ")
        for k in ("render", "depth_map", "normal_map", "out_normal_view", "albedo_map"):
            np.testing.assert_array_equal(got[k].view(np.uint32), on[k].view(np.uint32), err_msg=k)
    assert on["normal_map_from_depth"].any() and on["depth_pos"].any()


def test_prefiltered_flag_changes_nothing(orc):
    """`prefiltered` only arms a trap for an impossible state in the reference (auxiliary.h:167-171): same bits."""
    dgr = _dgr()
    sc = scenes.surface_scene(P=6_000, sh_degree=2, seed=2, scale_mu=0.03)
    cam = scenes.orbit_camera(0, 4, 160, 128, radius=3.5)
    a, b = _operator(dgr, sc, cam, scenes.GI_DEFAULTS), _operator(dgr, sc, cam, scenes.GI_DEFAULTS, prefiltered=True)
    for k in a:
        np.testing.assert_array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k],
                                      b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k], err_msg=k)
    check_forward(orc, sc, cam, tag="prefiltered ", prefiltered=True)


def test_debug_mode_checks_every_stage_and_writes_snapshots(orc, tmp_path, monkeypatch):
    """debug=True: the library synchronises and checks after every stage (CHECK_CUDA, auxiliary.h:178-185) -- same results
    as debug=False, forward and backward --; an error inside the forward / backward is re-raised after the argument tuple
    has been written to snapshot_fw.dump / snapshot_bw.dump in the working directory (R/.../__init__.py:115-139, 299-302)."""
    dgr = _dgr()
    monkeypatch.chdir(tmp_path)
    sc = scenes.surface_scene(P=5_000, sh_degree=2, seed=3, scale_mu=0.03)
    cam = scenes.orbit_camera(0, 4, 144, 112, radius=3.5)
    plain = hip_planes(hip_raw_forward(dgr, sc, cam))
    dbg = hip_planes(hip_raw_forward(dgr, sc, cam, debug=True))
    for k in plain:
        np.testing.assert_array_equal(plain[k].view(np.uint32) if plain[k].dtype == np.float32 else plain[k],
                                      dbg[k].view(np.uint32) if dbg[k].dtype == np.float32 else dbg[k], err_msg=k)

    def run(debug, sh_degree=2, fail_backward=False):
        t = {k: tt(sc[k], grad=True) for k in GAUSS_KEYS}
        st = settings(dgr, cam, sh_degree, debug=debug)
        out = dgr.GaussianRasterizer(st)(t["means3D"], torch.zeros_like(t["means3D"], requires_grad=True), t["opacities"],
                                         t["normal"], t["albedo"], t["roughness"], t["metallic"], shs=t["shs"],
                                         scales=t["scales"], rotations=t["rotations"])
        if fail_backward:
            def boom(*a, **k):
                raise RuntimeError("injected backward failure")
            monkeypatch.setattr(dgr._C, "rasterize_gaussians_backward", boom)
        (out[0].sum() + out[7].sum()).backward()
        torch.cuda.synchronize()
        return {k: v.grad.cpu().numpy() for k, v in t.items()}

    g0, g1 = run(False), run(True)
    for k in g0:
        np.testing.assert_allclose(g1[k], g0[k], rtol=2e-4, atol=1e-6 * max(np.abs(g0[k]).max(), 1e-20), err_msg=k)
    assert not os.path.exists("snapshot_fw.dump") and not os.path.exists("snapshot_bw.dump")
    # forward: an SH degree the coefficient count cannot hold is rejected by the library (GIGS_ERR_INVALID)
    with pytest.raises(Exception, match="SH degree"):
        run(True, sh_degree=3)
    assert os.path.exists("snapshot_fw.dump")
    snap = torch.load("snapshot_fw.dump", weights_only=True)
    assert len(snap) == 25 and snap[1].shape == (5_000, 3) and not snap[1].is_cuda  # the 25 positional arguments, on the CPU
    # without debug the same error is raised and nothing is written
    os.remove("snapshot_fw.dump")
    with pytest.raises(Exception, match="SH degree"):
        run(False, sh_degree=3)
    assert not os.path.exists("snapshot_fw.dump")
    with pytest.raises(RuntimeError, match="injected backward failure"):
        run(True, fail_backward=True)
    assert os.path.exists("snapshot_bw.dump")
    snap = torch.load("snapshot_bw.dump", weights_only=True)
    assert len(snap) == 31


@pytest.mark.parametrize("march", ["exact", "proj"])
def test_gi_at_delta_0_03125(orc, march):
    """delta = 0.03125: 65 azimuths x 32 elevations = 2 080 rays (loops at forward.cu:679-681), 65 of them
    of zero weight.  SSAO and SSR against the oracle on identical inputs, exact and default march."""
    import gigs_lib
    dgr = _dgr()
    sc = scenes.surface_scene(P=20_000, sh_degree=1, seed=3, scale_mu=0.02)
    cam = scenes.orbit_camera(0, 4, 144, 112, radius=3.5)
    W, H = 144, 112
    fx, fy = focal(cam)
    gi = dict(scenes.GI_DEFAULTS, delta=0.03125)
    r, ref = oracle_forward(orc, sc, cam)
    depth_f = orc.median3x3(ref["depth"])
    nd, pos = orc.depth_to_normal(W, H, fx, fy, cam["viewmatrix"], depth_f)
    posf = orc.median3x3(pos)
    a = (gi["radius"], gi["bias"], gi["thick"], gi["delta"], gi["step"], gi["start"])
    rgb = np.clip(ref["albedo"], 0, 1).astype(np.float32)
    F0 = np.full((3, H, W), 0.04, np.float32)
    occ_ref = orc.ssao(W, H, fx, fy, *a, ref["normal_view"], posf)
    col_ref, abd_ref = orc.ssr(W, H, fx, fy, *a, ref["normal_view"], posf, rgb, ref["albedo"], ref["roughness"], ref["metallic"], F0)
    with gigs_lib.options(gi_march=march):
        occ = dgr._C.SSAO(W, H, fx, fy, *a, tt(ref["normal_view"]), tt(posf)).cpu().numpy()
        col, abd = dgr._C.SSR(W, H, fx, fy, *a, tt(ref["normal_view"]), tt(posf), tt(rgb), tt(ref["albedo"]), tt(ref["roughness"]),
                              tt(ref["metallic"]), tt(F0))
    col, abd = col.cpu().numpy(), abd.cpu().numpy()
    tol = L1_TOL if march == "exact" else 2.5e-5
    for name, x, y in (("occlusion", occ, occ_ref), ("color", col, col_ref), ("abd", abd, abd_ref)):
        assert np.array_equal(np.isnan(x), np.isnan(y)), name
        d = np.abs(np.nan_to_num(x) - np.nan_to_num(y))
        assert d.mean() <= tol, f"{march} {name}: mean L1 {d.mean():.3e}"
        assert (d > 1e-5).mean() <= (1e-3 if march == "exact" else 2e-3), f"{march} {name}: {(d > 1e-5).mean():.2e} of the pixels moved"
    assert (occ_ref < 1.0).sum() > 100, "no occlusion in the test view"


def test_materials_only_backward_writes_the_declared_set_and_counts_violations():
    """gigs_ctx_set_materials_only (gigs-hip extension): with a loss that reaches the albedo / roughness / metallic planes only
    -- whose blend gradients do not feed dL/dalpha (backward.cu:580-590) -- the declared backward returns those three
    gradients as the complete backward computes them and nothing else (None: no tensor, nothing written), and its device-side
    check stays at zero; a loss that also reaches the colour plane is counted as a violation."""
    import gigs_lib
    dgr = _dgr()
    sc = scenes.surface_scene(P=6000, sh_degree=2, seed=4, scale_mu=0.04)
    cam = scenes.orbit_camera(2, 6, 160, 128, radius=3.4)
    st = settings(dgr, cam, 2)
    torch.manual_seed(3)
    w_alb, w_rm = torch.randn(3, 128, 160, device=DEV), torch.randn(2, 128, 160, device=DEV)

    def run(counter, colour):
        t = {k: tt(sc[k], grad=True) for k in GAUSS_KEYS}
        m2d = torch.zeros_like(t["means3D"], requires_grad=True)
        scope = gigs_lib.use(gigs_lib.current().derive(materials_only=counter)) if counter is not None else gigs_lib.use(gigs_lib.current())
        with scope:
            out = dgr.GaussianRasterizer(st)(t["means3D"], m2d, t["opacities"], t["normal"], t["albedo"], t["roughness"],
                                             t["metallic"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        loss = (out[7] * w_alb).sum() + (out[8] * w_rm[:1]).sum() + (out[9] * w_rm[1:]).sum()
        if colour:
            loss = loss + out[0].sum()
        loss.backward()  # the node carries the forward's context to autograd's thread
        torch.cuda.synchronize()
        return {k: (None if v.grad is None else v.grad.detach().cpu().numpy()) for k, v in t.items()}, m2d.grad

    full, m2d_full = run(None, False)
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    got, m2d_got = run(counter, False)
    assert int(counter.item()) == 0
    for k in ("albedo", "roughness", "metallic"):
        assert np.abs(full[k]).max() > 0
        d = np.abs(got[k] - full[k]).max() / np.abs(full[k]).max()
        assert d < 1e-5, (k, d)  # float-atomic sums: rounding differs from run to run
    for k in ("means3D", "opacities", "normal", "shs", "scales", "rotations"):
        assert got[k] is None, k
        assert float(np.abs(full[k]).max()) == 0.0, k  # what was not materialised is an exact zero in the complete backward
    assert torch.equal(m2d_got, m2d_full)  # zeros (and the densification slot)
    run(counter, True)
    assert int(counter.item()) > 0  # the colour plane's gradient reaches dL/dalpha, SH and geometry: counted


def test_split_sh_forward_reads_the_two_tensors_and_equals_the_concatenated_one():
    """gigs_ctx_set_split_sh (gigs-hip extension): `shs` = the optimizer's degree-0 tensor [P,1,3], coefficients 1..M-1 read
    from its _features_rest tensor [P,M-1,3] -- every output plane and the per-Gaussian colours bit for bit those of the
    concatenated [P,M,3] input (SH degree 3 and 1; P not a multiple of the 256-Gaussian block); its backward is the
    materials-only one, any other is refused."""
    import gigs_lib
    dgr = _dgr()
    for deg, P in ((3, 5003), (1, 4097)):
        sc = scenes.surface_scene(P=P, sh_degree=deg, seed=6, scale_mu=0.04)
        cam = scenes.orbit_camera(1, 6, 160, 128, radius=3.4)
        st = settings(dgr, cam, deg)
        t = {k: tt(sc[k]) for k in GAUSS_KEYS}
        dc, rest = t["shs"][:, :1, :].contiguous(), t["shs"][:, 1:, :].contiguous()

        def fwd(shs, **ctx_kw):
            a = {k: v.clone().requires_grad_(True) for k, v in t.items() if k != "shs"}
            with gigs_lib.use(gigs_lib.current().derive(**ctx_kw)):
                out = dgr.GaussianRasterizer(st)(a["means3D"], torch.zeros_like(a["means3D"], requires_grad=True), a["opacities"],
                                                 a["normal"], a["albedo"], a["roughness"], a["metallic"], shs=shs,
                                                 scales=a["scales"], rotations=a["rotations"])
            torch.cuda.synchronize()
            return out, a

        ref, _ = fwd(t["shs"])
        counter = torch.zeros(1, dtype=torch.int32, device=DEV)
        got, a = fwd(dc, sh_rest=rest, materials_only=counter)
        for i, (x, y) in enumerate(zip(ref, got)):
            assert torch.equal(torch.nan_to_num(x.float(), nan=-7.0), torch.nan_to_num(y.float(), nan=-7.0)), (deg, i)
        (got[7].sum() + got[8].sum() + got[9].sum()).backward()
        torch.cuda.synchronize()
        assert int(counter.item()) == 0 and float(a["albedo"].grad.abs().sum()) > 0 and a["opacities"].grad is None
        bad, _ = fwd(dc, sh_rest=rest)  # split SH without the materials-only declaration: the forward is fine ...
        with pytest.raises(Exception, match="materials-only"):
            bad[7].sum().backward()     # ... a complete backward is refused (it would have to read and write SH)
        torch.cuda.synchronize()
