"""CPU tests (no GPU): the oracle against the reference-pinned golden vectors, against the
frozen outputs, against structural invariants, and its backward against autograd of the
independent float64 restatement (oracle/torch_ref.py)."""
import os

import numpy as np
import pytest

import scenes
from helpers import GAUSS_KEYS, oracle_forward, random_pix_grads, small_scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sh_colour_matches_reference_eval_sh(orc):
    """oracle sh_to_rgb (forward.cu:22-80) == reference utils/sh_utils.eval_sh + 0.5, clamp."""
    g = np.load(os.path.join(GOLD, "ref_sh.npz"))
    means, campos = g["sh_means"], g["sh_campos"]
    P = means.shape[0]
    # a camera that sees everything in front of it; colours do not depend on the projection
    cam = scenes.look_at_camera(campos, (0.0, 0.0, 0.0), 64, 64, 1.2)
    cam["campos"] = campos.copy()
    sc = scenes.random_scene(P=P, sh_degree=3, seed=0)
    sc["means3D"] = means
    for deg in range(4):
        sc["shs"] = g[f"sh_deg{deg}_shs"]
        sc["sh_degree"] = deg
        r, out = oracle_forward(orc, sc, cam)
        vis = out["radii"] > 0
        assert vis.sum() > 50
        rgb = r.state("rgb").reshape(P, 3)
        np.testing.assert_allclose(rgb[vis], g[f"sh_deg{deg}_rgb"][vis], rtol=0, atol=2e-5)
        clamped = r.state("clamped").reshape(P, 3).astype(bool)
        ref_clamped = g[f"sh_deg{deg}_clamped"]
        near0 = np.abs(g[f"sh_deg{deg}_rgb"]) < 1e-5
        assert np.all((clamped[vis] == ref_clamped[vis]) | near0[vis])


def test_camera_conventions_match_reference():
    g = np.load(os.path.join(GOLD, "ref_camera.npz"))
    for i in range(3):
        fx, fy = g[f"proj_{i}_fov"]
        np.testing.assert_array_equal(scenes.projection_matrix(0.01, 100.0, fx, fy), g[f"proj_{i}"])
        cam = scenes.orbit_camera(i, 3, 400, 300)
        # viewmatrix = getWorld2View2(R, T).T (scene/cameras.py:75-77)
        np.testing.assert_allclose(cam["viewmatrix"], g[f"w2v_{i}"].T, rtol=0, atol=1e-6)
        # camera centre is the orbit position, |c| = 4
        assert abs(np.linalg.norm(cam["campos"]) - 4.0) < 1e-4
    s = np.load(os.path.join(GOLD, "ref_sh.npz"))
    np.testing.assert_allclose(scenes.rgb2sh(s["rgb2sh_in"]), s["rgb2sh_out"], atol=1e-6)


def test_oracle_reproduces_frozen_outputs(orc):
    g = np.load(os.path.join(GOLD, "oracle_frozen_small.npz"))
    orc.set_threads(1)
    sc = scenes.random_scene(P=600, sh_degree=2, seed=7, scale_mu=0.08)
    cam = scenes.orbit_camera(1, 5, 96, 80)
    r, out = oracle_forward(orc, sc, cam, bg=(0.1, 0.2, 0.3))
    assert out["num_rendered"] == int(g["num_rendered"])
    np.testing.assert_array_equal(out["radii"], g["radii"])
    np.testing.assert_array_equal(r.state("point_list"), g["point_list"])
    np.testing.assert_array_equal(r.state("ranges"), g["ranges"])
    np.testing.assert_array_equal(r.state("n_contrib"), g["n_contrib"])
    for k in ("color", "depth", "opacity", "normal"):
        np.testing.assert_allclose(out[k], g[k], rtol=0, atol=1e-6)


def test_ray_counts_fp32_accumulation(orc):
    # 32x16 at delta=0.0625 (the 17th theta lands one ulp above pi/2)
    assert orc.gi_ray_counts(0.0625) == (32, 16)
    assert orc.gi_ray_counts(0.125) == (16, 9)
    assert orc.gi_ray_counts(0.03125) == (65, 32)


def test_higher_msb(orc):
    assert orc.higher_msb(2500) == 12 and orc.higher_msb(625) == 10 and orc.higher_msb(4056) == 12
    assert orc.higher_msb(4346) == 13 and orc.higher_msb(1) == 1


def test_binning_invariants(orc):
    sc, cam = small_scene(P=2000, W=112, H=80, scale_mu=0.05)
    r, out = oracle_forward(orc, sc, cam)
    R = out["num_rendered"]
    keys, pl, ranges = r.state("keys"), r.state("point_list"), r.state("ranges").reshape(-1, 2)
    tt, off = r.state("tiles_touched"), r.state("point_offsets")
    assert R == tt.sum() == off[-1] and R > 0
    assert np.all(np.diff(keys.astype(np.uint64)) >= 0)  # sorted by (tile, depth bits)
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    for t, (lo, hi) in enumerate(ranges):
        assert np.all(tiles[lo:hi] == t)
    assert sum(hi - lo for lo, hi in ranges) == R
    # stability: equal keys keep Gaussian-index order
    same = keys[1:] == keys[:-1]
    assert np.all(pl[1:][same] > pl[:-1][same])
    assert set(np.unique(pl)) <= set(np.nonzero(out["radii"] > 0)[0])
    # culled Gaussians are exactly those behind the near plane or with an empty rect
    zc = sc["means3D"] @ cam["viewmatrix"][:3, 2] + cam["viewmatrix"][3, 2]
    assert np.all(out["radii"][zc <= 0.2] == 0)
    np.testing.assert_array_equal(orc.mark_visible(sc["means3D"], cam["viewmatrix"]), ~(zc.astype(np.float32) <= 0.2))


def test_forward_basic_properties(orc):
    sc, cam = small_scene(P=1500, W=100, H=70, scale_mu=0.06)
    r, out = oracle_forward(orc, sc, cam, bg=(0.2, 0.4, 0.6))
    T = r.state("final_T").reshape(70, 100)
    np.testing.assert_allclose(out["opacity"][0] + T, 1.0, atol=2e-5)  # sum w_i + T == 1
    assert np.all(T >= 1e-4 * 0.99) and np.all(out["opacity"] >= 0)
    empty = out["opacity"][0] <= 1e-6
    assert np.all(out["depth"][0][empty] == 0)
    # empty pixels: colour is the background and the view normal is NaN
    really_empty = r.state("n_contrib").reshape(70, 100) == 0
    if really_empty.any():
        np.testing.assert_allclose(out["color"][:, really_empty], np.array([[0.2], [0.4], [0.6]]) * np.ones((1, really_empty.sum())), atol=1e-6)
        assert np.all(np.isnan(out["normal_view"][:, really_empty]))
    # inference adds T to roughness (forward.cu:612-616)
    _, out_inf = oracle_forward(orc, sc, cam, bg=(0.2, 0.4, 0.6), inference=True)
    np.testing.assert_allclose(out_inf["roughness"][0], out["roughness"][0] + T, atol=1e-6)
    # linearity in the colours: precomputed colours scaled by 2 double the colour plane (bg = 0)
    cols = np.random.default_rng(0).uniform(0, 1, size=(1500, 3)).astype(np.float32)
    _, o1 = oracle_forward(orc, sc, cam, shs=None, colors_precomp=cols)
    _, o2 = oracle_forward(orc, sc, cam, shs=None, colors_precomp=2 * cols)
    np.testing.assert_allclose(o2["color"], 2 * o1["color"], rtol=1e-6, atol=1e-7)


def test_empty_and_degenerate_inputs(orc):
    cam = scenes.orbit_camera(0, 4, 40, 24)
    sc = scenes.random_scene(P=0, sh_degree=0)
    r, out = oracle_forward(orc, sc, cam)
    assert out["num_rendered"] == 0 and np.all(out["color"] == 0)
    # everything behind the camera
    sc = scenes.random_scene(P=50, sh_degree=0)
    sc["means3D"] = (np.asarray(cam["campos"])[None] * 1.5 + 0.01 * sc["means3D"]).astype(np.float32)
    r, out = oracle_forward(orc, sc, cam, bg=(1, 0, 0))
    assert out["num_rendered"] == 0 and np.all(out["radii"] == 0)
    np.testing.assert_allclose(out["color"][0], 1.0)


@pytest.mark.parametrize("deg,seed", [(0, 11), (1, 5), (2, 9), (3, 2)])
def test_backward_matches_autograd_of_float64_restatement(orc, deg, seed):
    from oracle import torch_ref
    orc.set_threads(1)
    W, H = 40, 32
    sc = scenes.random_scene(P=60, sh_degree=deg, seed=seed, extent=0.7, scale_mu=0.12, scale_sigma=0.3)
    cam = scenes.orbit_camera(seed % 4, 4, W, H, radius=3.0)
    bg = (0.3, 0.1, 0.6)
    r, out = oracle_forward(orc, sc, cam, bg=bg)
    rng = np.random.default_rng(seed)
    pg = random_pix_grads(rng, H, W)
    got = r.backward(grad_color=pg["color"], grad_opacity=pg["opacity"], grad_depth=pg["depth"],
                     grad_normal=pg["normal"], grad_albedo=pg["albedo"], grad_roughness=pg["roughness"],
                     grad_metallic=pg["metallic"])
    ref = torch_ref.render_with_grads(sc, cam, np.asarray(bg), r.state("point_list"), r.state("ranges"),
                                      out["radii"] > 0, pg)
    assert not ref["clamped_any"]
    # the float64 forward agrees with the oracle's fp32 forward, including the discrete part
    np.testing.assert_array_equal(ref["n_contrib"].ravel(), r.state("n_contrib"))
    np.testing.assert_allclose(out["color"], ref["color"], atol=3e-5)
    np.testing.assert_allclose(out["opacity"], ref["opacity"], atol=3e-5)
    np.testing.assert_allclose(out["albedo"], ref["albedo"], atol=3e-5)

    def close(a, b, name):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        scale = max(np.abs(b).max(), 1e-12)
        err = np.abs(a - b).max() / scale
        assert err < 2e-3, f"{name}: max rel-to-peak error {err:.3e}"

    close(got["means3D"], ref["d_means3D"], "means3D")
    close(got["means2D"][:, :2], ref["d_means2D"], "means2D")
    close(got["opacity"], ref["d_opacities"], "opacities")
    close(got["normal"], ref["d_normal"], "normal")
    close(got["albedo"], ref["d_albedo"], "albedo")
    close(got["roughness"], ref["d_roughness"], "roughness")
    close(got["metallic"], ref["d_metallic"], "metallic")
    close(got["sh"], ref["d_shs"], "shs")
    close(got["scales"], ref["d_scales"], "scales")
    close(got["rotations"], ref["d_rotations"], "rotations")
    # the abs-gradient accumulator dominates the signed one
    assert np.all(got["means2D"][:, 2] + 1e-6 >= np.abs(got["means2D"][:, 0]))


def test_gi_passes_properties(orc):
    sc = scenes.surface_scene(P=6000, sh_degree=0, seed=1, scale_mu=0.03)
    cam = scenes.orbit_camera(0, 4, 96, 72, radius=3.5)
    r, out = oracle_forward(orc, sc, cam)
    W, H = 96, 72
    fx, fy = W / (2 * cam["tanfovx"]), H / (2 * cam["tanfovy"])
    depth_f = orc.median3x3(out["depth"])
    nd, pos = orc.depth_to_normal(W, H, fx, fy, cam["viewmatrix"], depth_f)
    # border pixels keep depth_pos = 0 (forward.cu:936)
    assert np.all(pos[:, 0, :] == 0) and np.all(pos[:, :, -1] == 0)
    np.testing.assert_allclose(pos[2, 1:-1, 1:-1], depth_f[0, 1:-1, 1:-1])
    posf = orc.median3x3(pos)
    gi = scenes.GI_DEFAULTS
    occ = orc.ssao(W, H, fx, fy, gi["radius"], gi["bias"], gi["thick"], gi["delta"], gi["step"], gi["start"], out["normal_view"], posf)
    assert occ.min() >= 0 and occ.max() <= 1 and occ.mean() < 0.999  # something is occluded
    # README setting start=64 >= step: the march loop is empty -> occlusion == 1, indirect == 0
    occ1 = orc.ssao(W, H, fx, fy, gi["radius"], gi["bias"], gi["thick"], gi["delta"], 16, 64, out["normal_view"], posf)
    assert np.all(occ1 == 1.0)
    F0 = np.full((3, H, W), 0.04, np.float32)
    col, abd = orc.ssr(W, H, fx, fy, gi["radius"], gi["bias"], gi["thick"], gi["delta"], 16, 64, out["normal_view"], posf,
                       out["color"], out["albedo"], out["roughness"], out["metallic"], F0)
    assert np.all(col[~np.isnan(col)] == 0)
    col, abd = orc.ssr(W, H, fx, fy, gi["radius"], gi["bias"], gi["thick"], gi["delta"], gi["step"], gi["start"], out["normal_view"], posf,
                       out["color"], out["albedo"], out["roughness"], out["metallic"], F0)
    ok = ~np.isnan(abd)
    np.testing.assert_allclose(col[ok], (abd * out["albedo"])[ok], rtol=1e-6, atol=1e-9)  # color = gd * albedo
    assert np.nanmax(abd) > 0
    # SSR is linear in the incoming radiance
    col2, _ = orc.ssr(W, H, fx, fy, gi["radius"], gi["bias"], gi["thick"], gi["delta"], gi["step"], gi["start"], out["normal_view"], posf,
                      2 * out["color"], out["albedo"], out["roughness"], out["metallic"], F0)
    np.testing.assert_allclose(col2[ok], 2 * col[ok], rtol=1e-5, atol=1e-9)


def test_filters_definitions(orc):
    rng = np.random.default_rng(0)
    x = rng.normal(size=(2, 9, 11)).astype(np.float32)
    m = orc.median3x3(x)
    pad = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    for (c, y, xx) in [(0, 0, 0), (1, 4, 5), (0, 8, 10), (1, 0, 7)]:
        assert m[c, y, xx] == np.sort(pad[c, y:y + 3, xx:xx + 3].ravel())[4]
    x[0, 3, 3] = np.nan
    m = orc.median3x3(x)
    assert np.isnan(m[0, 2:5, 2:5]).all() and not np.isnan(m[0, 0, 0])
    # bilateral: constant image is a fixed point; idempotent on constants; preserves mean roughly
    c = np.full((3, 8, 8), 0.37, np.float32)
    np.testing.assert_allclose(orc.bilateral3x3(c), c, atol=1e-6)
    y = orc.bilateral3x3(rng.uniform(size=(3, 16, 16)).astype(np.float32))
    assert y.min() >= 0 and y.max() <= 1
