"""CPU tests of the oracle-side composition (oracle/stage2_ref.py): the pieces that can be pinned here are --
psnr by the reference's own utils/image_utils.py (tests/golden/ref_psnr.npz), the direction convention of
latlong_to_cubemap by construction, the sRGB pair by being mutual inverses -- and the FMA-contracted oracle twin
stays within the contraction noise floor of the plain build."""
import os

import numpy as np

import scenes
from oracle import stage2_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_psnr_matches_reference_fixture():
    src, ref = np.load(os.path.join(GOLD, "ref_loss.npz")), np.load(os.path.join(GOLD, "ref_psnr.npz"))
    for name in "abcd":
        got = stage2_ref.psnr(src[f"{name}_img"], src[f"{name}_gt"])
        assert abs(got - float(ref[f"{name}_psnr_mean"])) <= 2e-4, (name, got)


def test_srgb_pair_and_texture_lookup():
    x = np.linspace(0, 1, 1001, dtype=np.float32)
    np.testing.assert_allclose(stage2_ref.srgb_to_linear(stage2_ref.linear_to_srgb(x)), x, atol=2e-6)
    assert abs(float(stage2_ref.linear_to_srgb(np.float32(0.0031308))) - 323 / 25 * 0.0031308) < 1e-6
    rng = np.random.default_rng(0)
    tex = rng.uniform(size=(5, 7, 3)).astype(np.float32)
    # texel centres return the texel; u wraps
    uv = np.array([[(3 + 0.5) / 7, (2 + 0.5) / 5], [(0 + 0.5) / 7 + 1.0, (4 + 0.5) / 5]], np.float32)
    got = stage2_ref.texture2d_linear_wrap(tex, uv)
    np.testing.assert_allclose(got[0], tex[2, 3], atol=1e-6)
    np.testing.assert_allclose(got[1], tex[4, 0], atol=1e-5)
    # halfway between two texels: their mean
    mid = stage2_ref.texture2d_linear_wrap(tex, np.array([[4.0 / 7, 2.5 / 5]], np.float32))[0]
    np.testing.assert_allclose(mid, 0.5 * (tex[2, 3] + tex[2, 4]), atol=1e-6)


def test_latlong_to_cubemap_convention():
    env = np.zeros((64, 128, 3), np.float32)
    env[32, 64] = 100.0  # tu = tv = 0.5 -> direction (0, 0, -1): face 5 (relight.py:87-88, :102-103)
    assert stage2_ref.latlong_to_cubemap(env, [32, 32]).reshape(6, -1).sum(1).argmax() == 5
    env[:] = 0
    env[0, :] = 50.0     # tv -> 0: +y, face 2
    assert stage2_ref.latlong_to_cubemap(env, [32, 32]).reshape(6, -1).sum(1).argmax() == 2
    env[:] = 0
    env[32, 96] = 100.0  # tu = 0.75 -> atan2(x, -z) = pi/2 -> +x, face 0
    assert stage2_ref.latlong_to_cubemap(env, [32, 32]).reshape(6, -1).sum(1).argmax() == 0
    # a constant map stays constant
    c = stage2_ref.latlong_to_cubemap(np.full((16, 32, 3), 0.7, np.float32), [8, 8])
    np.testing.assert_allclose(c, 0.7, atol=1e-6)


def test_gbuffer_post_pad_normal_semantics(orc):
    H, W = 6, 7
    rng = np.random.default_rng(1)
    nm = rng.normal(size=(3, H, W)).astype(np.float32)
    nm[:, 0, 0] = 0
    nfd = rng.normal(size=(3, H, W)).astype(np.float32)
    nfd[:, 1, 1] = 0
    op = rng.uniform(size=(1, H, W)).astype(np.float32)
    op[0, 0, 0], op[0, 2, 2] = 0.001, 0.999
    r = dict(normal_map_from_depth=nfd, normal_map=nm, out_normal_view=nm.copy(), opacity_map=op)
    vm = np.eye(4, dtype=np.float32)
    a = stage2_ref.gbuffer_post(orc, r, vm, pad_normal=True)
    assert a["opacity_map"][0, 0, 0] == 0 and a["opacity_map"][0, 2, 2] == 1
    assert not a["normal_mask"][0, 0, 0] and a["normal_mask"][0, 1, 1]          # masks precede the padding
    np.testing.assert_allclose(a["normal_map_from_depth"][:, 1, 1], [0, 0, 1])    # empty depth-normal -> background normal
    b = stage2_ref.gbuffer_post(orc, r, vm, pad_normal=False)
    np.testing.assert_array_equal(b["normal_map_from_depth"][:, 1, 1], [0, 0, 0])
    n = np.linalg.norm(b["normal_map_from_depth"], axis=0)
    assert np.all((np.abs(n - 1) < 1e-5) | (n == 0))


def test_fma_oracle_twin_is_within_the_contraction_noise_floor(orc):
    """The same restated lines compiled with FMA contraction (what nvcc does to the reference): identical integer
    state on this scene and fp planes far inside north_star's 1e-4 (the measured floor is ~1e-7)."""
    fma = orc.variant("fma")
    fma.set_threads(min(8, fma.max_threads()))
    sc = scenes.surface_scene(P=4000, sh_degree=2, seed=2, scale_mu=0.03)
    cam = scenes.orbit_camera(1, 4, 96, 80, radius=3.5)
    gi = scenes.GI_DEFAULTS
    a = stage2_ref.operator_forward(orc, sc, cam, gi, 2, keep_state=True)
    b = stage2_ref.operator_forward(fma, sc, cam, gi, 2, keep_state=True)
    np.testing.assert_array_equal(a["radii"], b["radii"])
    assert a["num_rendered"] == b["num_rendered"]
    differs = 0
    for k in ("render", "depth_map", "albedo_map", "normal_map_from_depth", "depth_pos", "occlusion_map"):
        d = np.abs(np.nan_to_num(a[k]) - np.nan_to_num(b[k]))
        assert d.mean() <= 1e-5, (k, d.mean())
        differs += int((d > 0).any())
    assert differs > 0, "the twin build did not contract anything: it is not a second representative"
