"""BASELINE config C3 (relight.py: inference-only PBR + indirect under an HDR environment map) on the GPU:
the HIP latlong->cubemap conversion and the per-view relight sequence against the CPU oracle composition
(oracle/stage2_ref.py) at sizes the oracle finishes in seconds, the fused formulation against the op-by-op one,
and size-independent properties at 800x800.

PARITY UNPINNED for the third-party lookups (nvdiffrast dr.texture 2D/cube, kornia median): the oracle's own
definitions are the yardstick, as for the shade."""
import numpy as np
import pytest
import torch

import scenes
from oracle import stage2_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEYS = stage2_ref.KEYS


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def cam_t(cam):
    return {k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}


def view_dirs(cam):
    import pipeline
    return pipeline.view_dirs_for(cam_t(cam), pipeline.canonical_rays(cam, DEV), DEV)


@pytest.mark.parametrize("res,shape", [(16, (32, 64)), (64, (96, 200)), (256, (512, 1024))])
def test_latlong_to_cubemap_matches_oracle(res, shape):
    import relight
    env = scenes.synthetic_envmap(*shape, seed=res)
    got = relight.latlong_to_cubemap(tt(env), [res, res]).cpu().numpy()
    ref = stage2_ref.latlong_to_cubemap(env, [res, res])
    assert got.shape == (6, res, res, 3)
    # same formula; libm vs OCML atan2/acos and the linspace rounding move the lookup by ~1e-6 texels
    d = np.abs(got - ref)
    assert d.mean() <= 1e-5 * max(1.0, float(np.abs(ref).mean())), d.mean()
    assert d.max() <= 2e-3 * float(np.abs(ref).max()), d.max()


def test_latlong_to_cubemap_orientation():
    """A bright texel at a known (tu, tv) must land where relight.py's direction convention puts it: tu = 0.5 is
    -z (face 5), tv = 0 is +y (face 2)."""
    import relight
    env = np.zeros((64, 128, 3), np.float32)
    env[32, 64] = 100.0                      # tu = 0.5, tv = 0.5 -> direction (0, 0, -1)
    cube = relight.latlong_to_cubemap(tt(env), [32, 32]).cpu().numpy()
    per_face = cube.reshape(6, -1).sum(1)
    assert per_face.argmax() == 5
    env[:] = 0
    env[0, :] = 50.0                         # the whole top row: +y pole
    cube = relight.latlong_to_cubemap(tt(env), [32, 32]).cpu().numpy()
    assert cube.reshape(6, -1).sum(1).argmax() == 2


def _relight_case(P, W, H, light_res, seed, view):
    sc = scenes.surface_scene(P=P, sh_degree=2, seed=seed, scale_mu=0.03)
    cam = scenes.orbit_camera(view, 8, W, H, radius=3.5)
    env = scenes.synthetic_envmap(128, 256, seed=seed + 1)
    return sc, cam, env


@pytest.mark.parametrize("metallic,ratio,pad", [(False, None, False), (True, (0.9, 1.1, 0.8), False), (False, None, True)])
def test_relight_view_matches_oracle(orc, metallic, ratio, pad):
    import relight
    W, H, res = 176, 144, 64
    sc, cam, env = _relight_case(9000, W, H, res, seed=4, view=1)
    gi = scenes.GI_DEFAULTS
    rng = np.random.default_rng(0)
    alpha = (rng.uniform(size=(1, H, W)) > 0.1).astype(np.float32)
    light = relight.make_light(tt(env), res=res)
    np.testing.assert_allclose(light.base.detach().cpu().numpy(), stage2_ref.latlong_to_cubemap(env, [res, res]), rtol=2e-3, atol=1e-4)
    g = {k: tt(sc[k]) for k in KEYS}
    vd = view_dirs(cam)
    # oracle: the light levels from the SAME base cubemap the GPU uses (the conversion has its own test above)
    base = light.base.detach().cpu().numpy()
    diffuse, spec = stage2_ref.build_mips(orc, base)
    ref = stage2_ref.relight_view(orc, sc, cam, gi, 2, diffuse, spec, alpha_mask=alpha, albedo_ratio=ratio or (1, 1, 1),
                                  metallic=metallic, pad_normal=pad)
    for fused in ((False,) if pad else (False, True)):
        rl = relight.Relighter(light, gi, 2, metallic=metallic, fused=fused, pad_normal=pad)
        out = rl(cam_t(cam), g, vd, alpha_mask=tt(alpha), albedo_ratio=ratio)
        for k in ("render_direct", "IRR", "render_rgb", "occlusion"):
            a, b = out[k].cpu().numpy(), ref[k]
            assert np.array_equal(np.isnan(a), np.isnan(b)), (fused, k)
            d = np.abs(np.nan_to_num(a) - np.nan_to_num(b))
            assert d.mean() <= 1e-4, (fused, k, d.mean())   # north_star: 1e-4 mean per-pixel L1
        assert stage2_ref.psnr(np.nan_to_num(out["render_rgb"].cpu().numpy()), np.nan_to_num(ref["render_rgb"])) >= 60.0
    assert float(np.nan_to_num(ref["render_rgb"]).max()) > 0.2 and float(np.nan_to_num(ref["IRR"]).max()) > 0


def test_relight_fused_equals_unfused_c3_size():
    """800x800 / 300k Gaussians / 256^2 light (BASELINE configs[2]): fused == op-by-op, alpha mask honoured,
    occlusion in [0, 1], mips built once (the light's levels are not touched by a view)."""
    import relight
    W = H = 800
    sc = scenes.surface_scene(P=300_000, sh_degree=2, seed=0)
    cam = scenes.orbit_camera(11, 64, W, H, radius=3.5)
    env = scenes.synthetic_envmap(512, 1024, seed=1)
    gi = scenes.GI_DEFAULTS
    light = relight.make_light(tt(env), res=256)
    g = {k: tt(sc[k]) for k in KEYS}
    vd = view_dirs(cam)
    yy, xx = np.mgrid[0:H, 0:W]
    alpha = (((xx - 400) ** 2 + (yy - 400) ** 2) < 380 ** 2).astype(np.float32)[None]
    a = relight.Relighter(light, gi, 2, metallic=False, fused=True)
    spec_ptrs = [s.data_ptr() for s in light.specular]
    oa = a(cam_t(cam), g, vd, alpha_mask=tt(alpha))
    assert [s.data_ptr() for s in light.specular] == spec_ptrs
    b = relight.Relighter(light, gi, 2, metallic=False, fused=False)
    ob = b(cam_t(cam), g, vd, alpha_mask=tt(alpha))
    for k in ("render_direct", "IRR", "render_rgb"):
        x, y = oa[k].nan_to_num(), ob[k].nan_to_num()
        assert float((x - y).abs().max()) <= 2e-6, k
    # the whole view replayed from ONE hipGraph (asynchronous binning): the same image, also for a second camera pose
    c = relight.Relighter(light, gi, 2, metallic=False, fused=True, graphs=True)
    keep = {k: oa[k].clone() for k in ("render_direct", "IRR", "render_rgb")}
    cam2 = scenes.orbit_camera(30, 64, W, H, radius=3.5)
    vd2 = view_dirs(cam2)
    want2 = {k: v.clone() for k, v in a(cam_t(cam2), g, vd2, alpha_mask=tt(alpha)).items() if k in keep}
    for cm, v, want in ((cam, vd, keep), (cam2, vd2, want2), (cam, vd, keep)):
        oc = c(cam_t(cm), g, v, alpha_mask=tt(alpha))
        assert oc["num_rendered"] > 1_000_000
        for k in want:
            assert float((oc[k].nan_to_num() - want[k].nan_to_num()).abs().max()) <= 2e-6, k
    rgb = oa["render_rgb"]
    assert float(rgb[:, tt(alpha)[0] == 0].abs().max()) == 0.0
    assert torch.isfinite(rgb).all()
    occ = oa["occlusion"]
    assert float(occ.min()) >= 0.0 and float(occ.max()) <= 1.0 and float(occ.min()) < 1.0
