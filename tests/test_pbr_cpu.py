"""CPU tests of the shade / cubemap-light oracle: the C restatement (oracle/pbr_oracle.cpp)
against the independent float64 torch restatement (oracle/torch_pbr_ref.py), plus structural
properties of the sampling rule.  The texture lookups are nvdiffrast calls in the reference
(third party, absent, unpinned): PARITY UNPINNED for those; the cubemap filters restate the
reference's own CUDA (pbr/renderutils/c_src/cubemap.cu)."""
import numpy as np
import pytest
import torch

from oracle import torch_pbr_ref as tp


def make_light(rng, base=32):
    base_map = rng.uniform(0.2, 1.0, size=(6, base, base, 3)).astype(np.float32)
    return base_map


def make_gbuffer(rng, H, W):
    n = rng.normal(size=(H, W, 3)); n /= np.linalg.norm(n, axis=-1, keepdims=True)
    v = rng.normal(size=(H, W, 3)); v /= np.linalg.norm(v, axis=-1, keepdims=True)
    g = dict(normals=n, view_dirs=v, albedo=rng.uniform(0, 1, (H, W, 3)), roughness=rng.uniform(0.04, 1.0, (H, W, 1)),
             mask=rng.uniform(size=(H, W, 1)) > 0.2, occlusion=rng.uniform(0.3, 1, (H, W, 1)),
             metallic=rng.uniform(0, 1, (H, W, 1)))
    # exercise axis-aligned and cube-corner directions too
    g["normals"][0, 0] = [1, 0, 0]; g["normals"][0, 1] = [0, -1, 0]; g["normals"][0, 2] = np.array([1, 1, 1]) / np.sqrt(3)
    g["normals"][0, 3] = np.array([-1, 1, -1]) / np.sqrt(3); g["normals"][0, 4] = np.array([1, 1, 0]) / np.sqrt(2)
    return {k: (v.astype(np.float32) if v.dtype != bool else v) for k, v in g.items()}


def light_levels(orc, base_map, cutoff=0.99):
    """CubemapLight.build_mips with the C oracle (pbr/light.py:154-170)."""
    import importlib
    importlib.import_module("gi-gs_amd")
    from pbr.renderutils.ops import _ndf_cutoff
    spec = [base_map]
    while spec[-1].shape[1] > 16:
        spec.append(orc.cubemap_mip_fwd(spec[-1]))
    diffuse = orc.diffuse_cubemap_fwd(spec[-1])
    L = len(spec)
    out = []
    for idx in range(L):
        rough = (idx / (L - 2)) * (0.5 - 0.08) + 0.08 if idx < L - 1 else 1.0
        cc = _ndf_cutoff(rough, cutoff)
        b = orc.specular_bounds(spec[idx].shape[1], cc)
        rgbw = orc.specular_cubemap_fwd(spec[idx], b, rough, cc)
        out.append((rgbw[..., :3] / rgbw[..., 3:]).astype(np.float32))
    return diffuse, out


def test_cubemap_filters_match_float64_restatement(orc):
    rng = np.random.default_rng(0)
    cm = rng.uniform(0, 1, size=(6, 16, 16, 3)).astype(np.float32)
    d = orc.diffuse_cubemap_fwd(cm)
    np.testing.assert_allclose(d, tp.diffuse_cubemap(tp.to64(cm)).numpy(), rtol=2e-4, atol=2e-5)
    # backward == transpose of the same linear map
    g = rng.normal(size=(6, 16, 16, 3)).astype(np.float32)
    x = tp.to64(cm).requires_grad_(True)
    (tp.diffuse_cubemap(x) * tp.to64(g)).sum().backward()
    np.testing.assert_allclose(orc.diffuse_cubemap_bwd(g), x.grad.numpy(), rtol=2e-4, atol=2e-5)
    import importlib
    importlib.import_module("gi-gs_amd")
    from pbr.renderutils.ops import _ndf_cutoff
    for rough in (1.0, 0.5, 0.29):
        cc = _ndf_cutoff(rough, 0.99)
        b = orc.specular_bounds(16, cc)
        assert b.shape == (6, 16, 16, 24)
        rgbw = orc.specular_cubemap_fwd(cm, b, rough, cc)
        # the AABBs (integer structure, incl. the reference's corner-only tile cull) come from the oracle
        ref = tp.specular_cubemap_rgbw(tp.to64(cm), rough, cc, bounds=b).numpy()
        np.testing.assert_allclose(rgbw, ref, rtol=5e-4, atol=1e-5)
        g4 = rng.normal(size=(6, 16, 16, 4)).astype(np.float32)
        x = tp.to64(cm).requires_grad_(True)
        (tp.specular_cubemap_rgbw(x, rough, cc, bounds=b) * tp.to64(g4)).sum().backward()
        np.testing.assert_allclose(orc.specular_cubemap_bwd(b, g4, rough, cc), x.grad.numpy(), rtol=5e-4, atol=2e-5)
    # the window of texel t contains t itself; empty faces are encoded min > max
    b = orc.specular_bounds(64, 0.95).reshape(6, 64, 64, 6, 4)
    assert np.all(b[0, :, :, 0, 0] <= b[0, :, :, 0, 1])  # +x texels see their own face
    assert np.any(b[..., 0] > b[..., 1])  # and not the opposite one


def test_cubemap_mip(orc):
    rng = np.random.default_rng(1)
    cm = rng.uniform(0, 1, size=(6, 32, 32, 3)).astype(np.float32)
    m = orc.cubemap_mip_fwd(cm)
    ref = torch.nn.functional.avg_pool2d(torch.from_numpy(cm).permute(0, 3, 1, 2), (2, 2)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(m, ref, atol=1e-6)
    d = rng.normal(size=(6, 16, 16, 3)).astype(np.float32)
    np.testing.assert_allclose(orc.cubemap_mip_bwd(d), tp.cubemap_mip_bwd(tp.to64(d)).numpy(), atol=2e-6)
    # a constant gradient stays constant (weights of every lookup sum to one): 0.25 * c
    c = np.full((6, 16, 16, 3), 2.0, np.float32)
    np.testing.assert_allclose(orc.cubemap_mip_bwd(c), 0.5, atol=1e-6)


def test_cube_sampling_rule_properties():
    rng = np.random.default_rng(2)
    d = tp.to64(rng.normal(size=(4000, 3)))
    for res in (16, 64):
        idx, w = tp.cube_taps(res, d)
        assert torch.all((idx >= -1) & (idx < 6 * res * res))
        np.testing.assert_allclose(w.sum(-1).numpy(), 1.0, atol=1e-12)
        # a constant texture samples to the constant, everywhere incl. edges and corners
        tex = torch.full((6, res, res, 3), 0.7, dtype=tp.DT)
        np.testing.assert_allclose(tp.cube_sample(tex, d).numpy(), 0.7, atol=1e-12)
    # exactly at a texel centre the lookup returns that texel
    res = 16
    tex = tp.to64(rng.uniform(size=(6, res, res, 3)))
    dirs = tp.texel_dirs(res)
    np.testing.assert_allclose(tp.cube_sample(tex, dirs).numpy(), tex.reshape(-1, 3).numpy(), atol=1e-9)
    # continuity across a cube edge: directions straddling the +x/+z edge sample nearly equal values
    smooth = dirs[:, 0:1] * 0.3 + dirs[:, 1:2] * 0.2 + 0.5
    tex = smooth.expand(-1, 3).reshape(6, res, res, 3).contiguous()
    a = tp.cube_sample(tex, tp.to64([[1.0, 0.1, 0.999]]))
    b = tp.cube_sample(tex, tp.to64([[0.999, 0.1, 1.0]]))
    assert float((a - b).abs().max()) < 5e-3


@pytest.mark.parametrize("tone,gamma,use_metal", [(False, False, True), (True, True, True), (False, True, False)])
def test_shade_forward_c_oracle_matches_float64_restatement(orc, tone, gamma, use_metal):
    rng = np.random.default_rng(3)
    H, W = 24, 40
    g = make_gbuffer(rng, H, W)
    diffuse, spec = light_levels(orc, make_light(rng, 64))
    assert len(spec) == 3
    lut = np.fromfile(__import__("os").path.join(__import__("os").path.dirname(__file__), "..", "gi-gs_amd", "pbr", "brdf_256_256.bin"),
                      dtype=np.float32).reshape(256, 256, 2)
    bg = rng.uniform(size=(H, W, 3)).astype(np.float32)
    out = orc.shade_fwd(g["normals"], g["view_dirs"], g["albedo"], g["roughness"], g["mask"], g["occlusion"],
                        g["metallic"] if use_metal else None, bg, diffuse, spec, lut, tone=tone, gamma=gamma)
    ref = tp.shade(tp.to64(g["normals"]), tp.to64(g["view_dirs"]), tp.to64(g["albedo"]), tp.to64(g["roughness"]),
                   torch.from_numpy(g["mask"]), tp.to64(g["occlusion"]), tp.to64(g["metallic"]) if use_metal else None,
                   tp.to64(bg), tp.to64(diffuse), [tp.to64(s) for s in spec], tp.to64(lut), tone=tone, gamma=gamma)
    for name, r in zip(("render_rgb", "diffuse_rgb", "specular_rgb", "diffuse_light"), ref):
        np.testing.assert_allclose(out[name], r.numpy(), rtol=2e-4, atol=2e-5, err_msg=name)
    assert np.all(out["render_rgb"][~g["mask"][..., 0]] == bg[~g["mask"][..., 0]])
