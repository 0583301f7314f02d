"""CPU: the C-ABI shared library loads and exports every symbol include/gigs_hip.h declares
(no compute calls without a GPU), and the product never routes through the oracle."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "gigs_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gigs_[a-z0-9_]+)\s*\(", hdr)) - {"gigs_alloc_fn"})


def test_library_exports_every_declared_symbol():
    import gigs_lib
    syms = _declared_symbols()
    assert len(syms) >= 18
    assert sorted(gigs_lib.SIGNATURES) == syms, "gigs_lib.SIGNATURES and include/gigs_hip.h disagree"
    lib = ctypes.CDLL(gigs_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"libgigs_hip.so does not export {s}"
    gl = gigs_lib.lib()
    assert gl.gigs_build_arch() == b"gfx950"
    # pure host arithmetic entry points work without a GPU
    assert gl.gigs_required_image(800, 800) > 800 * 800 * 8
    assert gl.gigs_image_offset(800, 800, 0) == 0 and gl.gigs_binning_offset(1000, 3) > 0


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "gi-gs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                code = "\n".join(l for l in txt.splitlines() if not l.strip().startswith(("#", "//", "*")))
                assert not re.search(r"^\s*(from|import)\s+oracle\b", code, flags=re.M), f
                assert "libgigs_oracle" not in code and "gigs_oracle" not in code, f


def test_cpu_tensors_fail_loudly():
    import pytest
    import torch
    import diff_gaussian_rasterization as dgr
    import scenes
    sc = scenes.random_scene(P=8)
    cam = scenes.orbit_camera(0, 1, 32, 32)
    st = dgr.GaussianRasterizationSettings(32, 32, cam["tanfovx"], cam["tanfovy"], 0.8, 0.01, 0.05, 0.0625, 16, 8,
                                           torch.zeros(3), 1.0, torch.from_numpy(cam["viewmatrix"]),
                                           torch.from_numpy(cam["projmatrix"]), 0, torch.from_numpy(cam["campos"]),
                                           False, False, False, False)
    t = {k: torch.from_numpy(v) for k, v in sc.items() if k != "sh_degree"}
    with pytest.raises(RuntimeError, match="no CPU path"):
        dgr.GaussianRasterizer(st)(t["means3D"], torch.zeros_like(t["means3D"]), t["opacities"], t["normal"], t["albedo"],
                                   t["roughness"], t["metallic"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    with pytest.raises(Exception, match="excatly one of either SHs"):
        dgr.GaussianRasterizer(st)(t["means3D"], None, t["opacities"], t["normal"], t["albedo"], t["roughness"], t["metallic"])
    with pytest.raises(Exception, match="exactly one of either scale/rotation"):
        dgr.GaussianRasterizer(st)(t["means3D"], None, t["opacities"], t["normal"], t["albedo"], t["roughness"], t["metallic"], shs=t["shs"])
