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


def test_contexts_are_independent_and_validated():
    """gigs_ctx (include/gigs_hip.h): host-only API -- options round-trip, two contexts do not see each other, the default
    context is immutable, out-of-range values are rejected.  (The GPU suite drives two contexts on two streams.)"""
    import ctypes as C
    import gigs_lib
    gl = gigs_lib.lib()
    dflt = gigs_lib.Options()
    dflt.struct_bytes = C.sizeof(gigs_lib.Options)
    assert gl.gigs_ctx_get_options(None, C.byref(dflt)) == 0
    assert dflt.gi_march == 4 and dflt.blend_cull == 1 and dflt.gi_zero_rays == 0 and dflt.bucket_target == 1536
    a, b = gl.gigs_ctx_create(), gl.gigs_ctx_create()
    assert a and b and a != b
    oa = gigs_lib.Options()
    oa.struct_bytes = C.sizeof(gigs_lib.Options)
    assert gl.gigs_ctx_get_options(a, C.byref(oa)) == 0
    oa.gi_march, oa.binning_legacy, oa.gi_zero_rays = 0, 1, 1
    assert gl.gigs_ctx_set_options(a, C.byref(oa)) == 0
    ob = gigs_lib.Options()
    ob.struct_bytes = C.sizeof(gigs_lib.Options)
    assert gl.gigs_ctx_get_options(b, C.byref(ob)) == 0
    assert (ob.gi_march, ob.binning_legacy, ob.gi_zero_rays) == (4, 0, 0)  # b and the defaults are untouched
    assert gl.gigs_ctx_get_options(a, C.byref(ob)) == 0 and (ob.gi_march, ob.binning_legacy, ob.gi_zero_rays) == (0, 1, 1)
    # rejected: the default context, a bad value, a short struct
    assert gl.gigs_ctx_set_options(None, C.byref(oa)) < 0 and b"immutable" in gl.gigs_last_error()
    assert gl.gigs_ctx_set_async_binning(None, 10, None) < 0 and gl.gigs_ctx_set_blend_begin_event(None, None) < 0
    oa.gi_march = 9
    assert gl.gigs_ctx_set_options(a, C.byref(oa)) < 0
    assert gl.gigs_ctx_get_options(a, C.byref(ob)) == 0 and ob.gi_march == 0  # nothing changed
    oa.gi_march, oa.struct_bytes = 1, 8
    assert gl.gigs_ctx_set_options(a, C.byref(oa)) < 0
    assert gl.gigs_ctx_set_async_binning(a, 1 << 20, None) == 0 and gl.gigs_ctx_set_async_binning(a, 0, None) == 0
    gl.gigs_ctx_destroy(a)
    gl.gigs_ctx_destroy(b)
    # the Python side: derived contexts are interned, `with options()` nests and restores
    base = gigs_lib.current()
    with gigs_lib.options(gi_march="exact", blend_cull=0) as c1:
        assert gigs_lib.current() is c1 and c1.option("gi_march") == 0 and c1.option("blend_cull") == 0
        with gigs_lib.options(gi_cert=0) as c2:
            assert c2.option("gi_march") == 0 and c2.option("gi_cert") == 0
        assert gigs_lib.current() is c1
        assert gigs_lib.current().derive(gi_march="exact") is c1  # same settings, same native object
    assert gigs_lib.current() is base


def test_no_launch_path_reads_the_environment():
    """the library is re-entrant per stream -- switches live in gigs_ctx; getenv appears only where the default
    options are built (api.hip::options_from_env) and in -DGIGS_DIAG code."""
    csrc = os.path.join(ROOT, "gi-gs_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        txt = open(os.path.join(csrc, f)).read()
        txt = re.sub(r"#ifdef GIGS_DIAG.*?#endif", "", txt, flags=re.S)
        code = "\n".join(l.split("//")[0] for l in txt.splitlines())
        n = len(re.findall(r"\bgetenv\s*\(", code))
        if f == "api.hip":
            body = code[code.index("int env_int("):code.index("const gigs::Ctx& ctx_of")]
            assert n == len(re.findall(r"\bgetenv\s*\(", body)), "getenv outside env_int / options_from_env in api.hip"
        else:
            assert n == 0, f"{f} reads the environment in a launch path"
    assert "gigs_set_async_binning" not in open(os.path.join(ROOT, "include", "gigs_hip.h")).read()
