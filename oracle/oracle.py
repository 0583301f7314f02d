"""ctypes front-end of the CPU oracle (oracle/gigs_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package (gi-gs_amd/).

All arrays are numpy, C-contiguous, float32 unless stated.  Function names and argument
order mirror the `_C` functions of the reference extension
(R/ext.cpp:16-23, R/rasterize_points.cu) so that tests read like calls to the reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgigs_oracle.so")
_lib = None

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)


def build(force: bool = False) -> str:
    """Compile the oracle with the recipe in oracle/Makefile (g++ -O2 -ffp-contract=off; plus the
    FMA-contracted twin libgigs_oracle_fma.so)."""
    srcs = [os.path.join(_HERE, f) for f in ("gigs_oracle.cpp", "pbr_oracle.cpp")]
    libs = [os.path.join(_HERE, f) for f in ("libgigs_oracle.so", "libgigs_oracle_fma.so")]
    if force or any(not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs) for so in libs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def variant(name: str):
    """A second, independent copy of this module bound to another build of the same sources:
    variant("fma") = libgigs_oracle_fma.so (FMA contraction on, like nvcc's default for the reference's
    CUDA binary).  Used to measure how far two legitimate compilations of the cited lines are apart."""
    import importlib.util
    build()
    spec = importlib.util.spec_from_file_location("oracle_%s" % name, os.path.abspath(__file__))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod._LIB_PATH = os.path.join(_HERE, "libgigs_oracle_%s.so" % name)
    if not os.path.exists(mod._LIB_PATH):
        raise FileNotFoundError(mod._LIB_PATH)
    return mod


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_destroy.argtypes = [C.c_void_p]
        _lib.orc_state.restype = C.c_size_t
        _lib.orc_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib.orc_higher_msb.restype = C.c_uint32
        _lib.orc_higher_msb.argtypes = [C.c_uint32]
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def set_threads(n: int) -> None:
    lib().orc_set_threads(C.c_int(n))


def max_threads() -> int:
    return lib().orc_max_threads()


def _f(a: Optional[np.ndarray]):
    """float32 pointer or NULL for None / empty (the reference selects kernel branches on
    null data_ptr of empty tensors: R/diff_gaussian_rasterization/__init__.py:435-445)."""
    if a is None or a.size == 0:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(c_float_p)


def _c32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


_STATE = {
    "depths": (0, np.float32), "pos_view": (1, np.float32), "means2D": (2, np.float32),
    "cov3D": (3, np.float32), "conic_opacity": (4, np.float32), "rgb": (5, np.float32),
    "clamped": (6, np.uint8), "tiles_touched": (7, np.uint32), "point_offsets": (8, np.uint32),
    "keys_unsorted": (9, np.uint64), "vals_unsorted": (10, np.uint32), "keys": (11, np.uint64),
    "point_list": (12, np.uint32), "ranges": (13, np.uint32), "final_T": (14, np.float32),
    "n_contrib": (15, np.uint32), "dL_dconic": (16, np.float32), "dL_ddepth": (17, np.float32),
}


class Rasterizer:
    """Holds the geometry / binning / image state between forward and backward, like the three
    byte buffers the reference keeps in the autograd ctx (…/__init__.py:161-176)."""

    def __init__(self):
        self._h = C.c_void_p(lib().orc_create())
        self._fw = None

    def __del__(self):
        try:
            lib().orc_destroy(self._h)
        except Exception:
            pass

    def state(self, name: str) -> np.ndarray:
        which, dt = _STATE[name]
        n = lib().orc_state(self._h, which, None)
        out = np.zeros(n, dtype=dt)
        if n:
            lib().orc_state(self._h, which, out.ctypes.data_as(C.c_void_p))
        return out

    def counters(self) -> Dict[str, int]:
        buf = (C.c_uint64 * 3)()
        lib().orc_counters(self._h, buf)
        return {"R": int(buf[0]), "pairs_evaluated": int(buf[1]), "pairs_contributing": int(buf[2])}

    def forward(self, *, bg, means3D, opacities, normal, albedo, roughness, metallic,
                viewmatrix, projmatrix, campos, tanfovx, tanfovy, image_height, image_width,
                sh_degree=0, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, scale_modifier=1.0, argmax_depth=False, inference=False):
        """_C.rasterize_gaussians (rasterize_points.cu:130-252).  Returns a dict with
        num_rendered, color, radii, opacity, depth, normal, normal_view, pos, albedo,
        roughness, metallic."""
        P = int(means3D.shape[0])
        H, W = int(image_height), int(image_width)
        a = dict(bg=_c32(bg), means3D=_c32(means3D), opacities=_c32(opacities), normal=_c32(normal),
                 albedo=_c32(albedo), roughness=_c32(roughness), metallic=_c32(metallic),
                 viewmatrix=_c32(viewmatrix), projmatrix=_c32(projmatrix), campos=_c32(campos),
                 shs=None if shs is None else _c32(shs),
                 colors_precomp=None if colors_precomp is None else _c32(colors_precomp),
                 scales=None if scales is None else _c32(scales),
                 rotations=None if rotations is None else _c32(rotations),
                 cov3D_precomp=None if cov3D_precomp is None else _c32(cov3D_precomp))
        M = 0 if a["shs"] is None or a["shs"].size == 0 else int(a["shs"].shape[1])
        out = {k: np.zeros((c, H, W), np.float32) for k, c in
               [("color", 3), ("opacity", 1), ("depth", 1), ("normal", 3), ("normal_view", 3),
                ("pos", 3), ("albedo", 3), ("roughness", 1), ("metallic", 1)]}
        radii = np.zeros(P, np.int32)
        R = lib().orc_forward(
            self._h, C.c_int(P), C.c_int(int(sh_degree)), C.c_int(M), _f(a["bg"]), C.c_int(W), C.c_int(H),
            _f(a["means3D"]), _f(a["shs"]), _f(a["colors_precomp"]), _f(a["opacities"]), _f(a["normal"]),
            _f(a["albedo"]), _f(a["roughness"]), _f(a["metallic"]), _f(a["scales"]),
            C.c_float(scale_modifier), _f(a["rotations"]), _f(a["cov3D_precomp"]), _f(a["viewmatrix"]),
            _f(a["projmatrix"]), _f(a["campos"]), C.c_float(tanfovx), C.c_float(tanfovy),
            C.c_int(int(argmax_depth)), C.c_int(int(inference)), _f(out["color"]), _f(out["opacity"]),
            _f(out["depth"]), _f(out["normal"]), _f(out["normal_view"]), _f(out["pos"]), _f(out["albedo"]),
            _f(out["roughness"]), _f(out["metallic"]), radii.ctypes.data_as(c_int_p))
        out["radii"] = radii
        out["num_rendered"] = int(R)
        self._fw = dict(a=a, P=P, M=M, D=int(sh_degree), H=H, W=W, tanfovx=tanfovx, tanfovy=tanfovy,
                        scale_modifier=scale_modifier)
        return out

    def backward(self, *, grad_color, grad_opacity, grad_depth, grad_normal, grad_albedo,
                 grad_roughness, grad_metallic):
        """_C.rasterize_gaussians_backward (rasterize_points.cu:254-364) for the last forward."""
        s = self._fw
        a, P, M = s["a"], s["P"], s["M"]
        g = dict(color=_c32(grad_color), opacity=_c32(grad_opacity), depth=_c32(grad_depth),
                 normal=_c32(grad_normal), albedo=_c32(grad_albedo), roughness=_c32(grad_roughness),
                 metallic=_c32(grad_metallic))
        o = dict(means2D=np.zeros((P, 3), np.float32), colors=np.zeros((P, 3), np.float32),
                 opacity=np.zeros((P, 1), np.float32), normal=np.zeros((P, 3), np.float32),
                 albedo=np.zeros((P, 3), np.float32), roughness=np.zeros((P, 1), np.float32),
                 metallic=np.zeros((P, 1), np.float32), means3D=np.zeros((P, 3), np.float32),
                 cov3D=np.zeros((P, 6), np.float32), sh=np.zeros((P, M, 3), np.float32),
                 scales=np.zeros((P, 3), np.float32), rotations=np.zeros((P, 4), np.float32))
        if P == 0:
            return o
        lib().orc_backward(
            self._h, C.c_int(P), C.c_int(s["D"]), C.c_int(M), _f(a["bg"]), C.c_int(s["W"]), C.c_int(s["H"]),
            _f(a["means3D"]), _f(a["shs"]), _f(a["colors_precomp"]), _f(a["normal"]), _f(a["scales"]),
            _f(a["rotations"]), _f(a["cov3D_precomp"]), _f(a["viewmatrix"]), _f(a["projmatrix"]),
            _f(a["campos"]), C.c_float(s["scale_modifier"]), C.c_float(s["tanfovx"]), C.c_float(s["tanfovy"]),
            _f(g["depth"]), _f(g["color"]), _f(g["opacity"]), _f(g["normal"]), _f(g["albedo"]),
            _f(g["roughness"]), _f(g["metallic"]), _f(o["means2D"]), _f(o["opacity"]), _f(o["normal"]),
            _f(o["albedo"]), _f(o["roughness"]), _f(o["metallic"]), _f(o["colors"]), _f(o["means3D"]),
            _f(o["cov3D"]), _f(o["sh"]) if M else None, _f(o["scales"]), _f(o["rotations"]))
        return o


def mark_visible(means3D, viewmatrix):
    means3D, viewmatrix = _c32(means3D), _c32(viewmatrix)
    P = means3D.shape[0]
    out = np.zeros(P, np.uint8)
    if P:
        lib().orc_mark_visible(C.c_int(P), _f(means3D), _f(viewmatrix), out.ctypes.data_as(C.c_void_p))
    return out.astype(bool)


def depth_to_normal(W, H, fx, fy, viewmatrix, depth):
    """_C.depth_to_normal (rasterize_points.cu:387-405) -> (normal[3,H,W], pos[3,H,W])."""
    depth, viewmatrix = _c32(depth), _c32(viewmatrix)
    normal = np.zeros((3, H, W), np.float32)
    pos = np.zeros((3, H, W), np.float32)
    lib().orc_depth_to_normal(C.c_int(W), C.c_int(H), C.c_float(fx), C.c_float(fy), _f(viewmatrix),
                              _f(depth), _f(normal), _f(pos))
    return normal, pos


def ssao(W, H, fx, fy, radius, bias, thick, delta, step, start, normal, pos):
    """_C.SSAO (rasterize_points.cu:407-436) -> occlusion[1,H,W]."""
    normal, pos = _c32(normal), _c32(pos)
    occ = np.ones((1, H, W), np.float32)
    lib().orc_ssao(C.c_int(W), C.c_int(H), C.c_float(fx), C.c_float(fy), C.c_float(radius),
                   C.c_float(bias), C.c_float(thick), C.c_float(delta), C.c_int(step), C.c_int(start),
                   _f(normal), _f(pos), _f(occ))
    return occ


def ssr(W, H, fx, fy, radius, bias, thick, delta, step, start, normal, pos, rgb, albedo, roughness,
        metallic, F0):
    """_C.SSR (rasterize_points.cu:438-477) -> (color[3,H,W], abd[3,H,W])."""
    arrs = [_c32(x) for x in (normal, pos, rgb, albedo, roughness, metallic, F0)]
    color = np.zeros((3, H, W), np.float32)
    abd = np.zeros((3, H, W), np.float32)
    lib().orc_ssr(C.c_int(W), C.c_int(H), C.c_float(fx), C.c_float(fy), C.c_float(radius),
                  C.c_float(bias), C.c_float(thick), C.c_float(delta), C.c_int(step), C.c_int(start),
                  *[_f(x) for x in arrs], _f(color), _f(abd))
    return color, abd


def gi_ray_counts(delta):
    a, b = C.c_int(0), C.c_int(0)
    lib().orc_gi_ray_counts(C.c_float(delta), C.byref(a), C.byref(b))
    return a.value, b.value


def median3x3(x):
    x = _c32(x)
    Cc, H, W = x.shape
    out = np.empty_like(x)
    lib().orc_median3x3(C.c_int(Cc), C.c_int(H), C.c_int(W), _f(x), _f(out))
    return out


def bilateral3x3(x, sigma_color=1.0, sigma_space=(3.0, 3.0)):
    x = _c32(x)
    Cc, H, W = x.shape
    out = np.empty_like(x)
    lib().orc_bilateral3x3(C.c_int(Cc), C.c_int(H), C.c_int(W), C.c_float(sigma_color),
                           C.c_float(sigma_space[1]), C.c_float(sigma_space[0]), _f(x), _f(out))
    return out


def higher_msb(n: int) -> int:
    return int(lib().orc_higher_msb(C.c_uint32(n)))


# ---- deferred shade / cubemap light (oracle/pbr_oracle.cpp) ------------------------------------
def diffuse_cubemap_fwd(cubemap):
    c = _c32(cubemap)
    out = np.empty_like(c)
    lib().orc_diffuse_cubemap_fwd(C.c_int(c.shape[1]), _f(c), _f(out))
    return out


def diffuse_cubemap_bwd(grad_out):
    g = _c32(grad_out)
    out = np.zeros_like(g)
    lib().orc_diffuse_cubemap_bwd(C.c_int(g.shape[1]), _f(g), _f(out))
    return out


def specular_bounds(res, cos_cutoff):
    b = np.zeros((6, res, res, 24), np.float32)
    lib().orc_specular_bounds(C.c_int(res), C.c_float(cos_cutoff), _f(b))
    return b


def specular_cubemap_fwd(cubemap, bounds, roughness, cos_cutoff):
    c, b = _c32(cubemap), _c32(bounds)
    out = np.zeros((6, c.shape[1], c.shape[1], 4), np.float32)
    lib().orc_specular_cubemap_fwd(C.c_int(c.shape[1]), _f(c), _f(b), C.c_float(roughness), C.c_float(cos_cutoff), _f(out))
    return out


def specular_cubemap_bwd(bounds, grad_out, roughness, cos_cutoff):
    b, g = _c32(bounds), _c32(grad_out)
    res = g.shape[1]
    out = np.zeros((6, res, res, 3), np.float32)
    lib().orc_specular_cubemap_bwd(C.c_int(res), _f(b), _f(g), C.c_float(roughness), C.c_float(cos_cutoff), _f(out))
    return out


def cubemap_mip_fwd(cubemap):
    c = _c32(cubemap)
    r, ch = c.shape[1] // 2, c.shape[3]
    out = np.zeros((6, r, r, ch), np.float32)
    lib().orc_cubemap_mip_fwd(C.c_int(r), C.c_int(ch), _f(c), _f(out))
    return out


def cubemap_mip_bwd(dout):
    d = _c32(dout)
    r = d.shape[1]
    out = np.zeros((6, 2 * r, 2 * r, 3), np.float32)
    lib().orc_cubemap_mip_bwd(C.c_int(r), _f(d), _f(out))
    return out


def shade_fwd(normals, view_dirs, albedo, roughness, mask, occlusion, metallic, background, diffuse, specular, lut,
              tone=False, gamma=False):
    """pbr_shading forward on HWC numpy arrays -> dict(render_rgb, diffuse_rgb, specular_rgb, diffuse_light)."""
    H, W, _ = normals.shape
    arrs = dict(normals=_c32(normals), view_dirs=_c32(view_dirs), albedo=_c32(albedo), roughness=_c32(roughness),
                occlusion=None if occlusion is None else _c32(occlusion),
                metallic=None if metallic is None else _c32(metallic),
                background=None if background is None else _c32(background), diffuse=_c32(diffuse), lut=_c32(lut))
    m8 = np.ascontiguousarray(mask, dtype=np.uint8)
    spec = [_c32(s) for s in specular]
    sp = (c_float_p * len(spec))(*[_f(s) for s in spec])
    sr = (C.c_int * len(spec))(*[s.shape[1] for s in spec])
    outs = [np.zeros((H, W, 3), np.float32) for _ in range(4)]
    lut2 = arrs["lut"].reshape(-1, arrs["lut"].shape[-2], 2)
    lib().orc_shade_fwd(C.c_int(H), C.c_int(W), _f(arrs["normals"]), _f(arrs["view_dirs"]), _f(arrs["albedo"]),
                        _f(arrs["roughness"]), m8.ctypes.data_as(C.c_void_p), _f(arrs["occlusion"]), _f(arrs["metallic"]),
                        _f(arrs["background"]), _f(arrs["diffuse"]), C.c_int(arrs["diffuse"].shape[1]), C.c_int(len(spec)),
                        sp, sr, _f(arrs["lut"]), C.c_int(lut2.shape[1]), C.c_int(lut2.shape[0]), C.c_int(int(tone)),
                        C.c_int(int(gamma)), *[_f(o) for o in outs])
    return dict(render_rgb=outs[0], diffuse_rgb=outs[1], specular_rgb=outs[2], diffuse_light=outs[3])
