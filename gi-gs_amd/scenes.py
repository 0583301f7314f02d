"""Synthetic scenes and cameras for tests and bench.py (numpy only, no GPU).

The inputs follow the reference's conventions:
  * per-Gaussian inputs are POST-activation (scene/gaussian_model.py:178-263): sigmoid
    opacity / albedo / roughness / metallic, exp scales, L2-normalised rotations and normals;
  * `viewmatrix = W2C.T`, `projmatrix = viewmatrix @ P.T`, `campos = inverse(viewmatrix)[3,:3]`
    (scene/cameras.py:75-86) with P from utils/graphics_utils.py:62-82 (znear 0.01, zfar 100).
`projection_matrix` / `look_at_camera` are restatements pinned against the reference's
importable `getProjectionMatrix` / `getWorld2View2` by tests/golden/camera_*.npz.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

C0 = 0.28209479177387814


def rgb2sh(rgb):  # utils/sh_utils.py RGB2SH
    return (rgb - 0.5) / C0


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def projection_matrix(znear: float, zfar: float, fovX: float, fovY: float) -> np.ndarray:
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = np.zeros((4, 4), dtype=np.float32)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def look_at_camera(eye, target, width: int, height: int, fovx: float, up=(0.0, 0.0, 1.0)) -> Dict:
    """Camera looking from `eye` to `target`; view space is x right, y down, z forward."""
    eye = np.asarray(eye, np.float64)
    target = np.asarray(target, np.float64)
    f = target - eye
    f /= np.linalg.norm(f)
    upv = np.asarray(up, np.float64)
    r = np.cross(f, upv)
    if np.linalg.norm(r) < 1e-8:
        r = np.cross(f, np.array([0.0, 1.0, 0.0]))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)  # "down"
    R_w2c = np.stack([r, d, f], axis=0)  # rows
    W2C = np.eye(4, dtype=np.float64)
    W2C[:3, :3] = R_w2c
    W2C[:3, 3] = -R_w2c @ eye
    W2C = W2C.astype(np.float32)
    fovy = 2.0 * math.atan(math.tan(fovx / 2.0) * height / width)
    viewmatrix = np.ascontiguousarray(W2C.T)
    P = projection_matrix(0.01, 100.0, fovx, fovy)
    projmatrix = np.ascontiguousarray((viewmatrix @ P.T).astype(np.float32))
    campos = np.ascontiguousarray(np.linalg.inv(viewmatrix)[3, :3].astype(np.float32))
    return dict(viewmatrix=viewmatrix, projmatrix=projmatrix, campos=campos,
                tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5),
                image_width=int(width), image_height=int(height), fovx=fovx, fovy=fovy)


def orbit_camera(index: int, n_views: int, width: int, height: int, radius: float = 4.0,
                 fovx: float = 0.6911, elevation: float = 0.5) -> Dict:
    """View `index` of `n_views` on a radius-`radius` orbit looking at the origin (C1/C2)."""
    az = 2.0 * math.pi * (index + 0.25) / max(1, n_views)
    eye = (radius * math.cos(elevation) * math.cos(az), radius * math.cos(elevation) * math.sin(az),
           radius * math.sin(elevation))
    return look_at_camera(eye, (0.0, 0.0, 0.0), width, height, fovx)


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _common(rng, P, sh_degree_max, means, log_scale_mu, log_scale_sigma, normals=None):
    M = (sh_degree_max + 1) ** 2
    scales = np.exp(rng.normal(math.log(log_scale_mu), log_scale_sigma, size=(P, 3)))
    rotations = _unit(rng.normal(size=(P, 4)))
    opacities = sigmoid(rng.normal(size=(P, 1)))
    shs = np.zeros((P, M, 3))
    shs[:, 0, :] = rgb2sh(rng.uniform(0.0, 1.0, size=(P, 3)))
    if M > 1:
        shs[:, 1:, :] = rng.normal(0.0, 0.1, size=(P, M - 1, 3))
    if normals is None:
        normals = _unit(rng.normal(size=(P, 3)))
    albedo = sigmoid(rng.normal(size=(P, 3)))
    roughness = sigmoid(rng.normal(size=(P, 1)))
    metallic = sigmoid(rng.normal(size=(P, 1)))
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    return dict(means3D=f32(means), scales=f32(scales), rotations=f32(rotations), opacities=f32(opacities),
                shs=f32(shs), normal=f32(normals), albedo=f32(albedo), roughness=f32(roughness),
                metallic=f32(metallic), sh_degree=int(sh_degree_max))


def random_scene(P: int = 10_000, sh_degree: int = 0, seed: int = 0, extent: float = 1.3,
                 scale_mu: float = 0.03, scale_sigma: float = 0.5) -> Dict:
    """BASELINE config C1: uniform cloud in [-extent, extent]^3."""
    rng = np.random.default_rng(seed)
    means = rng.uniform(-extent, extent, size=(P, 3))
    return _common(rng, P, sh_degree, means, scale_mu, scale_sigma)


def surface_scene(P: int = 300_000, sh_degree: int = 2, seed: int = 0, scale_mu: float = 0.01,
                  scale_sigma: float = 0.4) -> Dict:
    """BASELINE config C2 stand-in: noisy sphere shells + a ground plane, so that the
    screen-space passes actually hit geometry; opacities are biased high as in a trained scene."""
    rng = np.random.default_rng(seed)
    n_plane = P // 4
    n_sph = P - n_plane
    centers = np.array([[0.0, 0.0, 0.0], [0.9, 0.5, -0.3], [-0.8, -0.6, -0.35], [0.1, -1.0, -0.45]])
    radii = np.array([0.6, 0.35, 0.3, 0.2])
    which = rng.choice(len(centers), size=n_sph, p=radii ** 2 / np.sum(radii ** 2))
    d = _unit(rng.normal(size=(n_sph, 3)))
    r = radii[which][:, None] * (1.0 + rng.normal(0.0, 0.01, size=(n_sph, 1)))
    sph = centers[which] + d * r
    plane = np.stack([rng.uniform(-1.6, 1.6, n_plane), rng.uniform(-1.6, 1.6, n_plane),
                      -0.65 + rng.normal(0.0, 0.004, n_plane)], axis=1)
    means = np.concatenate([sph, plane], axis=0)
    normals = np.concatenate([d, np.tile(np.array([[0.0, 0.0, 1.0]]), (n_plane, 1))], axis=0)
    normals = _unit(normals + rng.normal(0.0, 0.05, size=normals.shape))
    perm = rng.permutation(P)  # storage order carries no spatial coherence, as after densification
    s = _common(rng, P, sh_degree, means[perm], scale_mu, scale_sigma, normals=normals[perm])
    s["opacities"] = np.ascontiguousarray(sigmoid(rng.normal(1.5, 1.0, size=(P, 1))), dtype=np.float32)
    return s


GI_DEFAULTS = dict(radius=0.8, bias=0.01, thick=0.05, delta=0.0625, step=16, start=8)  # train.py:850-855


def synthetic_envmap(height: int = 512, width: int = 1024, seed: int = 1) -> np.ndarray:
    """HDR-like equirectangular environment map [H, W, 3] fp32 (stand-in for the TensoIR `bridge.hdr` of BASELINE
    config C3, which is not available offline): a sky/ground gradient, a few soft area lights and one small, very
    bright 'sun' so that the GGX pre-filter sees a high dynamic range."""
    rng = np.random.default_rng(seed)
    v = (np.arange(height) + 0.5) / height
    u = (np.arange(width) + 0.5) / width
    uu, vv = np.meshgrid(u, v)
    sky = np.stack([0.35 + 0.3 * (1 - vv), 0.45 + 0.35 * (1 - vv), 0.6 + 0.5 * (1 - vv)], -1)
    ground = np.stack([0.25 + 0 * vv, 0.22 + 0 * vv, 0.18 + 0 * vv], -1)
    img = np.where((vv < 0.5)[..., None], sky, ground) * (0.8 + 0.2 * np.cos(2 * np.pi * uu)[..., None])
    for _ in range(6):
        cu, cv, s = rng.uniform(0, 1), rng.uniform(0.1, 0.6), rng.uniform(0.02, 0.08)
        du = np.minimum(np.abs(uu - cu), 1 - np.abs(uu - cu))
        blob = np.exp(-0.5 * ((du / s) ** 2 + ((vv - cv) / s) ** 2))
        img += blob[..., None] * rng.uniform(1.0, 6.0, size=3)
    cu, cv = 0.3, 0.25
    du = np.minimum(np.abs(uu - cu), 1 - np.abs(uu - cu))
    img += np.exp(-0.5 * ((du / 0.006) ** 2 + ((vv - cv) / 0.006) ** 2))[..., None] * np.array([60.0, 55.0, 45.0])
    return np.ascontiguousarray(img, dtype=np.float32)
