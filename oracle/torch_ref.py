"""Independent PyTorch (CPU, float64) restatement of the rasterizer FORWARD, used only to
check the hand-written BACKWARD of the oracle (and through it the HIP backward) by autograd.

TEST INFRASTRUCTURE ONLY (see oracle/gigs_oracle.cpp header).

It is written from the maths, not from the kernels' instruction order: projection, EWA
covariance J W Sigma W^T J^T + 0.3 I, conic = inverse, SH colour, front-to-back alpha
blending.  The discrete structure (which Gaussians a tile sees, and in which order) is taken
from the oracle's `point_list` / `ranges`, because binning is integer work that has its own
bit-exact tests.

The reference's backward is NOT the exact gradient of its forward; the restatement encodes
the documented deviations so that autograd reproduces what the reference computes
(R/cuda_rasterizer/backward.cu):
  * normal / albedo / roughness / metallic / depth planes do not feed dL_dalpha (:580-590):
    their blend weights are detached;
  * the depth gradient ignores the forward's division by the accumulated opacity (:590 vs
    forward.cu:619): the surrogate plane is sum_i w_i * z_i;
  * dL_dnormal is zeroed on the image border (:497-501);
  * when the projected centre is clamped to 1.3 * tan(fov) the clamped coordinate is treated
    as independent of t.z and dL_dt{x,y} is zeroed (:177-178, :264-266);
  * SH colours clamped at 0 pass no gradient (:30-35); the quaternion is used un-normalised.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
      -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def _sh_color(deg, sh, dirs):
    """sh: [P, M, 3]; dirs: [P, 3] unit. Real SH basis as in forward.cu:22-80."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
               + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res + 0.5


def _quat_R(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def render_with_grads(inp: Dict[str, np.ndarray], cam: Dict, bg: np.ndarray, point_list: np.ndarray,
                      ranges: np.ndarray, visible: np.ndarray, pix_grads: Dict[str, np.ndarray],
                      scale_modifier: float = 1.0) -> Dict[str, np.ndarray]:
    """Returns the forward planes and d(sum_k <plane_k, pix_grads[k]>)/d(input) for every input."""
    dt = torch.float64
    W, H = cam["image_width"], cam["image_height"]
    t = {k: torch.tensor(np.asarray(v, np.float64), dtype=dt, requires_grad=True)
         for k, v in inp.items() if k in ("means3D", "scales", "rotations", "opacities", "shs", "normal",
                                          "albedo", "roughness", "metallic")}
    P = t["means3D"].shape[0]
    deg = int(inp["sh_degree"])
    vm = torch.tensor(np.asarray(cam["viewmatrix"], np.float64))
    pm = torch.tensor(np.asarray(cam["projmatrix"], np.float64))
    campos = torch.tensor(np.asarray(cam["campos"], np.float64))
    tanx, tany = cam["tanfovx"], cam["tanfovy"]
    fx, fy = W / (2.0 * tanx), H / (2.0 * tany)
    ndc_off = torch.zeros(P, 2, dtype=dt, requires_grad=True)  # receives "means2D.grad"

    hom = torch.cat([t["means3D"], torch.ones(P, 1, dtype=dt)], dim=1)
    pv = hom @ vm  # [P,4] view space (row-vector convention)
    ph = hom @ pm
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None] + ndc_off
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)

    # 3D covariance
    R = _quat_R(t["rotations"])
    S = torch.diag_embed(scale_modifier * t["scales"])
    Mm = R @ S
    Sigma = Mm @ Mm.transpose(1, 2)

    # EWA projection with the reference's clamp semantics
    tz = pv[:, 2]
    limx, limy = 1.3 * tanx, 1.3 * tany
    rx, ry = pv[:, 0] / tz, pv[:, 1] / tz
    cl_x = (rx < -limx) | (rx > limx)
    cl_y = (ry < -limy) | (ry > limy)
    tx = torch.where(cl_x, (rx.clamp(-limx, limx) * tz).detach(), pv[:, 0])
    ty = torch.where(cl_y, (ry.clamp(-limy, limy) * tz).detach(), pv[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz),
                     zero, fy / tz, -(fy * ty) / (tz * tz)], dim=-1).reshape(P, 2, 3)
    Rw2c = vm[:3, :3].T  # column-vector rotation
    T = J @ Rw2c
    cov = T @ Sigma @ T.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], dim=1)

    # colour
    dirs = t["means3D"] - campos[None]
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    rgb = torch.clamp_min(_sh_color(deg, t["shs"], dirs), 0.0)

    opac = t["opacities"][:, 0]
    zview = pv[:, 2]

    gx = (W + 15) // 16
    gy = (H + 15) // 16
    planes = {k: torch.zeros(n, H, W, dtype=dt) for k, n in
              [("color", 3), ("opacity", 1), ("depth_sur", 1), ("normal", 3), ("albedo", 3),
               ("roughness", 1), ("metallic", 1)]}
    n_contrib = np.zeros((H, W), np.int64)
    bgt = torch.tensor(np.asarray(bg, np.float64))
    for tile in range(gx * gy):
        ty_, tx_ = divmod(tile, gx)
        x0, y0 = tx_ * 16, ty_ * 16
        x1, y1 = min(x0 + 16, W), min(y0 + 16, H)
        ys, xs = torch.meshgrid(torch.arange(y0, y1, dtype=dt), torch.arange(x0, x1, dtype=dt), indexing="ij")
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        n = ys.numel()
        Tt = torch.ones(n, dtype=dt)
        done = torch.zeros(n, dtype=torch.bool)
        acc = {k: [torch.zeros(n, dtype=dt) for _ in range(v.shape[0])] for k, v in planes.items()}
        last = np.zeros(n, np.int64)
        lo, hi = int(ranges[2 * tile]), int(ranges[2 * tile + 1])
        for pos, k in enumerate(range(lo, hi)):
            g = int(point_list[k])
            dx = pix[g, 0] - xs
            dy = pix[g, 1] - ys
            power = -0.5 * (conic[g, 0] * dx * dx + conic[g, 2] * dy * dy) - conic[g, 1] * dx * dy
            alpha = torch.clamp_max(opac[g] * torch.exp(power), 0.99)
            ok = (~done) & (power <= 0) & (alpha >= 1.0 / 255.0)
            test_T = Tt * (1 - alpha)
            newly_done = ok & (test_T < 0.0001)
            done = done | newly_done
            ok = ok & ~newly_done
            if not bool(ok.any()):
                continue
            w = torch.where(ok, alpha * Tt, torch.zeros_like(Tt))
            wd = w.detach()
            for ch in range(3):
                acc["color"][ch] = acc["color"][ch] + w * rgb[g, ch]
                acc["normal"][ch] = acc["normal"][ch] + wd * t["normal"][g, ch]
                acc["albedo"][ch] = acc["albedo"][ch] + wd * t["albedo"][g, ch]
            acc["opacity"][0] = acc["opacity"][0] + w
            acc["roughness"][0] = acc["roughness"][0] + wd * t["roughness"][g, 0]
            acc["metallic"][0] = acc["metallic"][0] + wd * t["metallic"][g, 0]
            acc["depth_sur"][0] = acc["depth_sur"][0] + wd * zview[g]
            Tt = torch.where(ok, test_T, Tt)
            last[ok.numpy()] = pos + 1
        for ch in range(3):
            acc["color"][ch] = acc["color"][ch] + Tt * bgt[ch]
        hh, ww = y1 - y0, x1 - x0
        for k, v in planes.items():
            for ch in range(v.shape[0]):
                v[ch, y0:y1, x0:x1] = acc[k][ch].reshape(hh, ww)
        n_contrib[y0:y1, x0:x1] = last.reshape(hh, ww)

    g = {k: torch.tensor(np.asarray(v, np.float64)) for k, v in pix_grads.items()}
    gn = g["normal"].clone()
    gn[:, 0, :] = 0
    gn[:, -1, :] = 0
    gn[:, :, 0] = 0
    gn[:, :, -1] = 0
    loss = ((planes["color"] * g["color"]).sum() + (planes["opacity"] * g["opacity"]).sum()
            + (planes["depth_sur"] * g["depth"]).sum() + (planes["normal"] * gn).sum()
            + (planes["albedo"] * g["albedo"]).sum() + (planes["roughness"] * g["roughness"]).sum()
            + (planes["metallic"] * g["metallic"]).sum())
    loss.backward()
    out = {("d_" + k): (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in t.items()}
    out["d_means2D"] = ndc_off.grad.numpy() if ndc_off.grad is not None else np.zeros((P, 2))
    for k, v in planes.items():
        out[k] = v.detach().numpy()
    out["n_contrib"] = n_contrib
    out["clamped_any"] = bool(((cl_x | cl_y) & torch.tensor(visible)).any())
    return out
