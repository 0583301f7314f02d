"""PyTorch (CPU, float64) restatement of the deferred shade and of the cubemap light filters,
with autograd: the checker for the hand-written HIP backward kernels.

TEST INFRASTRUCTURE ONLY.  Texture sampling follows the rule written in include/gigs_hip.h
(modelled on nvdiffrast's dr.texture; third party, parity unpinned).
"""
from __future__ import annotations

import math

import numpy as np
import torch

DT = torch.float64


def cube_dir_raw(a, b, face):
    one = torch.ones_like(a)
    outs = [torch.stack(v, -1) for v in ((one, -b, -a), (-one, -b, a), (a, one, b), (a, -one, -b), (a, -b, one), (-a, -b, -one))]
    res = torch.zeros_like(outs[0])
    for f in range(6):
        res = torch.where((face == f)[..., None], outs[f], res)
    return res


def cube_face_uv(d):
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    ax, ay, az = x.abs(), y.abs(), z.abs()
    is_z = az > torch.maximum(ax, ay)
    is_y = (~is_z) & (ay > ax)
    c = torch.where(is_z, z, torch.where(is_y, y, x))
    a = torch.where(is_z, x, torch.where(is_y, x, z))
    b = torch.where(is_z, y, torch.where(is_y, z, y))
    face = torch.where(is_z, 4, torch.where(is_y, 2, 0)) + (c < 0).long()
    m = 0.5 / c.abs()
    m0 = torch.where((face == 0) | (face == 5), -m, m)
    m1 = torch.where(face != 2, -m, m)
    u = (a * m0 + 0.5).clamp(0, 1)
    v = (b * m1 + 0.5).clamp(0, 1)
    return face, u, v


def cube_taps(res, d):
    """-> idx [...,4] (long, -1 = dropped), w [...,4]"""
    face, u, v = cube_face_uv(d)
    fu, fv = u * res - 0.5, v * res - 0.5
    iu0, iv0 = torch.floor(fu).detach(), torch.floor(fv).detach()
    tu, tv = fu - iu0, fv - iv0
    idxs, ws = [], []
    for k in range(4):
        ox, oy = k & 1, k >> 1
        ix, iy = iu0.long() + ox, iv0.long() + oy
        w = (tu if ox else 1 - tu) * (tv if oy else 1 - tv)
        out_x = (ix < 0) | (ix >= res)
        out_y = (iy < 0) | (iy >= res)
        inside = (face * res + iy.clamp(0, res - 1)) * res + ix.clamp(0, res - 1)
        a = 2.0 * ((ix.to(DT) + 0.5) / res) - 1.0
        b = 2.0 * ((iy.to(DT) + 0.5) / res) - 1.0
        f2, u2, v2 = cube_face_uv(cube_dir_raw(a, b, face))
        x2 = torch.floor(u2 * res).long().clamp(0, res - 1)
        y2 = torch.floor(v2 * res).long().clamp(0, res - 1)
        wrapped = (f2 * res + y2) * res + x2
        idx = torch.where(out_x & out_y, torch.full_like(inside, -1), torch.where(out_x | out_y, wrapped, inside))
        idxs.append(idx)
        ws.append(w)
    idx = torch.stack(idxs, -1)
    w = torch.stack(ws, -1)
    valid = idx >= 0
    w = torch.where(valid, w, torch.zeros_like(w))
    dropped = (~valid).any(-1, keepdim=True)
    w = torch.where(dropped, w / w.sum(-1, keepdim=True), w)
    return idx, w


def cube_sample(tex, d):
    """tex [6,r,r,3]; d [...,3] -> [...,3]"""
    res = tex.shape[1]
    idx, w = cube_taps(res, d)
    flat = tex.reshape(-1, tex.shape[-1])
    vals = flat[idx.clamp(min=0)]  # [...,4,3]
    return (vals * w[..., None]).sum(-2)


def lut_sample(lut, u, v):
    """lut [h,w,2], linear + clamp"""
    h, w_ = lut.shape[0], lut.shape[1]
    fu, fv = u * w_ - 0.5, v * h - 0.5
    iu0, iv0 = torch.floor(fu).detach(), torch.floor(fv).detach()
    tu, tv = (fu - iu0)[..., None], (fv - iv0)[..., None]
    x0, x1 = iu0.long().clamp(0, w_ - 1), (iu0.long() + 1).clamp(0, w_ - 1)
    y0, y1 = iv0.long().clamp(0, h - 1), (iv0.long() + 1).clamp(0, h - 1)
    return (lut[y0, x0] * (1 - tu) * (1 - tv) + lut[y0, x1] * tu * (1 - tv) + lut[y1, x0] * (1 - tu) * tv
            + lut[y1, x1] * tu * tv)


def get_mip(r, L):
    MINR, MAXR = 0.08, 0.5
    return torch.where(r < MAXR, (r.clamp(MINR, MAXR) - MINR) / (MAXR - MINR) * (L - 2),
                       (r.clamp(MAXR, 1.0) - MAXR) / (1.0 - MAXR) + L - 2)


def linear_to_srgb(x):
    eps = torch.finfo(torch.float32).eps
    return torch.where(x <= 0.0031308, 323 / 25 * x, (211 * x.clamp(min=eps) ** (5 / 12) - 11) / 200)


def aces(x):
    a, b, c, d, e = 2.51, 0.03, 2.43, 0.59, 0.14
    return (x * (a * x + b)) / (x * (c * x + d) + e)


def shade(normals, view_dirs, albedo, roughness, mask, occlusion, metallic, background, diffuse, specular, lut,
          tone=False, gamma=False):
    """pbr/shade.py:108-241 on [H,W,*] float64 tensors."""
    n, v = normals, view_dirs
    ref = 2.0 * (n * v).sum(-1, keepdim=True).clamp(min=0.0) * n - v
    T = torch.tensor([[0, -1, 0], [0, 0, 1], [-1, 0, 0]], dtype=DT)
    nt, vt, rt = n @ T.T, v @ T.T, ref @ T.T
    dl = cube_sample(diffuse, nt)
    if occlusion is not None:
        dl = dl * occlusion
    drgb = dl * albedo
    nov = (nt * vt).sum(-1).clamp(1e-4, 1.0)
    fg = lut_sample(lut, nov, roughness[..., 0])
    L = len(specular)
    lvl = get_mip(roughness[..., 0], L).clamp(0, L - 1)
    l0 = torch.floor(lvl).detach().long().clamp(max=L - 1)
    l1 = (l0 + 1).clamp(max=L - 1)
    lf = (lvl - l0)[..., None]
    samples = torch.stack([cube_sample(s, rt) for s in specular], 0)  # [L,H,W,3]
    s0 = torch.gather(samples, 0, l0[None, ..., None].expand(1, *l0.shape, 3))[0]
    s1 = torch.gather(samples, 0, l1[None, ..., None].expand(1, *l1.shape, 3))[0]
    spec = torch.where((l1 != l0)[..., None], s0 * (1 - lf) + s1 * lf, s0)
    F0 = torch.full_like(albedo, 0.04) if metallic is None else (1.0 - metallic) * 0.04 + albedo * metallic
    refl = F0 * fg[..., 0:1] + fg[..., 1:2]
    srgb = spec * refl
    render = drgb + srgb
    render = aces(render).clamp(0, 1) if tone else render.clamp(0, 1)
    if gamma:
        render, drgb, srgb = linear_to_srgb(render), linear_to_srgb(drgb), linear_to_srgb(srgb)
    bg = torch.zeros_like(render) if background is None else background
    render = torch.where(mask, render, bg)
    return render, drgb, srgb, dl


# ---- cubemap filters (dense float64 restatement of RU/cubemap.cu) ---------------------------
def texel_dirs(N):
    s, y, x = torch.meshgrid(torch.arange(6), torch.arange(N), torch.arange(N), indexing="ij")
    fx = 2.0 * ((x.to(DT) + 0.5) / N) - 1.0
    fy = 2.0 * ((y.to(DT) + 0.5) / N) - 1.0
    d = cube_dir_raw(fx, fy, s)
    return (d / d.norm(dim=-1, keepdim=True)).reshape(-1, 3)


def pixel_areas(N):
    H = N // 2
    i = (torch.arange(N) - H).abs().to(DT)
    a = torch.atan((i + 1) / H) - torch.atan(i / H)
    return (a[:, None] * a[None, :])[None].expand(6, N, N).reshape(-1)


def diffuse_cubemap(cubemap):
    N = cubemap.shape[1]
    d = texel_dirs(N)
    w = (d @ d.T).clamp(0.0, 0.999) * pixel_areas(N)[None, :] / 3.141592
    return (w @ cubemap.reshape(-1, 3)).reshape(6, N, N, 3)


def bounds_mask(bounds):
    """[out, in] membership of texel `in` in the per-face AABB stored for texel `out`
    (RU/cubemap.cu:181-244; the AABBs are discrete structure taken from the C oracle)."""
    N = bounds.shape[1]
    b = torch.as_tensor(np.asarray(bounds)).reshape(6 * N * N, 6, 4).long()
    s, y, x = torch.meshgrid(torch.arange(6), torch.arange(N), torch.arange(N), indexing="ij")
    s, y, x = s.reshape(-1), y.reshape(-1), x.reshape(-1)
    bb = b[:, s, :]  # [out, in, 4]
    return (x[None] >= bb[..., 0]) & (x[None] <= bb[..., 1]) & (y[None] >= bb[..., 2]) & (y[None] <= bb[..., 3])


def specular_cubemap_rgbw(cubemap, roughness, cos_cutoff, bounds=None):
    N = cubemap.shape[1]
    d = texel_dirs(N)
    dp = d @ d.T  # [out, in]
    if bounds is not None:
        dp = torch.where(bounds_mask(bounds), dp, torch.full_like(dp, -2.0))
    alphaSqr = roughness ** 4
    Hh = d[:, None, :] + d[None, :, :]
    Hh = Hh / Hh.norm(dim=-1, keepdim=True).clamp(min=1e-30)
    vdh = (d[:, None, :] * Hh).sum(-1).clamp(0, 1)
    dd = (vdh * alphaSqr - vdh) * vdh + 1.0
    ndf = alphaSqr / (dd * dd * math.pi)
    w = torch.where(dp >= cos_cutoff, dp.clamp(min=0) * ndf * pixel_areas(N)[None, :] / 4.0, torch.zeros_like(dp))
    rgb = w @ cubemap.reshape(-1, 3)
    return torch.cat([rgb, w.sum(-1, keepdim=True)], -1).reshape(6, N, N, 4)


def cubemap_mip_bwd(dout):
    """pbr/light.py:62-79 with the sampling rule above. dout [6,r,r,3] -> [6,2r,2r,3]"""
    r = dout.shape[1]
    res = 2 * r
    out = torch.zeros(6, res, res, 3, dtype=DT)
    lin = torch.linspace(-1.0 + 1.0 / res, 1.0 - 1.0 / res, res, dtype=DT)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    for s in range(6):
        d = cube_dir_raw(gx, gy, torch.full_like(gx, s).long())
        d = d / d.norm(dim=-1, keepdim=True)
        out[s] = cube_sample(dout * 0.25, d)
    return out


def to64(a):
    return torch.tensor(np.asarray(a, np.float64), dtype=DT)
