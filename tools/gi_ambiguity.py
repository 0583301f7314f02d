"""Estimates, on the bench view, which fraction of SSAO rays a conservative screen-space pre-test (3x3 min/max depth
window around the approximately projected sample, slab [z - thick, z + bias]) could certify as hit-free."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gi-gs_amd"))
import numpy as np, torch
import torch.nn.functional as F
import scenes, pipeline
dev = "cuda:0"
sc = scenes.surface_scene(P=300_000, sh_degree=2, seed=0)
cam = scenes.orbit_camera(0, 64, 800, 800, radius=3.5)
gi = scenes.GI_DEFAULTS
g = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if hasattr(v, "dtype")}
camt = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
out, _, st = pipeline.rasterize(camt, g, 2, torch.zeros(3, device=dev), gi)
nrm, pos = out[10], out[11]          # raw view normal (NaN where empty), filtered position
H = W = 800
fx, fy = W / (2 * cam["tanfovx"]), H / (2 * cam["tanfovy"])
z = pos[2]
zpad = F.pad(z[None, None], (1, 1, 1, 1), mode="replicate")
zmin = -F.max_pool2d(-zpad, 3, 1)[0, 0]
zmax = F.max_pool2d(zpad, 3, 1)[0, 0]
# ray table (forward.cu:664-690 loop structure, float32)
rays = []
d = np.float32(gi["delta"]) * np.float32(math.pi)
phi = np.float32(0)
while float(phi) < 2 * math.pi:
    th = np.float32(0)
    while float(th) <= 0.5 * math.pi:
        rays.append((math.sin(th) * math.cos(phi), math.sin(th) * math.sin(phi), math.cos(th)))
        th = np.float32(float(th) + float(d) * 0.5)
    phi = np.float32(phi + d)
ts = torch.tensor(rays, device=dev, dtype=torch.float32)
ts = ts / ts.norm(dim=1, keepdim=True)
ys, xs = torch.meshgrid(torch.arange(0, H, 4, device=dev), torch.arange(0, W, 4, device=dev), indexing="ij")
ys, xs = ys.flatten(), xs.flatten()
n = F.normalize(nrm[:, ys, xs].T, dim=1)
ok = torch.isfinite(n).all(1)
ys, xs, n = ys[ok], xs[ok], n[ok]
p = pos[:, ys, xs].T
up = torch.tensor([0.0, 1.0, 0.0], device=dev)
t = F.normalize(up[None] - n * (n @ up)[:, None], dim=1)
b = F.normalize(torch.cross(n, t, dim=1), dim=1)
a = 1 + p[:, 2] / 100
amb_rays = torch.zeros(len(ys), device=dev)
eps = 1e-4
for r in range(ts.shape[0]):
    sv = t * ts[r, 0] + b * ts[r, 1] + n * ts[r, 2]
    amb = torch.zeros(len(ys), dtype=torch.bool, device=dev)
    alive = torch.ones(len(ys), dtype=torch.bool, device=dev)
    for j in range(gi["start"], gi["step"]):
        s = p + sv * (j * a * a * gi["radius"] / gi["step"])[:, None]
        q = s[:, :2] / (s[:, 2:3] + 1e-7)
        ix = torch.round(q[:, 0] * fx + W / 2).long(); iy = torch.round(q[:, 1] * fy + H / 2).long()
        far_out = (ix < -1) | (ix > W) | (iy < -1) | (iy > H)
        border = ~far_out & ((ix < 1) | (ix > W - 2) | (iy < 1) | (iy > H - 2))
        cx, cy = ix.clamp(0, W - 1), iy.clamp(0, H - 1)
        lo, hi = s[:, 2] - gi["thick"] - eps, s[:, 2] + gi["bias"] + eps
        certain = far_out | (~border & ((zmin[cy, cx] > hi) | (zmax[cy, cx] < lo)))
        amb |= alive & ~certain
        alive &= ~far_out
    amb_rays += amb.float()
frac = amb_rays / ts.shape[0]
print("pixels tested", len(ys), "rays", ts.shape[0])
print("ambiguous ray fraction: mean %.3f median %.3f p90 %.3f max %.3f" % (frac.mean(), frac.median(), frac.quantile(0.9), frac.max()))
