#!/usr/bin/env python
"""Could a WAVE-level interval bound certify march samples?  (CPU experiment, oracle G-buffer; round 4.)

Today every lane projects every sample and looks its 16-pixel block up in the min/max table (march2_cert): 11.5 vector
instructions per certified wave-sample, 92 % of the wave-samples certify for all 64 lanes.  The idea measured here: the 64
pixels of a wave are an 8x8 block -- neighbouring positions, similar tangent frames -- so per (ray, step) ONE interval
computation (component-wise min / max of the per-pixel constants over the wave's live lanes, interval products with the
ray's table direction, an interval division) bounds where ALL 64 samples land and what their hit intervals are; if that
pixel rectangle covers at most 2x2 table blocks, all of them full in-image blocks, and the union of the hit intervals
clears the blocks' min / max, the sample is certified for the whole wave without any per-lane work.  Evaluated with the
lanes as 64 different RAYS (the bound is wave-uniform per ray), it would cost ~50 vector instructions per (64 rays, step)
instead of 64 x 11.5.

Prints, for 8x8 tiles of the C2 view that hold geometry: the share of (tile, ray, step) triples the wave bound certifies,
beside the share the per-lane test certifies for all 64 lanes, and the share of (tile, ray) pairs whose every step is
wave-certified."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("gi-gs_amd")
import numpy as np  # noqa: E402

import scenes  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import stage2_ref  # noqa: E402


def ray_table(delta):
    sd = np.float32(delta * np.pi)
    rays = []
    phi = np.float32(0)
    while phi < 2 * np.pi:
        th = np.float32(0)
        while th <= 0.5 * np.pi:
            v = np.array([np.sin(th) * np.cos(phi), np.sin(th) * np.sin(phi), np.cos(th)], np.float64)
            rays.append(v / np.linalg.norm(v))
            th = np.float32(th + sd * 0.5)
        phi = np.float32(phi + sd)
    r = np.array(rays)
    return r[r[:, 2] < 1.0 - 1e-12]  # the live rays


def main():
    orc.build()
    orc.set_threads(orc.max_threads())
    W = H = int(os.environ.get("RES", 800))
    P = int(os.environ.get("P", 300_000))
    ntiles = int(os.environ.get("TILES", 300))
    which = os.environ.get("PLANE", "ssao")  # ssao: the operator's raw view normal; ssr: the post-processed one
    sc = scenes.surface_scene(P=P, sh_degree=2, seed=0)
    cam = scenes.orbit_camera(5, 64, W, H, radius=3.5)
    gi = dict(scenes.GI_DEFAULTS, start=16)
    raw = stage2_ref.operator_forward(orc, sc, cam, gi, 2)
    pos = raw["depth_pos"]
    nrm = raw["out_normal_view"]
    if which == "ssr":
        post = stage2_ref.gbuffer_post(orc, raw, cam["viewmatrix"])
        nrm = post["out_normal_view"]
    z = pos[2]
    fx, fy = stage2_ref.focal(cam)
    cx, cy = W / 2.0, H / 2.0
    g = scenes.GI_DEFAULTS
    radius, bias, thick, step, start = g["radius"], g["bias"], g["thick"], g["step"], g["start"]
    cm, hh = 0.5 * (bias - thick) - 1e-7, 0.5 * (bias + thick)
    rays = ray_table(g["delta"])
    B = 16
    hb, wb = (H + B - 1) // B, (W + B - 1) // B
    zp = np.zeros((hb * B, wb * B), np.float32)
    zp[:H, :W] = z
    zz = zp.reshape(hb, B, wb, B)
    zmax = zz.max(axis=(1, 3)).astype(np.float64)
    zmin = np.where(zz != 0, zz, np.inf).min(axis=(1, 3)).astype(np.float64)
    full = np.zeros((hb, wb), bool)
    full[:H // B, :W // B] = True
    rng = np.random.default_rng(0)
    js = np.arange(start, step)
    fj = js / step
    tot = dict(triples=0, lane_all=0, wave=0, wave_and_lane=0, rays=0, ray_all_wave=0, ray_all_lane=0, partial=0)
    by_cover = {}
    done = 0
    while done < ntiles:
        tx, ty = rng.integers(0, W // 8), rng.integers(0, H // 8)
        ys, xs = np.mgrid[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8]
        ys, xs = ys.ravel(), xs.ravel()
        n = nrm[:, ys, xs].T.astype(np.float64)
        with np.errstate(invalid="ignore", divide="ignore"):
            n = n / np.linalg.norm(n, axis=1, keepdims=True)
            up = np.array([0.0, 1.0, 0.0])
            t = up - n * n[:, 1:2]
            t = t / np.linalg.norm(t, axis=1, keepdims=True)
            b = np.cross(n, t)
            b = b / np.linalg.norm(b, axis=1, keepdims=True)
        live = np.isfinite(t).all(axis=1)
        if live.sum() == 0:
            continue
        done += 1
        if live.sum() < 64:
            tot["partial"] += 1
        n, t, b = n[live], t[live], b[live]
        p = pos[:, ys, xs].T.astype(np.float64)[live]
        a = 1 + p[:, 2] / 100
        s = a * a * radius
        # per-pixel constants of the projective march (gi.hip::make_fast / make_fast_tbn, mode 4)
        Ax, Ay, Dz = p[:, 0] * fx, p[:, 1] * fy, p[:, 2] + 1e-7
        M = np.stack([np.stack([t[:, 0] * s * fx, b[:, 0] * s * fx, n[:, 0] * s * fx], 1),
                      np.stack([t[:, 1] * s * fy, b[:, 1] * s * fy, n[:, 1] * s * fy], 1),
                      np.stack([t[:, 2] * s, b[:, 2] * s, n[:, 2] * s], 1)], 1)  # [L, 3(row), 3(col)]
        # ---- per-lane truth (all lanes certified on the tight table?) ----
        Bv = np.einsum("lrc,kc->lkr", M, rays)  # [L, R, 3]
        numx = Ax[:, None, None] + fj[None, None, :] * Bv[:, :, 0:1]
        numy = Ay[:, None, None] + fj[None, None, :] * Bv[:, :, 1:2]
        den = Dz[:, None, None] + fj[None, None, :] * Bv[:, :, 2:3]
        with np.errstate(invalid="ignore", divide="ignore"):
            ix = np.floor(numx / den + cx + 0.5)
            iy = np.floor(numy / den + cy + 0.5)
        inb = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H) & (den > 0)
        bx = np.clip(ix, 0, W - 1).astype(np.int64) // B
        by = np.clip(iy, 0, H - 1).astype(np.int64) // B
        lo, hi = den + cm - hh, den + cm + hh
        lane_cert = inb & full[by, bx] & ((lo > zmax[by, bx] + 1e-5) | ((hi < zmin[by, bx] - 1e-5) & (lo > 0)))
        lane_all = lane_cert.all(axis=0)  # [R, S]
        # ---- the wave bound: component-wise intervals over the live lanes ----
        Mlo, Mhi = M.min(axis=0), M.max(axis=0)  # [3, 3]
        rp, rn = np.maximum(rays, 0), np.minimum(rays, 0)  # [R, 3]
        Blo = rp @ Mlo.T + rn @ Mhi.T  # [R, 3(row)]
        Bhi = rp @ Mhi.T + rn @ Mlo.T
        nxl = Ax.min() + fj[None, :] * Blo[:, 0:1]
        nxh = Ax.max() + fj[None, :] * Bhi[:, 0:1]
        nyl = Ay.min() + fj[None, :] * Blo[:, 1:2]
        nyh = Ay.max() + fj[None, :] * Bhi[:, 1:2]
        dl = Dz.min() + fj[None, :] * Blo[:, 2:3]
        dh = Dz.max() + fj[None, :] * Bhi[:, 2:3]
        okd = dl > 1e-3
        with np.errstate(invalid="ignore", divide="ignore"):
            txl = np.minimum(nxl / dl, nxl / dh) + cx + 0.5
            txh = np.maximum(nxh / dl, nxh / dh) + cx + 0.5
            tyl = np.minimum(nyl / dl, nyl / dh) + cy + 0.5
            tyh = np.maximum(nyh / dl, nyh / dh) + cy + 0.5
        okd &= np.isfinite(txl) & np.isfinite(txh) & np.isfinite(tyl) & np.isfinite(tyh)
        bx0 = np.floor(np.where(okd, txl, 0) / B).astype(np.int64)
        bx1 = np.floor(np.where(okd, txh, 0) / B).astype(np.int64)
        by0 = np.floor(np.where(okd, tyl, 0) / B).astype(np.int64)
        by1 = np.floor(np.where(okd, tyh, 0) / B).astype(np.int64)
        span_ok = okd & (bx1 - bx0 <= 1) & (by1 - by0 <= 1) & (bx0 >= 0) & (by0 >= 0) & (bx1 < W // B) & (by1 < H // B)
        cbx0, cbx1 = np.clip(bx0, 0, wb - 1), np.clip(bx1, 0, wb - 1)
        cby0, cby1 = np.clip(by0, 0, hb - 1), np.clip(by1, 0, hb - 1)
        rmax = np.maximum(np.maximum(zmax[cby0, cbx0], zmax[cby0, cbx1]), np.maximum(zmax[cby1, cbx0], zmax[cby1, cbx1]))
        rmin = np.minimum(np.minimum(zmin[cby0, cbx0], zmin[cby0, cbx1]), np.minimum(zmin[cby1, cbx0], zmin[cby1, cbx1]))
        wlo, whi = dl + cm - hh, dh + cm + hh
        wave = span_ok & ((wlo > rmax + 1e-5) | ((whi < rmin - 1e-5) & (wlo > 0)))
        assert not (wave & ~lane_all).any(), "the wave bound certified a sample some lane cannot certify"
        tot["triples"] += wave.size
        tot["lane_all"] += int(lane_all.sum())
        tot["wave"] += int(wave.sum())
        tot["rays"] += wave.shape[0]
        tot["ray_all_wave"] += int(wave.all(axis=1).sum())
        tot["ray_all_lane"] += int(lane_all.all(axis=1).sum())
        key = "full" if live.sum() == 64 else "partial"
        c = by_cover.setdefault(key, [0, 0, 0])
        c[0] += wave.size; c[1] += int(wave.sum()); c[2] += int(lane_all.sum())
    print("plane %s, %d tiles (%d partially covered), %d live rays x %d steps" % (which, ntiles, tot["partial"], len(rays), len(js)))
    print("(tile, ray, step): all 64 lanes certify per lane %.3f | the wave bound certifies %.3f (= %.3f of those)"
          % (tot["lane_all"] / tot["triples"], tot["wave"] / tot["triples"], tot["wave"] / max(tot["lane_all"], 1)))
    print("(tile, ray): every step certified -- per lane %.3f, wave bound %.3f" % (tot["ray_all_lane"] / tot["rays"], tot["ray_all_wave"] / tot["rays"]))
    for k, c in by_cover.items():
        print("  %s tiles: wave %.3f, per lane %.3f" % (k, c[1] / c[0], c[2] / c[0]))


if __name__ == "__main__":
    main()
