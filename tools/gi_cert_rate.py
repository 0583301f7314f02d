#!/usr/bin/env python
"""How many SSAO march samples could a conservative coarse-depth test skip?  (CPU experiment, oracle G-buffer.)

For random 8x8 pixel tiles (= one wave of the HIP march) of the C2 view, every (ray, step) sample is projected as the
march does; a sample is *certifiable* if the min/max of the z plane over the BxB block that contains its pixel excludes
the hit interval [spz - thick, spz + bias] (zeros = empty pixels ignored for the min).  Reported per block size: the
fraction of certifiable samples, the fraction of (tile, ray, step) triples where ALL 64 lanes are certifiable (the
gather instruction can be skipped) and the mean fraction of active lanes in the remaining gathers."""
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


def main():
    orc.build()
    orc.set_threads(orc.max_threads())
    W = H = int(os.environ.get("RES", 800))
    P = int(os.environ.get("P", 300_000))
    sc = scenes.surface_scene(P=P, sh_degree=2, seed=0)
    cam = scenes.orbit_camera(5, 64, W, H, radius=3.5)
    gi = dict(scenes.GI_DEFAULTS, start=16)  # no march in the oracle: only the G-buffer is needed
    raw = stage2_ref.operator_forward(orc, sc, cam, gi, 2)
    nrm, pos = raw["out_normal_view"], raw["depth_pos"]
    z = pos[2]
    fx, fy = stage2_ref.focal(cam)
    cx, cy = W / 2.0, H / 2.0
    g = scenes.GI_DEFAULTS
    radius, bias, thick, step, start = g["radius"], g["bias"], g["thick"], g["step"], g["start"]
    # ray set
    sd = np.float32(g["delta"] * np.pi)
    rays = []
    phi = np.float32(0)
    while phi < 2 * np.pi:
        th = np.float32(0)
        while th <= 0.5 * np.pi:
            v = np.array([np.sin(th) * np.cos(phi), np.sin(th) * np.sin(phi), np.cos(th)], np.float64)
            rays.append(v / np.linalg.norm(v))
            th = np.float32(th + sd * 0.5)
        phi = np.float32(phi + sd)
    rays = np.array(rays)  # [512, 3]
    rng = np.random.default_rng(0)
    # tiles with geometry
    tiles = []
    while len(tiles) < 200:
        tx, ty = rng.integers(0, W // 8), rng.integers(0, H // 8)
        blk = nrm[:, ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8]
        if np.isfinite(blk).all() and (np.abs(blk).sum(0) > 0).all():
            tiles.append((tx, ty))
    pyr = {}
    for B in (8, 16, 32, 64):
        hb, wb = (H + B - 1) // B, (W + B - 1) // B
        zp = np.zeros((hb * B, wb * B), np.float32)
        zp[:H, :W] = z
        zz = zp.reshape(hb, B, wb, B)
        zmax = zz.max(axis=(1, 3))
        zmin = np.where(zz != 0, zz, np.inf).min(axis=(1, 3))
        pyr[B] = (zmin, zmax)
    stats = {B: dict(samples=0, cert=0, groups=0, groups_all=0, active_lanes=0) for B in pyr}
    # ray-level: one lookup at the block of the march's middle sample in a min/max table dilated by D blocks
    dil = {}
    for B, D in ((32, 2), (64, 1), (64, 2), (128, 1)):
        zmin, zmax = pyr.get(B, (None, None))
        if zmin is None:
            hb, wb = (H + B - 1) // B, (W + B - 1) // B
            zp = np.zeros((hb * B, wb * B), np.float32)
            zp[:H, :W] = z
            zz = zp.reshape(hb, B, wb, B)
            zmax = zz.max(axis=(1, 3))
            zmin = np.where(zz != 0, zz, np.inf).min(axis=(1, 3))
        hb, wb = zmin.shape
        pmin = np.pad(zmin, D, constant_values=np.inf)
        pmax = np.pad(zmax, D, constant_values=0)
        dmin = np.min([pmin[dy:dy + hb, dx:dx + wb] for dy in range(2 * D + 1) for dx in range(2 * D + 1)], axis=0)
        dmax = np.max([pmax[dy:dy + hb, dx:dx + wb] for dy in range(2 * D + 1) for dx in range(2 * D + 1)], axis=0)
        dil[(B, D)] = (dmin, dmax)
    ray_stats = {k: dict(rays=0, cert=0, groups=0, all=0, viol=0) for k in dil}
    # pair-level (round 3): ONE projection (of the pair's first sample) looked up in the 16-pixel table dilated by one
    # block certifies samples j and j+1 together; compared with certifying both on the tight table (two projections)
    zmin16, zmax16 = pyr[16]
    hb, wb = zmin16.shape
    pmin = np.pad(zmin16, 1, constant_values=np.inf)
    pmax = np.pad(zmax16, 1, constant_values=0)
    dmin16 = np.min([pmin[dy:dy + hb, dx:dx + wb] for dy in range(3) for dx in range(3)], axis=0)
    dmax16 = np.max([pmax[dy:dy + hb, dx:dx + wb] for dy in range(3) for dx in range(3)], axis=0)
    layout = {}
    group, gtab = {}, {}
    GROUP_TABLES = ((16, 1), (16, 2), (16, 3), (32, 1), (32, 2), (64, 1))
    pair = dict(groups=0, tight_all=0, dil_all=0, lanes=0, tight_lane=0, dil_lane=0, far=0)
    hits = 0
    for tx, ty in tiles:
        ys, xs = np.mgrid[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8]
        ys, xs = ys.ravel(), xs.ravel()
        n = nrm[:, ys, xs].T.astype(np.float64)
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        up = np.array([0.0, 1.0, 0.0])
        t = up - n * n[:, 1:2]
        t /= np.linalg.norm(t, axis=1, keepdims=True)
        b = np.cross(n, t)
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        p = pos[:, ys, xs].T.astype(np.float64)  # [64, 3]
        a = 1 + p[:, 2] / 100
        sv = rays[None, :, 0:1] * t[:, None, :] + rays[None, :, 1:2] * b[:, None, :] + rays[None, :, 2:3] * n[:, None, :]  # [64,512,3]
        js = np.arange(start, step)
        k = sv[:, :, None, :] * (js[None, None, :, None] * (a * a * radius / step)[:, None, None, None])
        sp = p[:, None, None, :] + k  # [64, 512, 8, 3]
        den = sp[..., 2] + 1e-7
        ix = np.floor(sp[..., 0] / den * fx + cx + 0.5).astype(np.int64)
        iy = np.floor(sp[..., 1] / den * fy + cy + 0.5).astype(np.int64)
        inb = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
        # ray is open until it leaves the image (hits ignored here: they are rare and only shorten marches)
        open_ = np.logical_and.accumulate(inb, axis=2)
        lo, hi = sp[..., 2] - thick, sp[..., 2] + bias
        zs = z[np.clip(iy, 0, H - 1), np.clip(ix, 0, W - 1)]
        hits += int(((zs <= hi) & (zs >= lo) & open_).sum())
        for (B, D), (dmin, dmax) in dil.items():
            mid = (step - start) // 2
            mx, my = np.clip(ix[:, :, mid], 0, W - 1) // B, np.clip(iy[:, :, mid], 0, H - 1) // B
            # every in-image sample of the ray must lie in the dilated window around the middle sample's block
            bxs, bys = np.clip(ix, 0, W - 1) // B, np.clip(iy, 0, H - 1) // B
            covered = ((np.abs(bxs - mx[:, :, None]) <= D) & (np.abs(bys - my[:, :, None]) <= D)) | ~open_
            lo_r = np.where(open_, lo, np.inf).min(axis=2)
            hi_r = np.where(open_, hi, -np.inf).max(axis=2)
            any_open = open_.any(axis=2)
            cert = ((hi_r < dmin[my, mx] - 1e-5) | (lo_r > dmax[my, mx] + 1e-5)) & (lo_r > 0) & covered.all(axis=2)
            st = ray_stats[(B, D)]
            st["rays"] += int(any_open.sum())
            st["cert"] += int((cert & any_open).sum())
            st["groups"] += cert.shape[1]
            st["all"] += int((cert | ~any_open).all(axis=0).sum())
            st["viol"] += int((~covered.all(axis=2) & any_open).sum())
        if True:
            bx, by = np.clip(ix, 0, W - 1) // 16, np.clip(iy, 0, H - 1) // 16
            tight = (((hi < zmin16[by, bx] - 1e-5) | (lo > zmax16[by, bx] + 1e-5)) & (lo > 0)) | ~open_
            a0, a1 = slice(0, ix.shape[2] - 1, 2), slice(1, ix.shape[2], 2)
            near = (np.abs(bx[..., a1] - bx[..., a0]) <= 1) & (np.abs(by[..., a1] - by[..., a0]) <= 1)
            dn, dx_ = dmin16[by[..., a0], bx[..., a0]], dmax16[by[..., a0], bx[..., a0]]
            both_dil = ((((hi[..., a0] < dn - 1e-5) & (hi[..., a1] < dn - 1e-5)) | ((lo[..., a0] > dx_ + 1e-5) & (lo[..., a1] > dx_ + 1e-5)))
                        & (lo[..., a0] > 0) & (lo[..., a1] > 0) & near & inb[..., a0] & inb[..., a1]) | ~open_[..., a0]
            both_tight = tight[..., a0] & tight[..., a1]
            # group-level: the four samples of a group lie on one image segment, so projecting its two END samples bounds
            # them all; one lookup in the B-pixel table dilated by D blocks around the first end's block certifies the group
            for (B, D) in GROUP_TABLES:
                if (B, D) not in gtab:
                    zmn, zmx = pyr[B]
                    h_, w_ = zmn.shape
                    pm, px_ = np.pad(zmn, D, constant_values=np.inf), np.pad(zmx, D, constant_values=0)
                    gtab[(B, D)] = (np.min([pm[dy:dy + h_, dx:dx + w_] for dy in range(2 * D + 1) for dx in range(2 * D + 1)], axis=0),
                                    np.max([px_[dy:dy + h_, dx:dx + w_] for dy in range(2 * D + 1) for dx in range(2 * D + 1)], axis=0))
                gmn, gmx = gtab[(B, D)]
                gbx, gby = np.clip(ix, 0, W - 1) // B, np.clip(iy, 0, H - 1) // B
                ns = ix.shape[2] // 4 * 4
                f, l = slice(0, ns, 4), slice(3, ns, 4)
                win = (np.abs(gbx[..., l] - gbx[..., f]) <= D) & (np.abs(gby[..., l] - gby[..., f]) <= D) & inb[..., f] & inb[..., l]
                glo = np.minimum(lo[..., f], lo[..., l])
                ghi = np.maximum(hi[..., f], hi[..., l])
                t0, t1 = gmn[gby[..., f], gbx[..., f]], gmx[gby[..., f], gbx[..., f]]
                ok = (((ghi < t0 - 1e-5) | (glo > t1 + 1e-5)) & (glo > 0) & win) | ~open_[..., f]
                live = open_[..., f].any(axis=0)
                tight4 = tight[..., :ns].reshape(tight.shape[0], tight.shape[1], -1, 4).all(axis=3)
                g = group.setdefault((B, D), [0, 0, 0, 0, 0])
                g[0] += int(live.sum()); g[1] += int((ok.all(axis=0) & live).sum())
                g[2] += int(open_[..., f].sum()); g[3] += int((ok & open_[..., f]).sum())
                g[4] += int((tight4.all(axis=0) & live).sum())
            # wave layouts: 8x8 pixels x 1 ray (shipped) against 4x4 pixels x 4 rays (neighbours in theta / in phi)
            nth = 1  # rays per azimuth: the table's inner loop runs over theta, whose first value 0 gives (0, 0, 1)
            while nth < len(rays) and rays[nth, 2] < 1.0 - 1e-12:
                nth += 1
            nph = len(rays) // nth
            q = tight.reshape(2, 4, 2, 4, nph, nth, -1).transpose(0, 2, 1, 3, 4, 5, 6).reshape(4, 16, nph, nth, -1)
            oq = open_.reshape(2, 4, 2, 4, nph, nth, -1).transpose(0, 2, 1, 3, 4, 5, 6).reshape(4, 16, nph, nth, -1)
            for name, (qq, oo) in dict(theta=(q[:, :, :, :nth // 4 * 4].reshape(4, 16, nph, nth // 4, 4, -1).transpose(0, 1, 4, 2, 3, 5),
                                              oq[:, :, :, :nth // 4 * 4].reshape(4, 16, nph, nth // 4, 4, -1).transpose(0, 1, 4, 2, 3, 5)),
                                       phi=(q.reshape(4, 16, nph // 4, 4, nth, -1).transpose(0, 1, 3, 2, 4, 5),
                                            oq.reshape(4, 16, nph // 4, 4, nth, -1).transpose(0, 1, 3, 2, 4, 5))).items():
                livegrp = oo.any(axis=(1, 2))
                layout.setdefault(name, [0, 0])
                layout[name][0] += int(livegrp.sum())
                layout[name][1] += int((qq.all(axis=(1, 2)) & livegrp).sum())
            livegrp = open_.any(axis=0)
            layout.setdefault("8x8x1", [0, 0])
            layout["8x8x1"][0] += int(livegrp.sum())
            layout["8x8x1"][1] += int((tight.all(axis=0) & livegrp).sum())
            live = open_[..., a0].any(axis=0)
            pair["groups"] += int(live.sum())
            pair["tight_all"] += int((both_tight.all(axis=0) & live).sum())
            pair["dil_all"] += int((both_dil.all(axis=0) & live).sum())
            pair["lanes"] += int(open_[..., a0].sum())
            pair["tight_lane"] += int((both_tight & open_[..., a0]).sum())
            pair["dil_lane"] += int((both_dil & open_[..., a0]).sum())
            pair["far"] += int((~near & open_[..., a1]).sum())
        for B, (zmin, zmax) in pyr.items():
            bx, by = np.clip(ix, 0, W - 1) // B, np.clip(iy, 0, H - 1) // B
            cert = ((hi < zmin[by, bx] - 1e-5) | (lo > zmax[by, bx] + 1e-5)) & (lo > 0)
            need = open_ & ~cert
            st = stats[B]
            st["samples"] += int(open_.sum())
            st["cert"] += int((open_ & cert).sum())
            grp_any_open = open_.any(axis=0)  # [512, 8]
            grp_need = need.any(axis=0)
            st["groups"] += int(grp_any_open.sum())
            st["groups_all"] += int((grp_any_open & ~grp_need).sum())
            st["active_lanes"] += int(need.sum())
    print("tiles %d, hit samples %.4f of open samples" % (len(tiles), hits / stats[8]["samples"]))
    for key, st in ray_stats.items():
        print("ray-level %s: rays certified %.3f | (tile, ray) with all 64 lanes certified %.3f | coverage violations %d"
              % (key, st["cert"] / st["rays"], st["all"] / st["groups"], st["viol"]))
    print("pairs (16-px blocks): lanes certified on the tight table (two projections) %.3f, on the 3x3-dilated table from one projection %.3f | "
          "(tile, ray, pair) with all 64 lanes certified: tight %.3f, dilated %.3f | second sample beyond the neighbouring block: %d"
          % (pair["tight_lane"] / pair["lanes"], pair["dil_lane"] / pair["lanes"], pair["tight_all"] / pair["groups"],
             pair["dil_all"] / pair["groups"], pair["far"]))
    for k, g in group.items():
        print("group of 4 samples, %d-px table dilated by %d: lanes certified %.3f | (tile, ray, group) with all 64 lanes certified %.3f "
              "(all four on the tight 16-px table: %.3f)" % (k[0], k[1], g[3] / g[2], g[1] / g[0], g[4] / g[0]))
    print("wave layouts, (wave, step) groups with all 64 lanes certified (16-px blocks): " +
          ", ".join("%s %.3f" % (k, v[1] / v[0]) for k, v in layout.items()))
    for B, st in stats.items():
        rem = st["groups"] - st["groups_all"]
        print("block %2d: certifiable samples %.3f | wave-gathers skipped entirely %.3f | lanes active in the remaining gathers %.2f / 64"
              % (B, st["cert"] / st["samples"], st["groups_all"] / st["groups"], st["active_lanes"] / max(rem, 1)))


if __name__ == "__main__":
    main()
