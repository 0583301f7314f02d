#!/usr/bin/env python
"""Noise floor of "parity with the reference's CUDA binary".

The reference is compiled by nvcc with its default -fmad=true (R/setup.py passes no fp flags), so the
binary fuses `a*b+c` wherever nvcc/ptxas see the pattern -- e.g. `dir.x * fx + cx` in get_coord
(ssr.h:133) and the `power` expression of the blend (forward.cu:531).  The oracle is built twice from the
same restated lines: libgigs_oracle.so (-ffp-contract=off) and libgigs_oracle_fma.so (-ffp-contract=fast
-mfma).  Neither is the CUDA binary; both are legitimate fp32 compilations of the cited source, and the
distance between them is the floor below which "matches the reference" cannot be decided here.

    python tools/oracle_noise.py [--scenes small,c2] [--threads N] [--out profiles/r02/oracle_noise.json]

TEST INFRASTRUCTURE (uses oracle/ only; nothing of the product runs here).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("gi-gs_amd")

import numpy as np  # noqa: E402

import scenes  # noqa: E402
from oracle import oracle as orc_off  # noqa: E402

KEYS = ["means3D", "opacities", "normal", "albedo", "roughness", "metallic", "shs", "scales", "rotations"]
PLANES = ["color", "opacity", "depth", "normal", "normal_view", "pos", "albedo", "roughness", "metallic"]


def run_oracle(orc, sc, cam, gi, sh_degree, gi_inputs=None, backward=True):
    """Rasterizer forward, the operator's filters, SSAO, SSR, backward -- the sequence bench.py's cpu_baseline times."""
    H, W = cam["image_height"], cam["image_width"]
    fx, fy = W / (2 * cam["tanfovx"]), H / (2 * cam["tanfovy"])
    r = orc.Rasterizer()
    out = r.forward(bg=np.zeros(3, np.float32), **{k: sc[k] for k in KEYS}, sh_degree=sh_degree,
                    viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], campos=cam["campos"],
                    tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], image_height=H, image_width=W)
    res = {k: out[k] for k in PLANES}
    res["radii"] = out["radii"]
    for k in ("point_list", "ranges", "n_contrib", "keys"):
        res[k] = r.state(k)
    depth_f = orc.median3x3(out["depth"])
    nd, pos = orc.depth_to_normal(W, H, fx, fy, cam["viewmatrix"], depth_f)
    res["normal_from_depth"] = orc.bilateral3x3(nd)
    posf = orc.median3x3(pos)
    res["depth_pos"] = posf
    a = (gi["radius"], gi["bias"], gi["thick"], gi["delta"], gi["step"], gi["start"])
    if gi_inputs is None:
        F0 = ((1.0 - out["metallic"]) * 0.04 + out["albedo"] * out["metallic"]).astype(np.float32)
        gi_inputs = dict(normal_view=out["normal_view"], posf=posf, rgb=out["color"], albedo=out["albedo"],
                         roughness=out["roughness"], metallic=out["metallic"], F0=F0)
    g = gi_inputs
    res["occlusion"] = orc.ssao(W, H, fx, fy, *a, g["normal_view"], g["posf"])
    res["ssr_color"], res["ssr_abd"] = orc.ssr(W, H, fx, fy, *a, g["normal_view"], g["posf"], g["rgb"], g["albedo"],
                                               g["roughness"], g["metallic"], g["F0"])
    if backward:
        rng = np.random.default_rng(7)
        gr = {k: rng.normal(size=(c, H, W)).astype(np.float32) / (H * W)
              for k, c in (("color", 3), ("opacity", 1), ("depth", 1), ("normal", 3), ("albedo", 3), ("roughness", 1),
                           ("metallic", 1))}
        res["grads"] = r.backward(**{"grad_" + k: v for k, v in gr.items()})
    return res, gi_inputs


def compare(a, b):
    """Per-plane mean L1 (over finite entries), changed-pixel fraction, index equality, gradient rel-L1."""
    rep = {}
    for k in PLANES + ["normal_from_depth", "depth_pos", "occlusion", "ssr_color", "ssr_abd"]:
        x, y = a[k], b[k]
        fin = np.isfinite(x) & np.isfinite(y)
        rep[k] = {"mean_l1": float(np.abs(x[fin] - y[fin]).mean()) if fin.any() else 0.0,
                  "max": float(np.abs(x[fin] - y[fin]).max()) if fin.any() else 0.0,
                  "changed_frac": float((x[fin] != y[fin]).mean()) if fin.any() else 0.0,
                  "nan_pattern_equal": bool(np.array_equal(np.isnan(x), np.isnan(y)))}
    rep["radii_equal"] = bool(np.array_equal(a["radii"], b["radii"]))
    rep["radii_diff"] = int((a["radii"] != b["radii"]).sum())
    same_len = a["point_list"].shape == b["point_list"].shape
    rep["num_rendered"] = [int(a["point_list"].shape[0]), int(b["point_list"].shape[0])]
    rep["point_list_equal"] = bool(same_len and np.array_equal(a["point_list"], b["point_list"]))
    rep["point_list_diff"] = int((a["point_list"] != b["point_list"]).sum()) if same_len else None
    rep["ranges_equal"] = bool(np.array_equal(a["ranges"], b["ranges"]))
    rep["n_contrib_flips"] = int((a["n_contrib"] != b["n_contrib"]).sum())
    if "grads" in a and "grads" in b:
        rep["grads_rel_l1"] = {k: float(np.abs(a["grads"][k] - b["grads"][k]).mean() /
                                        max(np.abs(a["grads"][k]).mean(), 1e-30)) for k in a["grads"]}
    return rep


def scene_set(which):
    gi = scenes.GI_DEFAULTS
    if which == "small":
        for seed, (W, H) in ((0, (160, 128)), (1, (203, 117))):
            sc = scenes.surface_scene(P=8000, sh_degree=2, seed=seed, scale_mu=0.03)
            yield "small_seed%d_%dx%d" % (seed, W, H), sc, scenes.orbit_camera(seed, 4, W, H, radius=3.5), gi, 2
    elif which == "mid":
        sc = scenes.surface_scene(P=60_000, sh_degree=2, seed=0, scale_mu=0.02)
        yield "mid_60k_400x400", sc, scenes.orbit_camera(3, 64, 400, 400, radius=3.5), gi, 2
    elif which == "c2":
        sc = scenes.surface_scene(P=300_000, sh_degree=2, seed=0)
        yield "c2_300k_800x800", sc, scenes.orbit_camera(5, 64, 800, 800, radius=3.5), gi, 2
    else:
        raise SystemExit("unknown scene set " + which)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="small,mid")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    orc_off.build()
    orc_fma = orc_off.variant("fma")
    n = args.threads or orc_off.max_threads()
    orc_off.set_threads(n)
    orc_fma.set_threads(n)
    report = {"threads": n, "builds": {"off": "-O2 -ffp-contract=off", "fma": "-O2 -ffp-contract=fast -mfma"},
              "scenes": {}}
    for which in args.scenes.split(","):
        for name, sc, cam, gi, deg in scene_set(which):
            t0 = time.perf_counter()
            a, gi_in = run_oracle(orc_off, sc, cam, gi, deg)
            b, _ = run_oracle(orc_fma, sc, cam, gi, deg)
            # the GI march alone: both builds on IDENTICAL inputs (the G-buffer of the contraction-off build)
            c, _ = run_oracle(orc_fma, sc, cam, gi, deg, gi_inputs=gi_in, backward=False)
            full = compare(a, b)
            gi_only = {k: compare(a, c)[k] for k in ("occlusion", "ssr_color", "ssr_abd")}
            report["scenes"][name] = {"whole_path": full, "gi_march_on_identical_inputs": gi_only,
                                      "seconds": round(time.perf_counter() - t0, 1)}
            print(name, "%.1fs" % (time.perf_counter() - t0), flush=True)
            for k in ("color", "depth", "normal_view", "albedo", "occlusion", "ssr_color"):
                print("   %-12s whole-path L1 %.3e  changed %.4f" % (k, full[k]["mean_l1"], full[k]["changed_frac"]))
            for k, v in gi_only.items():
                print("   %-12s march-only L1 %.3e  changed %.4f" % (k, v["mean_l1"], v["changed_frac"]))
            print("   radii diff %d, point_list equal %s (diff %s), n_contrib flips %d" %
                  (full["radii_diff"], full["point_list_equal"], full["point_list_diff"], full["n_contrib_flips"]))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
