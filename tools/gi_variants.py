#!/usr/bin/env python
"""Accuracy and speed of the GI march variants (GIGS_GI_MARCH) on the GPU.

Mode "exact" reproduces the CPU oracle bit for bit (tests/test_gpu_parity.py::test_gi_passes_match_oracle), so it
is the yardstick here: every other mode is compared with it on the same inputs -- mean per-pixel L1, largest
difference, fraction of changed pixels -- for SSAO (operator inputs: raw view-space normal, filtered depth
positions) and SSR (train.py's inputs), on the C2 bench view and on the GI scenes of the parity tests, at several GI
settings.  Timing: hipEvents around 10 launches after 2 warm-up launches.

    python tools/gi_variants.py --out gpurun_out/gi_variants.json
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("gi-gs_amd")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import diff_gaussian_rasterization as dgr  # noqa: E402
import gigs_lib  # noqa: E402
import pipeline  # noqa: E402
import scenes  # noqa: E402

DEV = "cuda:0"
MODES = ["exact", "hoist", "hoist_fma", "proj_nr", "proj"]
KEYS = ["means3D", "opacities", "normal", "albedo", "roughness", "metallic", "shs", "scales", "rotations"]


def gbuffer(sc, cam, gi, sh_degree):
    g = {k: torch.from_numpy(sc[k]).to(DEV) for k in KEYS}
    camt = {k: (torch.from_numpy(v).to(DEV) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    gigs_lib.set_options(gi_march="exact")
    with torch.no_grad():
        res = pipeline.render(camt, g, sh_degree, torch.zeros(3, device=DEV), dict(gi, start=gi["step"]))
        out, _, st = pipeline.rasterize(camt, g, sh_degree, torch.zeros(3, device=DEV), dict(gi, start=gi["step"]))
    # operator SSAO inputs: raw out_normal_view (tuple index 10) and the filtered depth positions (index 11)
    raw_nview, posf = out[10], out[11]
    albedo, rough, metal = res["albedo_map"], res["roughness_map"], res["metallic_map"]
    F0 = (1.0 - metal) * 0.04 + albedo * metal
    rgb = pipeline.srgb_to_linear((albedo * 0.7).clamp(0, 1))  # any plausible linear radiance plane
    return dict(raw_nview=raw_nview.contiguous(), posf=posf.contiguous(), nview=res["out_normal_view"].contiguous(),
                rgb=rgb.contiguous(), albedo=albedo.contiguous(), rough=rough.contiguous(), metal=metal.contiguous(),
                F0=F0.contiguous())


def run_mode(mode, gb, cam, gi, reps):
    gigs_lib.set_options(gi_march=mode)
    W, H = cam["image_width"], cam["image_height"]
    fx, fy = W / (2 * cam["tanfovx"]), H / (2 * cam["tanfovy"])
    a = (gi["radius"], gi["bias"], gi["thick"], gi["delta"], gi["step"], gi["start"])

    def ssao():
        return dgr._C.SSAO(W, H, fx, fy, *a, gb["raw_nview"], gb["posf"])

    def ssr():
        return dgr._C.SSR(W, H, fx, fy, *a, gb["nview"], gb["posf"], gb["rgb"], gb["albedo"], gb["rough"], gb["metal"], gb["F0"])

    def timed(fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        return r, e0.elapsed_time(e1) / reps

    occ, t_ao = timed(ssao)
    (col, abd), t_sr = timed(ssr)
    return dict(occ=occ.cpu().numpy(), col=col.cpu().numpy(), abd=abd.cpu().numpy()), dict(ssao_ms=t_ao, ssr_ms=t_sr)


def diff(a, b):
    fin = np.isfinite(a) & np.isfinite(b)
    d = np.abs(a[fin] - b[fin])
    return dict(mean_l1=float(d.mean()), max=float(d.max()), changed_frac=float((d > 0).mean()),
                over_1e5_frac=float((d > 1e-5).mean()), nan_equal=bool(np.array_equal(np.isnan(a), np.isnan(b))))


def cases(full):
    gi0 = scenes.GI_DEFAULTS
    yield ("gi_test_208x160", scenes.surface_scene(P=30_000, sh_degree=1, seed=3, scale_mu=0.02),
           scenes.orbit_camera(0, 4, 208, 160, radius=3.5), 1,
           [gi0, dict(gi0, step=12, start=5, delta=0.125), dict(gi0, start=0), dict(gi0, radius=1.6, start=4)])
    yield ("operator_test_176x144", scenes.surface_scene(P=15_000, sh_degree=2, seed=6, scale_mu=0.025),
           scenes.orbit_camera(2, 6, 176, 144, radius=3.5), 2, [gi0])
    yield ("ragged_611x403", scenes.surface_scene(P=80_000, sh_degree=2, seed=2, scale_mu=0.015),
           scenes.orbit_camera(7, 16, 611, 403, radius=3.2), 2, [gi0])
    if full:
        sc = scenes.surface_scene(P=300_000, sh_degree=2, seed=0)
        for v in (5, 21, 40):
            yield ("c2_view%d_800x800" % v, sc, scenes.orbit_camera(v, 64, 800, 800, radius=3.5), 2,
                   [gi0] + ([dict(gi0, start=0)] if v == 5 else []))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only-sweep", action="store_true")
    ap.add_argument("--modes", default="", help="comma-separated subset of the non-exact modes (default: all)")
    ap.add_argument("--no-sweep", action="store_true")
    args = ap.parse_args()
    if args.modes:
        MODES[1:] = args.modes.split(",")
    report = {"modes": MODES, "cases": []}
    for name, sc, cam, deg, gis in ([] if args.only_sweep else cases(not args.quick)):
        for gi in gis:
            gb = gbuffer(sc, cam, gi, deg)
            ref, t_ref = run_mode("exact", gb, cam, gi, args.reps)
            rec = {"case": name, "gi": gi, "exact_ms": t_ref, "modes": {}}
            for mode in MODES[1:]:
                got, t = run_mode(mode, gb, cam, gi, args.reps)
                rec["modes"][mode] = dict(t, occlusion=diff(got["occ"], ref["occ"]), ssr_color=diff(got["col"], ref["col"]),
                                          ssr_abd=diff(got["abd"], ref["abd"]))
            report["cases"].append(rec)
            print("%s step=%d start=%d delta=%g radius=%g: exact %.3f / %.3f ms" %
                  (name, gi["step"], gi["start"], gi["delta"], gi["radius"], t_ref["ssao_ms"], t_ref["ssr_ms"]), flush=True)
            for mode, m in rec["modes"].items():
                print("   %-10s ssao %.3f ms ssr %.3f ms | occ L1 %.2e max %.1e changed %.4f | ssr L1 %.2e changed %.4f | abd L1 %.2e"
                      % (mode, m["ssao_ms"], m["ssr_ms"], m["occlusion"]["mean_l1"], m["occlusion"]["max"],
                         m["occlusion"]["changed_frac"], m["ssr_color"]["mean_l1"], m["ssr_color"]["changed_frac"],
                         m["ssr_abd"]["mean_l1"]), flush=True)
    # workgroup pixel rectangle (GIGS_GI_TILE_LOG2W): once the march is no longer VALU-bound the z-plane gathers of a
    # wave cost by the number of image rows they touch (tools/microbench/gather_rate.hip)
    if not args.quick and not args.no_sweep:
        sc = scenes.surface_scene(P=300_000, sh_degree=2, seed=0)
        cam = scenes.orbit_camera(5, 64, 800, 800, radius=3.5)
        gi = scenes.GI_DEFAULTS
        gb = gbuffer(sc, cam, gi, 2)
        report["tile_sweep"] = {}
        for log2w in (2, 3, 4, 5, 6):
            gigs_lib.set_options(gi_tile_log2w=log2w)
            row = {}
            for mode in ("exact", "hoist_fma", "proj"):
                _, t = run_mode(mode, gb, cam, gi, args.reps)
                row[mode] = t
            gigs_lib.set_options(gi_cert=0)
            _, row["proj_nocert"] = run_mode("proj", gb, cam, gi, args.reps)
            gigs_lib.set_options(gi_cert=1)
            report["tile_sweep"]["%dx%d" % (1 << log2w, 64 >> log2w)] = row
            print("tile %2dx%-2d " % (1 << log2w, 64 >> log2w) + "  ".join("%s %.3f/%.3f ms" % (m, v["ssao_ms"], v["ssr_ms"]) for m, v in row.items()), flush=True)
        gigs_lib.set_options(gi_tile_log2w=3)
    gigs_lib.set_options(gi_march="proj")
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
