#!/usr/bin/env python
"""Soak run of the captured stage-2 iteration: ITERS iterations over 8 views, a prune every PRUNE iterations (every parameter
tensor and its Adam moments replaced -> the graphs re-captured), a learning-rate change every 97 iterations, with and without
the frozen-geometry cache.  Reports iterations/s, re-captures, the allocator's high-water mark at a few points (a leak of graph
memory pools or scratch would show as growth from one prune interval to the next) and that losses / parameters stay finite."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("gi-gs_amd")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import densify  # noqa: E402
import pbr  # noqa: E402
import pipeline  # noqa: E402
import scenes  # noqa: E402
import train_iteration as ti  # noqa: E402


def main():
    iters = int(os.environ.get("ITERS", 1500))
    prune_every = int(os.environ.get("PRUNE", 400))
    dev = torch.device("cuda:0")
    H = W = int(os.environ.get("RES", 256))
    P = int(os.environ.get("P", 60_000))
    sc = scenes.surface_scene(P=P, sh_degree=2, seed=8, scale_mu=0.02)
    cams = [scenes.orbit_camera(i, 8, W, H, radius=3.5) for i in range(8)]
    cams = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    gi = scenes.GI_DEFAULTS
    lut = pbr.get_brdf_lut().to(dev)
    rays = pipeline.canonical_rays(cams[0], dev)
    vds = [pipeline.view_dirs_for(c, rays, dev) for c in cams]
    torch.manual_seed(1)
    gts = [torch.rand(3, H, W, device=dev) * 0.6 for _ in cams]
    for cache in (False, True):
        torch.manual_seed(2)
        light = pbr.CubemapLight(base_res=64, device=dev)
        raw = ti.raw_from_scene(sc, dev)
        # frozen geometry is what stage 2 converges to: zero learning rates for it make the cache take effect at once
        lrs = dict(xyz=0.0, scaling=0.0, rotation=0.0, opacity=0.0, normal=0.0) if cache else None
        tr = ti.Stage2Trainer(raw, light, lut, gi, 2, graphs=True, geometry_cache=cache, lrs=lrs)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        marks = []
        for it in range(iters):
            out = tr.iteration(cams[it % 8], gts[it % 8], vds[it % 8])
            if it % 97 == 96:
                tr.set_lr("albedo", 0.05 if (it // 97) % 2 else 0.02)
            if it % prune_every == prune_every - 1:
                loss = float(out["loss"])
                assert np.isfinite(loss), (it, loss)
                Pn = tr.raw["xyz"].shape[0]
                mask = torch.zeros(Pn, dtype=torch.bool, device=dev)
                mask[::17] = True
                new, _ = densify.prune_points(tr.optimizer, densify.DensifyState(Pn, dev), mask)
                tr.replace_parameters(dict(new))
                torch.cuda.synchronize()
                marks.append((it + 1, tr.raw["xyz"].shape[0], torch.cuda.memory_allocated() >> 20, torch.cuda.max_memory_allocated() >> 20,
                              torch.cuda.memory_reserved() >> 20, round(loss, 6)))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = all(bool(torch.isfinite(p).all()) for p in tr.raw.values()) and bool(torch.isfinite(light.base).all())
        whole = tr.stepper.whole
        stats = getattr(tr.stepper, "geom_cache", None)
        recaptures = sum(w.recaptures for w in tr.stepper._wholes.values()) if getattr(tr.stepper, "_wholes", None) else getattr(whole, "recaptures", -1)
        print("geometry cache %s: %d iterations in %.1f s (%.0f it/s incl. %d re-captures), parameters finite: %s"
              % ("on" if cache else "off", iters, dt, iters / dt, recaptures, ok), flush=True)
        if stats is not None and hasattr(stats, "stats"):
            print("   cache stats:", stats.stats, flush=True)
        for m in marks:
            print("   after iteration %d: P = %d, allocated %d MiB, peak %d MiB, reserved %d MiB, loss %.6f" % m, flush=True)
        tr.close()
        del tr, raw, light
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
