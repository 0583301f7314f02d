#!/usr/bin/env python
"""Experiment: how many renders/s does ONE MI355X deliver with K views of a micro-batch in flight at once?

The headline step renders one view per step; its phases alternate between latency-bound kernels (binning, blend, the
backward walks) that leave most SIMDs idle and the VALU-bound marches.  K host threads, each with its own stream, library
context scopes, whole-step graphs and gradient buffers, replay K independent views concurrently (the micro-batch a
view-parallel trainer would hand to one GPU); the GPU's queues interleave their kernels.  Prints renders/s for K = 1, 2, 3.
(Timing experiment: every thread owns copies of the parameters and of the light.)"""
import importlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("gi-gs_amd")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pbr  # noqa: E402
import pipeline  # noqa: E402
import scenes  # noqa: E402

PARAM_KEYS = ("means3D", "shs", "opacities", "normal", "albedo", "roughness", "metallic", "scales", "rotations")


def main():
    cfg = os.environ.get("CFG", "c2")
    P, W, H, deg = dict(c2=(300_000, 800, 800, 2), c4=(3_000_000, 1237, 822, 3))[cfg]
    steps = int(os.environ.get("STEPS", 40))
    dev = torch.device("cuda", 0)
    sc = scenes.surface_scene(P=P, sh_degree=deg, seed=0)
    gi = dict(scenes.GI_DEFAULTS)
    cams = [scenes.orbit_camera(i, 64, W, H, radius=3.5) for i in range(0, 64, 8)]
    cams_t = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, W, device=dev), indexing="ij")
    gt = torch.stack([0.5 + 0.3 * torch.sin(6 * xx), 0.5 + 0.3 * torch.cos(5 * yy), 0.4 + 0.2 * xx * yy])
    lut = pbr.get_brdf_lut().to(dev)
    rays = pipeline.canonical_rays(cams[0], dev)
    vds = [pipeline.view_dirs_for(c, rays, dev) for c in cams_t]
    torch.cuda.synchronize()

    class Worker:
        def __init__(self, k):
            self.k = k
            self.stream = torch.cuda.Stream()
            torch.manual_seed(3)
            self.light = pbr.CubemapLight(base_res=256, device=dev)
            self.g = {key: torch.from_numpy(sc[key]).to(dev).requires_grad_(True) for key in PARAM_KEYS}
            self.step = None

        def warm(self):  # captures: one worker at a time
            with torch.cuda.stream(self.stream):
                self.step = pipeline.Stage2Step(self.light, lut, gi, deg, graphs=True, fused=True)
                for i in range(4):
                    self.one(i)
                torch.cuda.synchronize()

        def one(self, i):
            for p in list(self.g.values()) + [self.light.base]:
                p.grad = None
            vi = (i * 3 + self.k) % len(cams_t)
            self.step(cams_t[vi], self.g, gt, vds[vi])

        def run(self, n, start, done, per_step=None):
            with torch.cuda.stream(self.stream):
                start.wait()
                for i in range(n):
                    self.one(i)
                    if per_step is not None:  # a micro-batch step: every view's backward has ended before the next step starts
                        self.stream.synchronize()
                        per_step.wait()
                self.stream.synchronize()
            done.append(time.perf_counter())

    workers = [Worker(k) for k in range(4)]
    for w in workers:
        w.warm()
    for K, joined in ((1, False), (2, False), (2, True), (3, True), (4, True), (1, True)):
        start = threading.Barrier(K + 1)
        done = []
        per_step = threading.Barrier(K) if joined else None
        ts = [threading.Thread(target=workers[k].run, args=(steps, start, done, per_step)) for k in range(K)]
        for t in ts:
            t.start()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        start.wait()
        for t in ts:
            t.join()
        dt = max(done) - t0
        print("%s: %d view(s) in flight%s: %.1f renders/s (%.3f ms per view, %.3f ms per micro-batch)"
              % (cfg, K, " (joined every step)" if joined else " (free-running)", K * steps / dt, 1e3 * dt / (K * steps), 1e3 * dt / steps), flush=True)


if __name__ == "__main__":
    main()
