"""Diagnostic (round 3): capture / replay / destroy whole-step graphs repeatedly in ONE process: does hipGraphLaunch fail after
some number of graph execs have existed (the GPU test session crashes in hip::Graph::UpdateStreams on a dead parallel stream)?"""
import gc
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
importlib.import_module("gi-gs_amd")
import pbr  # noqa: E402
import pipeline  # noqa: E402
import scenes  # noqa: E402

DEV = "cuda:0"
tt = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).to(DEV).requires_grad_(grad)  # noqa: E731
KEYS = ["means3D", "opacities", "normal", "albedo", "roughness", "metallic", "shs", "scales", "rotations"]
sc = scenes.surface_scene(P=8000, sh_degree=2, seed=4, scale_mu=0.03)
gi = scenes.GI_DEFAULTS
H, W = 128, 160
cam = scenes.orbit_camera(2, 6, W, H, radius=3.5)
camt = {k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
gt = torch.rand(3, H, W, device=DEV) * 0.5
lut = pbr.get_brdf_lut().to(DEV)
vd = pipeline.view_dirs_for(camt, pipeline.canonical_rays(cam, DEV), DEV)
mode = sys.argv[2] if len(sys.argv) > 2 else "free"
hold = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    light = pbr.CubemapLight(base_res=64, device=DEV)
    g = {k: tt(sc[k], grad=True) for k in KEYS}
    step = pipeline.Stage2Step(light, lut, gi, 2, fused=True, graphs=True)
    for i in range(3):
        for t in list(g.values()) + [light.base]:
            t.grad = None
        o = step(camt, g, gt, vd)
    torch.cuda.synchronize()
    print(it, "ok", float(o["loss"]), flush=True)
    if mode == "hold":
        hold.append(step)
    else:
        del step, light, g, o
        gc.collect()
print("done")
