"""Diagnostic (round 3): does the whole-step graph's replay depend on WHICH pooled torch stream the capture gets?
Runs the binning-overflow scenario of tests/test_gpu_pbr.py in a fresh process per k, after k extra torch.cuda.Stream() calls."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, importlib, numpy as np, torch
sys.path.insert(0, %r); importlib.import_module("gi-gs_amd")
sys.path.insert(0, %r + "/tests")
import pbr, pipeline, scenes
k = int(sys.argv[1])
keep = [torch.cuda.Stream() for _ in range(k)]
DEV = "cuda:0"
tt = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).to(DEV).requires_grad_(grad)
KEYS = ["means3D", "opacities", "normal", "albedo", "roughness", "metallic", "shs", "scales", "rotations"]
sc = scenes.surface_scene(P=20_000, sh_degree=2, seed=4, scale_mu=0.03)
gi = scenes.GI_DEFAULTS
H, W = 160, 208
cam = scenes.orbit_camera(2, 6, W, H, radius=3.5)
camt = {k_: (tt(v) if isinstance(v, np.ndarray) else v) for k_, v in cam.items()}
gt = torch.rand(3, H, W, device=DEV) * 0.5
lut = pbr.get_brdf_lut().to(DEV)
vd = pipeline.view_dirs_for(camt, pipeline.canonical_rays(cam, DEV), DEV)
light = pbr.CubemapLight(base_res=64, device=DEV)
g = {k_: tt(sc[k_], grad=True) for k_ in KEYS}
step = pipeline.Stage2Step(light, lut, gi, 2, fused=True, graphs=True)
step.whole = pipeline.WholeStepGraph(step, camt, g)
step.whole.capacity = 65536
o = step(camt, g, gt, vd)
torch.cuda.synchronize()
for i in range(3):
    for t in list(g.values()) + [light.base]:
        t.grad = None
    o = step(camt, g, gt, vd)
torch.cuda.synchronize()
print("ok", k, step.whole.recaptures, float(o["loss"]))
''' % (ROOT, ROOT)

res = {}
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 34):
    r = subprocess.run([sys.executable, "-c", CHILD, str(k)], capture_output=True, text=True, timeout=120)
    res[k] = r.returncode
    print(k, r.returncode, r.stdout.strip()[-60:], flush=True)
print("crashing k:", [k for k, v in res.items() if v != 0])
