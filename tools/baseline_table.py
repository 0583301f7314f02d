"""Fills BASELINE.md's result table rows that bench.py does not print: the CPU restatement at C1 (1 thread and all
host cores, forward only) and the HIP path at C1 / C3 (inference, forward only).  Run on the GPU box."""
import os, sys, time, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gi-gs_amd"))
import numpy as np, torch
import scenes, pipeline
from oracle import oracle as orc

KEYS = ["means3D", "opacities", "normal", "albedo", "roughness", "metallic", "shs", "scales", "rotations"]
res = {}
# ---- C1 on the CPU restatement
orc.build()
sc = scenes.random_scene(P=10_000, sh_degree=0, seed=0)
cam = scenes.orbit_camera(0, 8, 400, 400)
for threads in (1, orc.max_threads()):
    orc.set_threads(threads)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        r = orc.Rasterizer()
        r.forward(bg=np.zeros(3, np.float32), **{k: sc[k] for k in KEYS}, sh_degree=0, viewmatrix=cam["viewmatrix"],
                  projmatrix=cam["projmatrix"], campos=cam["campos"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                  image_height=400, image_width=400)
        best = min(best, time.perf_counter() - t0)
    res["C1_cpu_%d_threads_fwd_ms" % threads] = round(best * 1e3, 2)
# ---- HIP: C1 forward, C3 inference forward (rasterize + filters + SSAO + shade + SSR, no backward)
dev = "cuda:0"
import pbr
def hip_forward_ms(sc, cam, sh_degree, inference, full):
    g = {k: torch.from_numpy(sc[k]).to(dev) for k in KEYS}
    camt = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    gi = scenes.GI_DEFAULTS
    H, W = cam["image_height"], cam["image_width"]
    light = pbr.CubemapLight(base_res=256).to(dev); lut = pbr.get_brdf_lut().to(dev)
    rays = pipeline.canonical_rays(cam, dev); vd = pipeline.view_dirs_for(camt, rays, dev)
    front = pipeline.Stage2Front(light, lut)
    def once():
        with torch.no_grad():
            out, _, st = pipeline.rasterize(camt, g, sh_degree, torch.zeros(3, device=dev), gi, inference=inference)
            if full:
                (_, radii, _, _, nfd, nm, occ, alb, rough, metal, onv, dpos) = out
                (rd, rf, mf, F0, lin, onv2, maskf) = front(nfd, nm, onv, alb, rough, metal, occ, st.viewmatrix, vd)
                ssr = pipeline.Gaussian_SSR(cam["tanfovx"], cam["tanfovy"], W, H, gi["radius"], gi["bias"], gi["thick"],
                                            gi["delta"], gi["step"], gi["start"])
                ssr(onv2, dpos, lin, alb, rf, mf, F0)
    for _ in range(3): once()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): once()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / 20 * 1e3, 3)
res["C1_hip_fwd_ms"] = hip_forward_ms(sc, cam, 0, False, False)
sc3 = scenes.surface_scene(P=300_000, sh_degree=2, seed=0)
cam3 = scenes.orbit_camera(0, 64, 800, 800, radius=3.5)
res["C3_hip_inference_fwd_ms"] = hip_forward_ms(sc3, cam3, 2, True, True)
res["host_cores"] = orc.max_threads()
print(json.dumps(res))
