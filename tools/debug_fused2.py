import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gi-gs_amd"))
import numpy as np, torch
import scenes, pipeline, pbr
DEV = "cuda:0"
tt = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).to(DEV).requires_grad_(grad)
sc = scenes.surface_scene(P=10_000, sh_degree=2, seed=9, scale_mu=0.025)
gi = scenes.GI_DEFAULTS
H, W = 128, 160
cams = [scenes.orbit_camera(i, 6, W, H, radius=3.5) for i in (1, 4)]
camts = [{k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
torch.manual_seed(1)
gt = torch.rand(3, H, W, device=DEV) * 0.5
lut = pbr.get_brdf_lut().to(DEV)
rays = pipeline.canonical_rays(cams[0], DEV)
vds = [pipeline.view_dirs_for(c, rays, DEV) for c in camts]
KEYS = ["means3D", "opacities", "normal", "shs", "albedo", "roughness", "metallic", "scales", "rotations"]
modes = sys.argv[1:] or ["fused", "fused_graph"]
for mode in modes:
    torch.manual_seed(2)
    light = pbr.CubemapLight(base_res=64, device=DEV)
    g = {k: tt(sc[k], grad=True) for k in KEYS}
    step = pipeline.Stage2Step(light, lut, gi, 2, metallic=True, fused=mode != "unfused", graphs=mode == "fused_graph")
    for ci in (0, 1, 0):
        for t in list(g.values()) + [light.base]:
            t.grad = None
        o = step(camts[ci], g, gt, vds[ci])
        torch.cuda.synchronize()
        print(mode, ci, float(o["loss"]), [float(g[k].grad.abs().max()) for k in ("albedo", "roughness", "metallic")],
              float(light.base.grad.abs().max()), flush=True)
