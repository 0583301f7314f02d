import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gi-gs_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import scenes, pipeline, pbr
from stage2_fused import Stage2FusedBack
DEV = "cuda:0"
tt = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).to(DEV).requires_grad_(grad)
sc = scenes.surface_scene(P=10_000, sh_degree=2, seed=9, scale_mu=0.025)
gi = scenes.GI_DEFAULTS
H, W = 128, 160
cam = scenes.orbit_camera(1, 6, W, H, radius=3.5)
camt = {k: (tt(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
torch.manual_seed(1)
gt = torch.rand(3, H, W, device=DEV) * 0.5
lut = pbr.get_brdf_lut().to(DEV)
rays = pipeline.canonical_rays(cam, DEV)
vd = pipeline.view_dirs_for(camt, rays, DEV)
KEYS = ["means3D", "opacities", "normal", "shs", "albedo", "roughness", "metallic", "scales", "rotations"]
res = {}
for mode in ("eager", "graph"):
    torch.manual_seed(2)
    light = pbr.CubemapLight(base_res=64, device=DEV)
    g = {k: tt(sc[k], grad=True) for k in KEYS}
    out, pts, st = pipeline.rasterize(camt, g, 2, torch.zeros(3, device=DEV), gi)
    (_, radii, _, _, nfd, normal_map, occ, albedo_map, roughness_map, metallic_map, onv, depth_pos) = out
    cfg = dict(H=H, W=W, gi=gi, focal_x=W / (2.0 * cam["tanfovx"]), focal_y=H / (2.0 * cam["tanfovy"]), metallic=True,
               indirect=True, gamma=False, tone=False)
    back = Stage2FusedBack(lut, cfg)
    leaves = [albedo_map.detach().clone().requires_grad_(True), roughness_map.detach().clone().requires_grad_(True),
              metallic_map.detach().clone().requires_grad_(True)]
    args = (normal_map.detach(), onv.detach(), leaves[0], leaves[1], leaves[2], occ.detach(), depth_pos.detach(),
            st.viewmatrix, vd, gt)
    if mode == "graph":
        with torch.no_grad():
            back(*args)
        sample = tuple(a.detach().clone().requires_grad_(a.requires_grad) for a in args)
        back = torch.cuda.make_graphed_callables(back, sample, allow_unused_input=True)
    for it in range(2):
        for l in leaves + [light.base]:
            l.grad = None
        loss, rr, rd, irr = back(*args)
        loss.backward()
        torch.cuda.synchronize()
        print(mode, it, float(loss), [float(l.grad.abs().max()) for l in leaves], float(light.base.grad.abs().max()))
    res[mode] = [l.grad.clone() for l in leaves] + [light.base.grad.clone()]
for a, b, n in zip(res["eager"], res["graph"], ("albedo", "rough", "metal", "base")):
    print(n, float((a - b).abs().max()), float(a.abs().max()))
