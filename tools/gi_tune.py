"""Times gigs_ssao / gigs_ssr alone on the C2 stand-in G-buffer for a list of env settings (tuning aid).
usage: python tools/gi_tune.py KEY=VAL[,KEY=VAL] ...   (each argument is one configuration; '-' = defaults)"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gi-gs_amd"))
import torch
import scenes, pipeline, gigs_lib
import diff_gaussian_rasterization as dgr

dev = "cuda:0"
sc = scenes.surface_scene(P=300_000, sh_degree=2, seed=0)
cam = scenes.orbit_camera(0, 8, 800, 800)
gi = dict(scenes.GI_DEFAULTS)
g = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if hasattr(v, "dtype")}
out, _, st = pipeline.rasterize(cam, g, 2, torch.zeros(3, device=dev), gi)
onv_raw, depth_pos = out[10], out[11]
W = H = 800
fx, fy = W / (2 * cam["tanfovx"]), H / (2 * cam["tanfovy"])
C = dgr._C
rgb = torch.rand(3, H, W, device=dev); alb = torch.rand(3, H, W, device=dev)
rough = torch.rand(1, H, W, device=dev); met = torch.rand(1, H, W, device=dev); F0 = torch.rand(3, H, W, device=dev)
ref = None
for cfg in sys.argv[1:] or ["-"]:
    keys = []
    if cfg != "-":
        for kv in cfg.split(","):
            k, v = kv.split("="); os.environ[k] = v; keys.append(k)
    def run():
        occ = C.SSAO(W, H, fx, fy, gi["radius"], gi["bias"], gi["thick"], gi["delta"], gi["step"], gi["start"], onv_raw, depth_pos)
        col, abd = C.SSR(W, H, fx, fy, gi["radius"], gi["bias"], gi["thick"], gi["delta"], gi["step"], gi["start"],
                         onv_raw, depth_pos, rgb, alb, rough, met, F0)
        return occ, col
    for _ in range(2): res = run()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(5): res = run()
    e[1].record(); torch.cuda.synchronize()
    ms = e[0].elapsed_time(e[1]) / 5
    if ref is None: ref = [r.clone() for r in res]
    same = all(torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)) for a, b in zip(ref, res))
    print(json.dumps({"cfg": cfg, "ssao+ssr_ms": round(ms, 4), "bit_identical_to_first": same}), flush=True)
    for k in keys: os.environ.pop(k, None)
