import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
importlib.import_module("gi-gs_amd")
import numpy as np, torch
import pipeline, scenes, pbr
dev = "cuda:0"
sc = scenes.surface_scene(P=300_000, sh_degree=2, seed=0)
g = {k: torch.from_numpy(sc[k]).to(dev).requires_grad_(True) for k in ["means3D", "opacities", "normal", "albedo", "roughness", "metallic", "shs", "scales", "rotations"]}
cams = [scenes.orbit_camera(i, 64, 800, 800, radius=3.5) for i in range(64)]
camts = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
gt = torch.rand(3, 800, 800, device=dev)
light = pbr.CubemapLight(base_res=256, device=dev)
lut = pbr.get_brdf_lut().to(dev)
rays = pipeline.canonical_rays(cams[0], dev)
vds = [pipeline.view_dirs_for(c, rays, dev) for c in camts]
step = pipeline.Stage2Step(light, lut, scenes.GI_DEFAULTS, 2, graphs=True, fused=True)
params = list(g.values()) + [light.base]
W = pipeline.WholeStepGraph
orig_sync = torch.cuda.Event.synchronize
marks = {}
def sync(self):
    t0 = time.perf_counter(); r = orig_sync(self); marks["sync_end"] = time.perf_counter(); marks.setdefault("sync_wait", []).append(marks["sync_end"] - t0); return r
torch.cuda.Event.synchronize = sync
orig_replay = torch.cuda.CUDAGraph.replay
def replay(self):
    t0 = time.perf_counter()
    if "sync_end" in marks and marks.get("next_is_fwd", True):
        marks.setdefault("host_gap", []).append(t0 - marks["sync_end"])
    marks["next_is_fwd"] = not marks.get("next_is_fwd", True)
    r = orig_replay(self); marks.setdefault("replay", []).append(time.perf_counter() - t0); return r
for i in range(8):
    for p in params: p.grad = None
    step(camts[i], g, gt, vds[i])
torch.cuda.synchronize()
torch.cuda.CUDAGraph.replay = replay
marks.clear()
t0 = time.perf_counter()
for i in range(8, 72):
    for p in params: p.grad = None
    step(camts[i % 64], g, gt, vds[i % 64])
torch.cuda.synchronize()
t1 = time.perf_counter()
f = lambda a: "mean %.0f us, max %.0f us" % (1e6 * np.mean(a), 1e6 * np.max(a))
print("step %.3f ms" % (1e3 * (t1 - t0) / 64))
print("host: end of the flag wait -> next forward launch:", f(marks["host_gap"][1:]))
print("graph launch call:", f(marks["replay"]))
print("flag wait:", f(marks["sync_wait"]))
# where does the event between the two graph launches land on the GPU's clock?
w = step.whole
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
torch.cuda.synchronize()
e0.record(); orig_replay(w.gf); e1.record(); orig_replay(w.gb); e2.record()
torch.cuda.synchronize()
print("forward graph %.3f ms, forward + backward %.3f ms (GPU event times)" % (e0.elapsed_time(e1), e0.elapsed_time(e2)))
e0.record(); orig_replay(w.gf); e1.record()
t0 = time.perf_counter(); orig_sync(e1); t1 = time.perf_counter()
print("host wait for the forward alone: %.3f ms" % (1e3 * (t1 - t0)))
orig_replay(w.gb); torch.cuda.synchronize()
ev = torch.cuda.Event()
e0.record(); orig_replay(w.gf); ev.record(); orig_replay(w.gb)
t0 = time.perf_counter(); orig_sync(ev); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host wait for the forward with the backward queued behind it: %.3f ms, then %.3f ms to the end" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
