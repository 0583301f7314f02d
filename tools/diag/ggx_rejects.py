"""How much of the cached GGX pair-weight tables is rejected candidates (weight -1: inside the window's rectangles, outside the cone)?"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); importlib.import_module("gi-gs_amd")
import torch
import pbr
import pbr.renderutils.ops as ops
light = pbr.CubemapLight(base_res=256).to("cuda:0")
light.build_mips()
torch.cuda.synchronize()
tot = rej = 0
for key, t in ops._weightTables.items():
    if t is None:
        continue
    w = t[1]
    n, r = w.numel(), int((w < 0).sum())
    z = int((w == 0).sum())
    print("res %4d rough %.3f: %11d candidates, %5.1f %% rejected, %5.1f %% exact zeros, %.1f per texel" % (key[0], key[1], n, 100.0 * r / n, 100.0 * z / n, n / (6 * key[0] ** 2)))
    tot += n; rej += r
print("all levels: %d candidates (%.2f GB fp32), %.1f %% rejected" % (tot, tot * 4 / 1e9, 100.0 * rej / tot))
