"""Times the densification pieces on the GPU.

    python tools/densify_bench.py [--P 300000] [--M 9]

  * gigs_densify_stats (every iteration) against the masked torch ops of scene/gaussian_model.py:933-945 + train.py:494-498
  * densify_and_prune end to end, and its gigs_gather_rows launch alone (algorithmic bytes: 8 B per moved float + 5 B per row
    of index / flag) against rebuilding the same thirty tensors with one torch index op each
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gi-gs_amd"))

import densify  # noqa: E402
import gigs_lib  # noqa: E402
import optim  # noqa: E402

HBM_PEAK = 8.0e12


def wall_ms(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=300000)
    ap.add_argument("--M", type=int, default=9)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    P, M = a.P, a.M
    shapes = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (M - 1, 3), "opacity": (1,), "normal": (3,), "albedo": (3,),
              "roughness": (1,), "metallic": (1,), "scaling": (3,), "rotation": (4,)}
    floats_per_row = sum(int(torch.tensor(s).prod()) for s in shapes.values())
    g = torch.Generator(device=dev).manual_seed(0)
    res = {"P": P, "M": M, "floats_per_gaussian": floats_per_row}

    # ---- statistics
    st = densify.DensifyState(P, dev)
    grad = torch.randn(P, 3, device=dev, generator=g) * 3e-4
    radii = (torch.rand(P, device=dev, generator=g) * 40 - 8).to(torch.int32)

    def torch_stats():
        vis = radii > 0
        st.max_radii2D[vis] = torch.max(st.max_radii2D[vis], radii[vis].float())
        st.xyz_gradient_accum[vis] += torch.norm(grad[vis, :2], dim=-1, keepdim=True)
        ab = torch.norm(grad[vis, :1].abs() + grad[vis, 1:2].abs(), dim=-1, keepdim=True)
        st.xyz_gradient_accum_abs[vis] += ab
        st.xyz_gradient_accum_abs_max[vis] = torch.max(st.xyz_gradient_accum_abs_max[vis], ab)
        st.denom[vis] += 1

    hip = lambda: densify.add_densification_stats(st, grad, radii)  # noqa: E731
    hip()
    torch.cuda.synchronize()
    with gigs_lib.profile() as prof:
        for _ in range(20):
            hip()
        torch.cuda.synchronize()
    k_ms = prof.stages["densify_stats"][0] / prof.stages["densify_stats"][1]
    vis_frac = float((radii > 0).float().mean())
    bytes_stats = P * 4 + vis_frac * P * (12 + 40)
    res["densify_stats"] = {"kernel_ms": round(k_ms, 4), "GBps": round(bytes_stats / k_ms / 1e6, 1),
                            "hbm_frac": round(bytes_stats / (k_ms * 1e-3) / HBM_PEAK, 4),
                            "hip_wall_ms": round(wall_ms(hip), 4), "torch_wall_ms": round(wall_ms(torch_stats), 4)}

    # ---- densify_and_prune
    def fresh():
        ps = {n: torch.nn.Parameter(torch.randn((P,) + s, device=dev, generator=g)) for n, s in shapes.items()}
        with torch.no_grad():
            ps["scaling"].copy_(torch.log(torch.rand(P, 3, device=dev, generator=g) * 0.08 + 1e-3))
            ps["opacity"].mul_(2.5)
        opt = optim.FusedAdam([{"params": [ps[n]], "lr": 1e-3, "name": n} for n in shapes], lr=0.0, eps=1e-15)
        for p in ps.values():
            p.grad = torch.zeros_like(p)
        opt.step()
        s2 = densify.DensifyState(P, dev)
        for _ in range(4):
            densify.add_densification_stats(s2, torch.randn(P, 3, device=dev, generator=g) * 3e-4, radii)
        return opt, s2

    walls, kernels, rows = [], [], []
    for _ in range(4):
        opt, s2 = fresh()
        torch.cuda.synchronize()
        with gigs_lib.profile() as prof:
            t0 = time.perf_counter()
            new, _ = densify.densify_and_prune(opt, s2, 2e-4, 0.05, 4.0, 20, generator=g)
            torch.cuda.synchronize()
            walls.append((time.perf_counter() - t0) * 1e3)
        kernels.append(prof.stages["gather_rows"][0])
        rows.append(int(new["xyz"].shape[0]))
    n_out = rows[-1]
    moved = 3 * floats_per_row * n_out * 8 + 30 * n_out * 5
    res["densify_and_prune"] = {"rows_out": n_out, "wall_ms": round(min(walls[1:]), 3),
                                "gather_rows_kernel_ms": round(min(kernels[1:]), 4),
                                "gather_GBps": round(moved / min(kernels[1:]) / 1e6, 1),
                                "gather_hbm_frac": round(moved / (min(kernels[1:]) * 1e-3) / HBM_PEAK, 4)}
    # the same data movement as thirty torch index ops
    opt, _ = fresh()
    tensors = []
    for grp in opt.param_groups:
        p = grp["params"][0]
        tensors += [p.detach(), opt.state[p]["exp_avg"], opt.state[p]["exp_avg_sq"]]
    src = torch.randint(0, P, (n_out,), device=dev, generator=g)
    res["densify_and_prune"]["torch_index_30_tensors_ms"] = round(wall_ms(lambda: [t[src] for t in tensors], iters=5), 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
