#!/usr/bin/env python
"""Two-rank view-parallel TRAINING on real kernels: does N-rank training take the update a single process takes
on the sum of the ranks' per-view gradients?

    python tools/dp_training_check.py            # parent: starts 2 ranks (gloo, both on the visible GPU) -> one JSON line

Every rank holds the same perturbed scene and light, runs `train_iteration.Stage2Trainer(...).data_parallel()` for a few
complete stage-2 iterations (three hipGraphs each; the all-reduce sits between the backward and the captured Adam), rank r on
view 2 i + r.  Checked: (a) the ranks' parameters stay bit-identical; (b) they equal, to fp32 atomics noise, a single-process
run that accumulates both views' gradients (eager loss.backward() twice) and steps FusedAdam once per iteration.
The parent process never touches the GPU (it only launches the ranks: no exec from a GPU-initialised process)."""
import importlib
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITERS, WORLD = 5, 2


def rank_main():
    sys.path.insert(0, ROOT)
    importlib.import_module("gi-gs_amd")
    import numpy as np
    import torch
    import torch.distributed as dist

    import activations
    import optim
    import pbr
    import pipeline
    import scenes
    import train_iteration as ti
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    H = W = 96
    deg = 1
    sc = scenes.surface_scene(P=4000, sh_degree=deg, seed=6, scale_mu=0.04)
    n_views = 2 * ITERS
    cams = [scenes.orbit_camera(i, n_views, W, H, radius=3.5) for i in range(n_views)]
    cams = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    gi = scenes.GI_DEFAULTS
    lut = pbr.get_brdf_lut().to(dev)
    rays = pipeline.canonical_rays(cams[0], dev)
    vds = [pipeline.view_dirs_for(c, rays, dev) for c in cams]
    gen = torch.Generator(device=dev).manual_seed(3)
    targets = [torch.rand(3, H, W, device=dev, generator=gen) * 0.6 for _ in range(n_views)]

    def fresh():
        torch.manual_seed(11)
        light = pbr.CubemapLight(base_res=64, device=dev)
        raw = ti.raw_from_scene(sc, dev)
        g2 = torch.Generator(device=dev).manual_seed(5)
        with torch.no_grad():
            for k in ("albedo", "roughness", "metallic"):
                raw[k].add_(torch.randn(raw[k].shape, device=dev, generator=g2) * 0.5)
        return raw, light

    # one step on its own first: the graphs are captured before the process group exists (as bench.py does)
    raw, light = fresh()
    tr = ti.Stage2Trainer(raw, light, lut, gi, deg, graphs=True)
    tr.data_parallel()
    dist_ready = [False]
    hook = tr.stepper.before_update
    tr.stepper.before_update = lambda: hook() if dist_ready[0] else None
    snapshot = ({k: v.detach().clone() for k, v in raw.items()}, light.base.detach().clone(),
                {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for o in (tr.optimizer, tr.light_optimizer)
                 for p, st in o.state.items()})
    tr.iteration(cams[rank], targets[rank], vds[rank])  # capture (its update is undone below)
    torch.cuda.synchronize()
    with torch.no_grad():
        for k, v in raw.items():
            v.copy_(snapshot[0][k])
        light.base.copy_(snapshot[1])
        for o in (tr.optimizer, tr.light_optimizer):
            for p, st in o.state.items():
                st["step"].zero_(); st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
    dist.init_process_group("gloo")
    dist_ready[0] = True
    losses = []
    for i in range(ITERS):
        v = world * i + rank
        losses.append(float(tr.iteration(cams[v], targets[v], vds[v])["loss"]))
    torch.cuda.synchronize()
    assert tr.stepper.whole is not None and tr.stepper.whole.go is not None
    mine = torch.cat([raw[k].detach().reshape(-1) for k in ("albedo", "roughness", "metallic")] + [light.base.detach().reshape(-1)])
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    ranks_equal = all(torch.equal(gathered[0], g) for g in gathered[1:])
    out = None
    if rank == 0:
        # single process: both views' gradients accumulated by two eager backward passes, one FusedAdam step per iteration
        raw1, light1 = fresh()
        step = pipeline.Stage2Step(light1, lut, gi, deg, fused=True, graphs=False, prepare=activations.activate,
                                   regularizer=ti.Stage2Regularizer(light1))
        opt = optim.FusedAdam([{"params": [raw1[k]], "lr": ti.DEFAULT_LRS[k], "name": k} for k in ti.RAW_KEYS], lr=0.0, eps=1e-15)
        lopt = optim.FusedAdam([{"params": [light1.base], "lr": 0.05}], lr=0.05)
        for i in range(ITERS):
            for r in range(world):
                v = world * i + r
                step(cams[v], raw1, targets[v], vds[v])   # loss.backward(): .grad accumulates over the two views
            opt.step(); lopt.step()
            for p in list(raw1.values()) + [light1.base]:
                p.grad = None
            with torch.no_grad():
                light1.clamp_(min=0.0)
        torch.cuda.synchronize()
        ref = torch.cat([raw1[k].detach().reshape(-1) for k in ("albedo", "roughness", "metallic")] + [light1.base.detach().reshape(-1)])
        start = torch.cat([snapshot[0][k].reshape(-1) for k in ("albedo", "roughness", "metallic")] + [snapshot[1].reshape(-1)])
        moved = float((ref - start).abs().max())
        diff = float((gathered[0] - ref).abs().max())
        geometry_untouched = all(torch.equal(raw[k].detach(), snapshot[0][k]) for k in ("xyz", "scaling", "rotation", "opacity", "f_dc"))
        out = dict(world=world, backend="gloo (both ranks on one GPU)", iterations=ITERS, ranks_bit_identical=bool(ranks_equal),
                   max_abs_diff_vs_single_process=diff, max_parameter_movement=moved, relative=diff / max(moved, 1e-30),
                   geometry_untouched=bool(geometry_untouched), losses_rank0=[round(x, 6) for x in losses],
                   formulation="Stage2Trainer.data_parallel(): 3 hipGraphs per iteration, all-reduce between backward and captured Adam")
        ok = ranks_equal and geometry_untouched and diff <= 2e-3 * max(moved, 1e-6)
        out["ok"] = bool(ok)
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if out is not None and not out["ok"]:
        sys.exit(1)


def main():
    if "RANK" in os.environ:
        return rank_main()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % WORLD, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)]
    sys.exit(subprocess.call(cmd, env=env))


if __name__ == "__main__":
    main()
