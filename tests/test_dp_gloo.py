"""CPU, world_size 2, gloo: the view-parallel gradient / statistics reduction (gi-gs_amd/dp.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    importlib.import_module("gi-gs_amd")
    import dp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    P = 257
    params = [torch.zeros(P, 3, requires_grad=True), torch.zeros(P, 9, 3, requires_grad=True),
              torch.zeros(P, 1, requires_grad=True), torch.zeros(P, 4, requires_grad=True)]
    grads = [torch.randn(p.shape, generator=g) for p in params]
    for p, gr in zip(params[:-1], grads[:-1]):
        p.grad = gr.clone()  # the last parameter has no gradient on purpose
    flat = dp.allreduce_gradients(params)
    assert flat.numel() == sum(p.numel() for p in params)
    # statistics
    vg = torch.randn(P, 3, generator=g)
    radii = torch.randint(0, 30, (P,), generator=g)
    st = dp.per_view_densification_stats(vg, radii)
    dp.reduce_densification_stats(st["xyz_gradient_accum"], st["xyz_gradient_accum_abs"], st["denom"], st["max_radii2D"])
    views = [dp.view_for(s, rank, world, 7) for s in range(5)]
    torch.save(dict(grads=[p.grad for p in params], local=grads, vg=vg, radii=radii, stats=st, views=views),
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_view_parallel_reduction_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"r{k}.pt")) for k in range(world)]
    # every rank holds the same, summed gradients; the gradient-less parameter reduces to zeros
    for i in range(3):
        want = r[0]["local"][i] + r[1]["local"][i]
        for k in range(world):
            torch.testing.assert_close(r[k]["grads"][i], want)
    assert float(r[0]["grads"][3].abs().sum()) == 0.0
    # statistics: sums of per-view norms (NOT the norm of the summed gradient), max of radii
    vis = [x["radii"] > 0 for x in r]
    want_accum = sum(torch.where(v[:, None], torch.norm(x["vg"][:, :2], dim=-1, keepdim=True), torch.zeros(1)) for v, x in zip(vis, r))
    want_denom = sum(v[:, None].float() for v in vis)
    want_max = torch.maximum(r[0]["radii"], r[1]["radii"]).float()
    for k in range(world):
        torch.testing.assert_close(r[k]["stats"]["xyz_gradient_accum"], want_accum)
        torch.testing.assert_close(r[k]["stats"]["denom"], want_denom)
        torch.testing.assert_close(r[k]["stats"]["max_radii2D"], want_max)
    summed_norm = torch.norm((r[0]["vg"] + r[1]["vg"])[:, :2], dim=-1, keepdim=True)
    assert not torch.allclose(want_accum, summed_norm)
    # views are disjoint across ranks within a step and cover the set round-robin
    assert r[0]["views"] == [0, 2, 4, 6, 1] and r[1]["views"] == [1, 3, 5, 0, 2]
