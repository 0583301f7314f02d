"""CPU, world_size 2, gloo: the view-parallel gradient / statistics reduction (gi-gs_amd/dp.py) -- the flat
gradient slab with the asynchronous (communication-side) all-reduce, gradient-less parameters, densification
statistics reduced as statistics, and replicated densification decisions."""
import os
import socket

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
    from oracle import densify_ref
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    P = 257
    params = [torch.zeros(P, 3, requires_grad=True), torch.zeros(P, 9, 3, requires_grad=True),
              torch.zeros(P, 1, requires_grad=True), torch.zeros(P, 4, requires_grad=True),
              torch.zeros(P, 2, requires_grad=True)]
    grads = [torch.randn(p.shape, generator=g) for p in params]
    for p, gr in zip(params[:3], grads[:3]):
        p.grad = gr.clone()
    # params[3] has no gradient on any rank; params[4] has one on rank 1 only
    if rank == 1:
        params[4].grad = grads[4].clone()
    flat = dp.allreduce_gradients(params)  # convenience path: presence agreed across ranks
    assert flat.numel() == sum(p.numel() for p in params)
    conv = [None if p.grad is None else p.grad.clone() for p in params]

    # the persistent slab, asynchronous: gradients born in the slab (views), one stray tensor, one absent parameter
    slab = dp.GradSlab(params[:4])
    sink = slab.sink(["a", "b", "c", "d"])
    assert sink["b"].data_ptr() == slab.views[1].data_ptr()
    out = []
    for step in range(2):
        for p in params:
            p.grad = None
        local = [torch.randn(p.shape, generator=g) for p in params[:3]]
        sink["a"].copy_(local[0]); params[0].grad = sink["a"]          # written in place (rasterizer backward + grad_sink)
        sink["b"].copy_(local[1]); params[1].grad = sink["b"]
        params[2].grad = local[2].clone()                                # a stray tensor: copied into the slab once
        slab.allreduce_async()
        busy = torch.randn(64, 64, generator=g) @ torch.randn(64, 64, generator=g)  # work beside the collective
        slab.wait()
        assert params[0].grad.data_ptr() == slab.views[0].data_ptr() and params[2].grad.data_ptr() == slab.views[2].data_ptr()
        out.append(dict(local=local, reduced=[None if p.grad is None else p.grad.clone() for p in params[:4]], busy=float(busy.sum())))

    # stage-2 pattern: only parameters 1 and 2 (adjacent: one collective) and 4 carry gradients, the others are exact zeros
    slab5 = dp.GradSlab(params)
    for p in params:
        p.grad = None
    slab5.attach()
    part = [torch.randn(p.shape, generator=g) for p in params]
    for i in (1, 2, 4):
        params[i].grad.copy_(part[i])
    assert len(slab5._ranges([1, 2, 4])) == 2
    slab5.allreduce_async(only=[1, 2, 4], check_rest_zero=True)
    slab5.wait()
    only_out = dict(local=part, reduced=[p.grad.clone() for p in params])
    params[0].grad.fill_(1.0)  # the premise is violated: the check notices
    try:
        slab5.allreduce_async(only=[1, 2, 4], check_rest_zero=True)
        only_caught = False
        slab5.wait()
    except RuntimeError:
        only_caught = True

    # packed: the slab already holds the step's gradients (a captured backward wrote the views) and `.grad` is not in play:
    # nothing is gathered before the collective (a parameter whose .grad is None must NOT have its view zeroed) and nothing
    # is assigned afterwards (train_iteration.Stage2Trainer.data_parallel on the graph path)
    slabp = dp.GradSlab(params[:3])
    for p in params:
        p.grad = None
    pk = [torch.randn(p.shape, generator=g) for p in params[:3]]
    for v, t in zip(slabp.views, pk):
        v.copy_(t)
    slabp.allreduce_async(only=[1, 2], packed=True)
    slabp.wait()
    packed_out = dict(local=pk, reduced=[v.clone() for v in slabp.views], grads_none=all(p.grad is None for p in params[:3]))

    # densification replaces the parameter OBJECTS (cat_tensors_to_optimizer, scene/gaussian_model.py:669-706): the slab is
    # rebuilt from the current list and reduces the grown tensors
    grown = [torch.zeros(P + 5 + rank * 0, 3, requires_grad=True), params[1]]
    slab2 = dp.GradSlab(params[:2])
    assert slab2.rebuild(params[:2]) is False
    assert slab2.rebuild(grown) is True and slab2.views[0].shape == (P + 5, 3) and slab2.flat.numel() == (P + 5) * 3 + params[1].numel()
    gg = [torch.randn(p.shape, generator=g) for p in grown]
    for p, t in zip(grown, gg):
        p.grad = t.clone()
    slab2.allreduce_async()
    slab2.wait()
    rebuilt = dict(local=gg, reduced=[p.grad.clone() for p in grown])
    # attach(skip=sink): sinked parameters keep .grad = None, the rest alias the slab
    sk = slab2.sink(["a"])
    slab2.attach(skip=sk)
    assert grown[0].grad is None and grown[1].grad.data_ptr() == slab2.views[1].data_ptr()

    # statistics
    vg = torch.randn(P, 3, generator=g)
    radii = torch.randint(0, 30, (P,), generator=g)
    st = dp.per_view_densification_stats(vg, radii)
    dp.reduce_densification_stats(st["xyz_gradient_accum"], st["xyz_gradient_accum_abs"], st["denom"], st["max_radii2D"],
                                  st["xyz_gradient_accum_abs_max"])
    # replicated densification: identical reduced statistics + identically seeded draws -> identical decisions
    gm = torch.Generator().manual_seed(7)  # the same on every rank
    prm = {n: torch.randn(P, *s, generator=gm) for n, s in
           zip(densify_ref.NAMES, [(3,), (1, 3), (3, 3), (1,), (3,), (3,), (1,), (1,), (3,), (4,)])}
    prm["scaling"] = prm["scaling"] * 0.3 - 3.0
    stats = dict(accum=st["xyz_gradient_accum"].clone(), accum_abs=st["xyz_gradient_accum_abs"].clone(),
                 accum_abs_max=st["xyz_gradient_accum_abs_max"].clone(), denom=st["denom"].clone(),
                 max_radii2D=st["max_radii2D"].clone())
    model = dict(params=prm, exp_avg={n: torch.zeros_like(t) for n, t in prm.items()},
                 exp_avg_sq={n: torch.zeros_like(t) for n, t in prm.items()}, stats=stats)
    z_clone, z_split = torch.randn(P, 3, generator=gm), torch.randn(4 * P, 3, generator=gm)
    densify_ref.densify_and_prune(model, 0.6, 0.005, 2.0, None, z_clone, z_split)
    dens = {n: model["params"][n].clone() for n in densify_ref.NAMES}
    assert dens["xyz"].shape[0] != P  # something was cloned / split / pruned
    views = [dp.view_for(s, rank, world, 7) for s in range(5)]
    torch.save(dict(conv=conv, local=grads, slab=out, only=only_out, packed=packed_out, rebuilt=rebuilt, only_caught=only_caught, vg=vg, radii=radii, stats=st, views=views, dens=dens),
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_view_parallel_reduction_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"r{k}.pt")) for k in range(world)]
    # every rank holds the same, summed gradients
    for i in range(3):
        want = r[0]["local"][i] + r[1]["local"][i]
        for k in range(world):
            torch.testing.assert_close(r[k]["conv"][i], want)
    for k in range(world):
        assert r[k]["conv"][3] is None            # no gradient anywhere: stays None (Adam must skip it, as on one GPU)
        torch.testing.assert_close(r[k]["conv"][4], r[1]["local"][4])  # a gradient on one rank only: the other adds zeros
    # the slab path, two consecutive steps
    for step in range(2):
        for i in range(3):
            want = r[0]["slab"][step]["local"][i] + r[1]["slab"][step]["local"][i]
            for k in range(world):
                torch.testing.assert_close(r[k]["slab"][step]["reduced"][i], want)
        assert r[0]["slab"][step]["reduced"][3] is None and r[1]["slab"][step]["reduced"][3] is None
    # partial reduction (stage-2 pattern): the listed parameters are summed, the others stay the exact zeros they were
    for i in range(5):
        for k in range(world):
            want = r[0]["only"]["local"][i] + r[1]["only"]["local"][i] if i in (1, 2, 4) else torch.zeros_like(r[0]["only"]["local"][i])
            torch.testing.assert_close(r[k]["only"]["reduced"][i], want)
    assert r[0]["only_caught"] and r[1]["only_caught"]
    for k in range(world):  # packed: the reduced stretch is summed in place, the rest of the slab and every .grad untouched
        assert r[k]["packed"]["grads_none"]
        torch.testing.assert_close(r[k]["packed"]["reduced"][0], r[k]["packed"]["local"][0])
        for i in (1, 2):
            torch.testing.assert_close(r[k]["packed"]["reduced"][i], r[0]["packed"]["local"][i] + r[1]["packed"]["local"][i])
    for i in range(2):  # the slab rebuilt around a grown parameter
        for k in range(world):
            torch.testing.assert_close(r[k]["rebuilt"]["reduced"][i], r[0]["rebuilt"]["local"][i] + r[1]["rebuilt"]["local"][i])
    # statistics: sums of per-view norms (NOT the norm of the summed gradient), max of radii and of the abs term
    vis = [x["radii"] > 0 for x in r]
    zero = torch.zeros(1)
    want_accum = sum(torch.where(v[:, None], torch.norm(x["vg"][:, :2], dim=-1, keepdim=True), zero) for v, x in zip(vis, r))
    abs_xy = [torch.where(v[:, None], x["vg"][:, 0:1].abs() + x["vg"][:, 1:2].abs(), zero) for v, x in zip(vis, r)]
    want_denom = sum(v[:, None].float() for v in vis)
    want_max = torch.maximum(r[0]["radii"], r[1]["radii"]).float()
    for k in range(world):
        torch.testing.assert_close(r[k]["stats"]["xyz_gradient_accum"], want_accum)
        torch.testing.assert_close(r[k]["stats"]["xyz_gradient_accum_abs"], abs_xy[0] + abs_xy[1])   # gaussian_model.py:941-943
        torch.testing.assert_close(r[k]["stats"]["xyz_gradient_accum_abs_max"], torch.maximum(abs_xy[0], abs_xy[1]))  # :944
        torch.testing.assert_close(r[k]["stats"]["denom"], want_denom)
        torch.testing.assert_close(r[k]["stats"]["max_radii2D"], want_max)
    summed_norm = torch.norm((r[0]["vg"] + r[1]["vg"])[:, :2], dim=-1, keepdim=True)
    assert not torch.allclose(want_accum, summed_norm)
    # replicated densification decisions are identical across ranks
    for n in r[0]["dens"]:
        assert torch.equal(r[0]["dens"][n], r[1]["dens"][n]), n
    # views are disjoint across ranks within a step and cover the set round-robin
    assert r[0]["views"] == [0, 2, 4, 6, 1] and r[1]["views"] == [1, 3, 5, 0, 2]
