"""CPU: the reference's PLY and checkpoint layouts (gi-gs_amd/scene_io.py).  Parity is pinned by the layout the
reference's own writer code defines (scene/gaussian_model.py:397-465; train.py:466-490); `plyfile` is absent here, so the
byte-level header is compared with the format plyfile documents for a float32 vertex element."""
import io
import os

import numpy as np
import pytest
import torch

import scene_io


def _params(P=37, K=9, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (K - 1, 3), "opacity": (1,), "normal": (3,), "albedo": (3,),
              "roughness": (1,), "metallic": (1,), "scaling": (3,), "rotation": (4,)}
    return {n: torch.randn((P,) + s, generator=g) for n, s in shapes.items()}


def test_ply_layout_and_round_trip(tmp_path):
    p = _params()
    path = os.path.join(tmp_path, "point_cloud", "iteration_7", "point_cloud.ply")
    scene_io.save_ply(path, p)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [l.split()[2] for l in lines[3:]]
    assert all(l.startswith("property float ") for l in lines[3:])
    assert names == (["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(24)] + ["opacity"] +
                     ["normal_0", "normal_1", "normal_2", "albedo_0", "albedo_1", "albedo_2", "roughness", "metallic",
                      "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])
    table = np.frombuffer(body, dtype="<f4").reshape(37, len(names))
    # channel-major SH: f_rest_k = features_rest[:, k % 8, k // 8] (the transpose(1, 2) at :430-437)
    assert np.array_equal(table[:, names.index("f_rest_9")], p["f_rest"][:, 1, 1].numpy())
    assert np.array_equal(table[:, names.index("f_dc_2")], p["f_dc"][:, 0, 2].numpy())
    assert np.array_equal(table[:, names.index("rot_3")], p["rotation"][:, 3].numpy())
    back = scene_io.load_ply(path, max_sh_degree=2)
    for n in scene_io.NAMES:
        assert back[n].shape == p[n].shape and torch.equal(back[n], p[n]), n
    with pytest.raises(ValueError, match="f_rest_"):
        scene_io.load_ply(path, max_sh_degree=3)  # the reference asserts the SH count too (:517)


def test_ply_reader_handles_ascii_and_extra_elements(tmp_path):
    path = os.path.join(tmp_path, "a.ply")
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\nproperty double y\n"
                "property uchar z\nend_header\n1.5 2.5 3\n-1 0.25 255\n")
    v = scene_io.read_ply_vertices(path)
    assert v["x"].tolist() == [1.5, -1.0] and v["y"].tolist() == [2.5, 0.25] and v["z"].tolist() == [3, 255]
    with open(path, "w") as f:
        f.write("ply\nformat binary_big_endian 1.0\nelement vertex 0\nproperty float x\nend_header\n")
    with pytest.raises(ValueError, match="unsupported PLY format"):
        scene_io.read_ply_vertices(path)


def test_checkpoint_round_trip_keeps_the_reference_tuple(tmp_path):
    import types
    p = {n: torch.nn.Parameter(t) for n, t in _params(P=11).items()}
    lrs = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=1.25e-4, opacity=0.05, normal=0.05, albedo=0.05, roughness=0.05,
               metallic=0.05, scaling=5e-3, rotation=1e-3)
    opt = torch.optim.Adam([{"params": [p[n]], "lr": lrs[n], "name": n} for n in scene_io.NAMES], lr=0.0, eps=1e-15)
    for t in p.values():
        t.grad = torch.ones_like(t)
    opt.step()
    stats = types.SimpleNamespace(max_radii2D=torch.arange(11.0), xyz_gradient_accum=torch.rand(11, 1),
                                  xyz_gradient_accum_abs=torch.rand(11, 1), xyz_gradient_accum_abs_max=torch.rand(11, 1),
                                  denom=torch.ones(11, 1))
    cap = scene_io.capture(2, p, stats, opt, 3.5)
    assert len(cap) == 18 and cap[0] == 2 and cap[4] is p["scaling"] and cap[6] is p["opacity"] and cap[17] == 3.5
    cube = {"base": torch.rand(6, 4, 4, 3)}
    path = os.path.join(tmp_path, "chkpnt30000.pth")
    scene_io.save_checkpoint(path, cap, cube, {"state": {}, "param_groups": []}, 30000)
    ck = scene_io.load_checkpoint(path)
    assert ck["iteration"] == 30000 and torch.equal(ck["cubemap"]["base"], cube["base"])
    deg, params, st, opt_dict, scale = scene_io.restore(ck["gaussians"])
    assert deg == 2 and scale == 3.5 and torch.equal(st["max_radii2D"], torch.arange(11.0))
    for n in scene_io.NAMES:
        assert torch.equal(params[n], p[n].detach())
    # the optimizer state loads into a fresh optimizer over the restored tensors (GaussianModel.restore :170-176)
    q = {n: torch.nn.Parameter(params[n].clone()) for n in scene_io.NAMES}
    opt2 = torch.optim.Adam([{"params": [q[n]], "lr": 0.0, "name": n} for n in scene_io.NAMES], lr=0.0, eps=1e-15)
    opt2.load_state_dict(opt_dict)
    assert [g["name"] for g in opt2.param_groups] == scene_io.NAMES and opt2.param_groups[0]["lr"] == 1.6e-4
    assert torch.equal(opt2.state[q["xyz"]]["exp_avg"], opt.state[p["xyz"]]["exp_avg"])
    with pytest.raises(ValueError, match="18"):
        scene_io.restore(cap[:5])
    torch.save({"something": 1}, path)
    with pytest.raises(ValueError, match="not a GI-GS checkpoint"):
        scene_io.load_checkpoint(path)


def test_fused_adam_checkpoint_resumes_under_torch_adam(tmp_path):
    """A chkpntN.pth written while training with FusedAdam must load into the reference's torch.optim.Adam
    (GaussianModel.restore, scene/gaussian_model.py:151-176) AND step there: the saved param_groups carry every
    key torch's Adam reads (weight_decay, amsgrad, maximize, ...)."""
    import types
    import optim
    p = {n: torch.nn.Parameter(t) for n, t in _params(P=9).items()}
    lrs = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=1.25e-4, opacity=0.05, normal=0.05, albedo=0.05, roughness=0.05,
               metallic=0.05, scaling=5e-3, rotation=1e-3)
    opt = optim.FusedAdam([{"params": [p[n]], "lr": lrs[n], "name": n} for n in scene_io.NAMES], lr=0.0, eps=1e-15)
    ref_keys = set(torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))]).param_groups[0].keys())
    assert ref_keys <= set(opt.param_groups[0].keys()), ref_keys - set(opt.param_groups[0].keys())
    # the state FusedAdam.step() leaves behind after two updates (the step itself needs a GPU: tests/test_gpu_losses.py)
    for t in p.values():
        opt.state[t] = {"step": torch.tensor(2.0), "exp_avg": torch.full_like(t, 0.01), "exp_avg_sq": torch.full_like(t, 1e-4)}
    stats = types.SimpleNamespace(max_radii2D=torch.zeros(9), xyz_gradient_accum=torch.zeros(9, 1),
                                  xyz_gradient_accum_abs=torch.zeros(9, 1), xyz_gradient_accum_abs_max=torch.zeros(9, 1),
                                  denom=torch.zeros(9, 1))
    path = os.path.join(tmp_path, "chkpnt7000.pth")
    scene_io.save_checkpoint(path, scene_io.capture(2, p, stats, opt, 1.0), {"base": torch.rand(6, 4, 4, 3)},
                             {"state": {}, "param_groups": []}, 7000)
    _, params, _, opt_dict, _ = scene_io.restore(scene_io.load_checkpoint(path)["gaussians"])
    q = {n: torch.nn.Parameter(params[n].clone()) for n in scene_io.NAMES}
    ref = torch.optim.Adam([{"params": [q[n]], "lr": 0.0, "name": n} for n in scene_io.NAMES], lr=0.0, eps=1e-15)
    ref.load_state_dict(opt_dict)
    before = q["xyz"].detach().clone()
    for t in q.values():
        t.grad = torch.ones_like(t)
    ref.step()  # KeyError 'weight_decay' before the inert keys were saved
    assert float(ref.state[q["xyz"]]["step"]) == 3.0 and not torch.equal(q["xyz"].detach(), before)
    # and the other direction still works; non-default values of the inert keys are refused, not ignored
    back = optim.FusedAdam([{"params": [q[n]], "lr": 0.0, "name": n} for n in scene_io.NAMES], lr=0.0, eps=1e-15)
    back.load_state_dict(ref.state_dict())
    assert back.param_groups[0]["lr"] == 1.6e-4
    with pytest.raises(ValueError):
        optim.FusedAdam([torch.nn.Parameter(torch.zeros(1))], weight_decay=0.1)
    sd = ref.state_dict()
    sd["param_groups"][0]["amsgrad"] = True
    with pytest.raises(ValueError):
        back.load_state_dict(sd)
