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


def test_bench_real_scene_hook_reads_a_saved_ply(tmp_path):
    """bench.py --scene / --cameras (real data if present on the box): a point_cloud.ply written by save_ply
    and a transforms json come back as the post-activation arrays of the getters (scene/gaussian_model.py:178-263) and
    as camera dicts at the bench resolution; the same through a chkpntN.pth."""
    import argparse
    import importlib
    import json
    import sys

    import numpy as np
    import torch

    import activations
    import densify
    import scene_io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    torch.manual_seed(0)
    P, deg = 500, 2
    K = (deg + 1) ** 2
    raw = dict(xyz=torch.randn(P, 3), f_dc=torch.randn(P, 1, 3), f_rest=torch.randn(P, K - 1, 3) * 0.1, opacity=torch.randn(P, 1),
               normal=torch.randn(P, 3), albedo=torch.randn(P, 3), roughness=torch.randn(P, 1), metallic=torch.randn(P, 1),
               scaling=torch.randn(P, 3) * 0.3 - 3.0, rotation=torch.randn(P, 4))
    ply = str(tmp_path / "point_cloud.ply")
    scene_io.save_ply(ply, raw)
    assert scene_io.ply_sh_degree(ply) == deg
    # poses: two frames of a NeRF-synthetic style json
    c2w = np.eye(4)
    c2w[:3, 3] = [0.0, -4.0, 0.5]
    tf = str(tmp_path / "transforms_train.json")
    with open(tf, "w") as f:
        json.dump({"camera_angle_x": 0.6911, "frames": [{"file_path": "./train/r_0", "transform_matrix": c2w.tolist()},
                                                        {"file_path": "./train/r_1", "transform_matrix": c2w.tolist()}]}, f)
    args = argparse.Namespace(scene=ply, cameras=tf, gaussians=None, sh_degree=None)
    sc, cams, tag = bench.load_workload(args, 96, 64)
    assert args.gaussians == P and args.sh_degree == deg and ply in tag and len(cams) == 2
    want = activations.activate_torch(raw)
    for k in ("means3D", "shs", "opacities", "normal", "albedo", "roughness", "metallic", "scales", "rotations"):
        np.testing.assert_allclose(sc[k], want[k].numpy(), rtol=1e-6, atol=1e-7, err_msg=k)
    assert sc["shs"].shape == (P, K, 3)
    assert cams[0]["image_width"] == 96 and cams[0]["image_height"] == 64 and cams[0]["viewmatrix"].shape == (4, 4)
    assert abs(cams[0]["tanfovx"] - np.tan(0.6911 / 2)) < 1e-6 and np.allclose(cams[0]["campos"], [0.0, -4.0, 0.5], atol=1e-6)
    # without --cameras: 64 orbit views around the cloud
    args2 = argparse.Namespace(scene=ply, cameras=None, gaussians=None, sh_degree=None)
    _, cams2, tag2 = bench.load_workload(args2, 80, 80)
    assert len(cams2) == 64 and "orbit" in tag2 and "eye_target" in cams2[0]
    # the same scene from a checkpoint
    params = {k: torch.nn.Parameter(v.clone()) for k, v in raw.items()}
    opt = torch.optim.Adam([{"params": [params[k]], "lr": 1e-3, "name": k} for k in scene_io.NAMES], lr=0.0, eps=1e-15)
    ck = str(tmp_path / "chkpnt30000.pth")
    scene_io.save_checkpoint(ck, scene_io.capture(deg, params, densify.DensifyState(P, "cpu"), opt, 1.0), {}, {}, 30000)
    sc3, _, _ = bench.load_workload(argparse.Namespace(scene=ck, cameras=None, gaussians=None, sh_degree=None), 64, 64)
    for k in ("means3D", "shs", "opacities", "scales"):
        np.testing.assert_allclose(sc3[k], sc[k], rtol=1e-6, atol=1e-7, err_msg=k)
