"""CPU: gi-gs_amd/dataset_readers.py against tests/golden/ref_reader.npz (the reference's own graphics / general / SH
utilities run on seeded inputs, tests/golden/make_reader_golden.py) on a synthetic Blender-style scene written to disk."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

import dataset_readers as dr

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_reader.npz"))


def _write_scene(root):
    W, H = (int(v) for v in GOLD["size"])
    img = Image.fromarray(GOLD["image_rgba"], "RGBA")
    for split, idx in (("train", range(0, 4)), ("test", range(4, 5))):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames = []
        for i in idx:
            img.save(os.path.join(root, split, f"r_{i}.png"))
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": GOLD["frames"][i].tolist()})
        # the reference resolves images by basename under DATA_SUBDIR (scene/dataset_readers.py:246-248)
        with open(os.path.join(root, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": float(GOLD["fovx"]), "frames": frames}, f)
    return W, H


def test_blender_reader_matches_reference_camera_maths(tmp_path, monkeypatch):
    W, H = _write_scene(str(tmp_path))
    monkeypatch.setenv("DATA_SUBDIR", "train")
    cams = dr.readCamerasFromTransforms(str(tmp_path), "transforms_train.json", False)
    assert [c.image_name for c in cams] == ["r_0", "r_1", "r_2", "r_3"] and cams[0].width == W and cams[0].height == H
    for i, info in enumerate(cams):
        assert abs(info.FovY - float(GOLD[f"cam{i}_fovy"])) < 1e-12
        cam = dr.camera_from_info(info)
        assert np.allclose(cam["viewmatrix"].numpy(), GOLD[f"cam{i}_viewmatrix"], atol=1e-6)
        assert np.allclose(cam["projmatrix"].numpy(), GOLD[f"cam{i}_projmatrix"], atol=2e-5)
        assert np.allclose(cam["campos"].numpy(), GOLD[f"cam{i}_campos"], atol=1e-5)
        assert cam["original_image"].shape == (3, H, W) and cam["gt_alpha_mask"].shape == (1, H, W)
        assert abs(cam["tanfovx"] - np.tan(float(GOLD["fovx"]) / 2)) < 1e-12
    half = dr.camera_from_info(cams[0], resolution=2)
    assert half["image_width"] == W // 2 and half["image_height"] == H // 2
    got = torch.cat((half["original_image"], half["gt_alpha_mask"]))
    assert np.allclose(got.numpy(), GOLD["image_resized_20x15"], atol=1e-6)
    norm = dr.getNerfppNorm(cams)
    centers = np.stack([GOLD[f"cam{i}_campos"] for i in range(4)])
    assert np.allclose(norm["translate"], -centers.mean(0), atol=1e-5)
    assert abs(norm["radius"] - 1.1 * np.linalg.norm(centers - centers.mean(0), axis=1).max()) < 1e-4


def test_scene_info_lr_schedule_and_init_cloud(tmp_path, monkeypatch):
    _write_scene(str(tmp_path))
    monkeypatch.setenv("DATA_SUBDIR", "train")
    for i in range(4, 5):  # the test split's image, resolved under the same sub-directory
        Image.fromarray(GOLD["image_rgba"], "RGBA").save(os.path.join(tmp_path, "train", f"r_{i}.png"))
    info = dr.readNerfSyntheticInfo(str(tmp_path), False, eval=True)
    assert len(info["train_cameras"]) == 4 and len(info["test_cameras"]) == 1
    assert len(dr.readNerfSyntheticInfo(str(tmp_path), False, eval=False)["train_cameras"]) == 5
    f = dr.get_expon_lr_func(lr_init=1.6e-4 * 3.7, lr_final=1.6e-6 * 3.7, lr_delay_mult=0.01, max_steps=30000)
    g = dr.get_expon_lr_func(lr_init=1e-2, lr_final=1e-4, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=2000)
    steps = GOLD["lr_steps"]
    assert np.allclose([f(int(s)) for s in steps], GOLD["lr_xyz"], rtol=1e-12)
    assert np.allclose([g(int(s)) for s in steps], GOLD["lr_delayed"], rtol=1e-12)
    assert dr.get_expon_lr_func(0.0, 0.0)(10) == 0.0 and f(-1) == 0.0
    cloud = dr.random_init_cloud(1000, np.random.default_rng(0))
    assert cloud["points"].shape == (1000, 3) and np.abs(cloud["points"]).max() <= 1.3 and not cloud["normals"].any()
    assert np.allclose(GOLD["sh_in"] * 0.28209479177387814 + 0.5, GOLD["sh2rgb"])  # SH2RGB as used for the colours
    assert cloud["colors"].min() >= 0.5 and cloud["colors"].max() <= 0.5 + 0.28209479177387814 / 255.0 + 1e-12


COLMAP = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_colmap.npz"))


def _write_colmap(root):
    sparse = os.path.join(root, "sparse", "0")
    os.makedirs(sparse, exist_ok=True)
    for fn, key in (("cameras.bin", "cameras_bin"), ("images.bin", "images_bin"), ("points3D.bin", "points3D_bin")):
        open(os.path.join(sparse, fn), "wb").write(COLMAP[key].tobytes())
    return sparse


def test_colmap_binary_readers_match_the_reference_loader(tmp_path):
    """tests/golden/ref_colmap.npz = what the reference's scene/colmap_loader.py read from the same bytes."""
    sparse = _write_colmap(str(tmp_path))
    intr = dr.read_colmap_cameras_bin(os.path.join(sparse, "cameras.bin"))
    assert sorted(intr) == COLMAP["cam_ids"].tolist()
    for cid, c in intr.items():
        assert c.model == str(COLMAP[f"cam{cid}_model"]) and [c.width, c.height] == COLMAP[f"cam{cid}_wh"].tolist()
        assert np.array_equal(c.params, COLMAP[f"cam{cid}_params"])
    extr = dr.read_colmap_images_bin(os.path.join(sparse, "images.bin"))
    assert list(extr) == COLMAP["img_ids"].tolist()  # file order is kept
    for iid, im in extr.items():
        assert np.array_equal(im.qvec, COLMAP[f"img{iid}_qvec"]) and np.array_equal(im.tvec, COLMAP[f"img{iid}_tvec"])
        assert im.camera_id == int(COLMAP[f"img{iid}_camera_id"]) and im.name == str(COLMAP[f"img{iid}_name"])
        assert np.array_equal(im.xys.reshape(-1, 2), COLMAP[f"img{iid}_xys"])
        assert np.array_equal(im.point3D_ids, COLMAP[f"img{iid}_p3d"])
        assert np.allclose(dr.qvec2rotmat(im.qvec), COLMAP[f"img{iid}_R"], rtol=0, atol=1e-15)
    xyz, rgb, err = dr.read_colmap_points3d_bin(os.path.join(sparse, "points3D.bin"))
    assert np.array_equal(xyz, COLMAP["xyz"]) and np.array_equal(rgb, COLMAP["rgb"]) and np.array_equal(err, COLMAP["err"])


def test_colmap_scene_info(tmp_path):
    _write_colmap(str(tmp_path))
    os.makedirs(os.path.join(tmp_path, "images"))
    for iid in COLMAP["img_ids"].tolist():
        name = os.path.basename(str(COLMAP[f"img{iid}_name"]))
        wh = COLMAP[f"cam{int(COLMAP[f'img{iid}_camera_id'])}_wh"]
        Image.new("RGB", (int(wh[0]), int(wh[1])), (10 * (iid % 20), 40, 90)).save(os.path.join(tmp_path, "images", name))
    info = dr.readColmapSceneInfo(str(tmp_path), None, eval=True, llffhold=2)
    names = [c.image_name for c in info["train_cameras"] + info["test_cameras"]]
    assert sorted(names) == ["view_a", "view_b", "view_c", "view_d", "view_e"]
    assert [c.image_name for c in info["test_cameras"]] == ["view_a", "view_c", "view_e"]  # every 2nd of the sorted list
    by_name = {c.image_name: c for c in info["train_cameras"] + info["test_cameras"]}
    cam = by_name["view_b"]  # image id 10, PINHOLE camera 1
    assert np.allclose(cam.R, COLMAP["img10_R"].T) and np.array_equal(cam.T, COLMAP["img10_tvec"])
    assert abs(cam.FovX - 2 * np.arctan(64 / (2 * 70.0))) < 1e-12 and abs(cam.FovY - 2 * np.arctan(48 / (2 * 72.5))) < 1e-12
    simple = by_name["view_a"]  # SIMPLE_PINHOLE camera 2
    assert abs(simple.FovX - 2 * np.arctan(50 / (2 * 55.0))) < 1e-12 and abs(simple.FovY - 2 * np.arctan(40 / (2 * 55.0))) < 1e-12
    assert info["point_cloud"]["points"].shape == (17, 3) and info["point_cloud"]["colors"].max() <= 1.0
    assert len(dr.readColmapSceneInfo(str(tmp_path), None, eval=False)["train_cameras"]) == 5
    d = dr.camera_from_info(cam)
    assert d["image_width"] == 64 and d["image_height"] == 48 and d["gt_alpha_mask"].min() == 1.0
