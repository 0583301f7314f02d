"""Blender / NeRF-synthetic (TensoIR) scene reading and the host-side schedule helpers of the training loop
(second half).  Pure host code: json + numpy + PIL.

    readCamerasFromTransforms, readNerfSyntheticInfo, getNerfppNorm    scene/dataset_readers.py:56-78, 223-327
    camera_from_info                                                    utils/camera_utils.py:29-75 + scene/cameras.py:21-91
                                                                        -> the camera dict gi-gs_amd/pipeline.py takes
    random_init_cloud                                                   scene/dataset_readers.py:303-312
    get_expon_lr_func                                                   utils/general_utils.py:33-71 (xyz learning rate)

    read_colmap_*_bin, readColmapCameras, readColmapSceneInfo           scene/colmap_loader.py:27-274; scene/dataset_readers.py:81-214
"""
from __future__ import annotations

import json
import math
import os
from pathlib import Path
from typing import Callable, Dict, List, NamedTuple, Optional

import numpy as np
import torch

from scenes import projection_matrix


class CameraInfo(NamedTuple):  # scene/dataset_readers.py:35-46
    uid: int
    R: np.ndarray
    T: np.ndarray
    FovY: float
    FovX: float
    image: object  # PIL.Image.Image
    image_path: str
    image_name: str
    width: int
    height: int


def fov2focal(fov: float, pixels: int) -> float:  # utils/graphics_utils.py:86-87
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal: float, pixels: int) -> float:  # utils/graphics_utils.py:90-91
    return 2 * math.atan(pixels / (2 * focal))


def world2view(R: np.ndarray, t: np.ndarray, translate=np.array([0.0, 0.0, 0.0]), scale: float = 1.0) -> np.ndarray:
    """utils/graphics_utils.py:43-58 (getWorld2View2): R is stored transposed ("due to glm"), t is the w2c translation."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(C2W))


def pose_from_transform(transform_matrix) -> tuple:
    """scene/dataset_readers.py:236-244: Blender camera-to-world (Y up, Z back) -> (R, T) of the COLMAP convention."""
    c2w = np.array(transform_matrix, dtype=np.float64)
    c2w[:3, 1:3] *= -1
    w2c = np.linalg.inv(c2w)
    return np.transpose(w2c[:3, :3]), w2c[:3, 3]


def readCamerasFromTransforms(path: str, transformsfile: str, white_background: bool, extension: str = ".png") -> List[CameraInfo]:
    from PIL import Image
    with open(os.path.join(path, transformsfile)) as f:
        contents = json.load(f)
    fovx = contents["camera_angle_x"]
    cam_infos = []
    sub = os.environ.get("DATA_SUBDIR", "")  # scene/dataset_readers.py:246-248
    for idx, frame in enumerate(contents["frames"]):
        cam_name = os.path.join(path, frame["file_path"] + extension)
        R, T = pose_from_transform(frame["transform_matrix"])
        image_path = os.path.join(path, sub, os.path.basename(cam_name))
        image = Image.open(image_path)
        fovy = focal2fov(fov2focal(fovx, image.size[0]), image.size[1])
        cam_infos.append(CameraInfo(uid=idx, R=R, T=T, FovY=fovy, FovX=fovx, image=image, image_path=image_path,
                                    image_name=Path(cam_name).stem, width=image.size[0], height=image.size[1]))
    return cam_infos


def getNerfppNorm(cam_infos: List[CameraInfo]) -> Dict:
    """scene/dataset_readers.py:56-78: translate = -mean camera centre, radius = 1.1 * the largest distance from it
    (`cameras_extent`, the `extent` of densify_and_prune)."""
    centers = np.hstack([np.linalg.inv(world2view(c.R, c.T))[:3, 3:4] for c in cam_infos])
    center = np.mean(centers, axis=1, keepdims=True)
    diagonal = np.max(np.linalg.norm(centers - center, axis=0, keepdims=True))
    return {"translate": -center.flatten(), "radius": diagonal * 1.1}


def readNerfSyntheticInfo(path: str, white_background: bool, eval: bool, extension: str = ".png") -> Dict:
    """scene/dataset_readers.py:284-327 without the PLY side effects: cameras, normalisation and the initial cloud."""
    train = readCamerasFromTransforms(path, "transforms_train.json", white_background, extension)
    test = readCamerasFromTransforms(path, "transforms_test.json", white_background, extension)
    if not eval:
        train.extend(test)
        test = []
    return dict(train_cameras=train, test_cameras=test, nerf_normalization=getNerfppNorm(train))


def pil_to_torch(pil_image, resolution) -> torch.Tensor:
    """utils/general_utils.py:24-30 (PILtoTorch): resize, /255, [C,H,W]."""
    arr = torch.from_numpy(np.array(pil_image.resize(resolution))) / 255.0
    return arr.permute(2, 0, 1) if arr.dim() == 3 else arr.unsqueeze(dim=-1).permute(2, 0, 1)


def camera_from_info(info: CameraInfo, resolution: int = 1, resolution_scale: float = 1.0, device="cpu") -> Dict:
    """loadCam + Camera.__init__ (utils/camera_utils.py:29-75; scene/cameras.py:21-91) -> the camera dict of
    gi-gs_amd/scenes.py plus `original_image` [3,H,W] (clamped to [0,1]) and `gt_alpha_mask` [1,H,W]."""
    orig_w, orig_h = info.image.size
    if resolution in (1, 2, 4, 8):
        res = (round(orig_w / (resolution_scale * resolution)), round(orig_h / (resolution_scale * resolution)))
    else:
        down = (orig_w / 1600 if orig_w > 1600 else 1) if resolution == -1 else orig_w / resolution
        scale = float(down) * float(resolution_scale)
        res = (int(orig_w / scale), int(orig_h / scale))
    img = pil_to_torch(info.image, res)
    W2C = world2view(info.R, info.T)
    viewmatrix = np.ascontiguousarray(W2C.T)
    P = projection_matrix(0.01, 100.0, info.FovX, info.FovY)
    projmatrix = np.ascontiguousarray((viewmatrix @ P.T).astype(np.float32))
    campos = np.ascontiguousarray(np.linalg.inv(viewmatrix)[3, :3].astype(np.float32))
    H, W = int(img.shape[1]), int(img.shape[2])
    mask = img[3:4] if img.shape[0] == 4 else torch.ones((1, H, W))
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    return dict(viewmatrix=t(viewmatrix), projmatrix=t(projmatrix), campos=t(campos),
                tanfovx=math.tan(info.FovX * 0.5), tanfovy=math.tan(info.FovY * 0.5), image_width=W, image_height=H,
                fovx=info.FovX, fovy=info.FovY, original_image=img[:3].clamp(0.0, 1.0).float().to(device),
                gt_alpha_mask=mask.float().to(device), image_name=info.image_name, uid=info.uid)


def cameras_from_transforms(transformsfile: str, width: int, height: int) -> List[Dict]:
    """The poses of a Blender / NeRF-synthetic `transforms_*.json` (scene/dataset_readers.py:223-281) as camera dicts at a
    GIVEN resolution, without opening the image files (bench.py --cameras: the poses and the field of view are what the
    workload needs).  Same matrices as camera_from_info for an image of that size."""
    with open(transformsfile) as f:
        contents = json.load(f)
    fovx = float(contents["camera_angle_x"])
    fovy = focal2fov(fov2focal(fovx, width), height)
    cams = []
    for idx, frame in enumerate(contents["frames"]):
        R, T = pose_from_transform(frame["transform_matrix"])
        viewmatrix = np.ascontiguousarray(world2view(R, T).T)
        P = projection_matrix(0.01, 100.0, fovx, fovy)
        cams.append(dict(viewmatrix=viewmatrix, projmatrix=np.ascontiguousarray((viewmatrix @ P.T).astype(np.float32)),
                         campos=np.ascontiguousarray(np.linalg.inv(viewmatrix)[3, :3].astype(np.float32)),
                         tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5), image_width=int(width),
                         image_height=int(height), fovx=fovx, fovy=fovy, image_name=os.path.basename(str(frame.get("file_path", idx))), uid=idx))
    return cams


def random_init_cloud(num_pts: int = 100_000, rng: Optional[np.random.Generator] = None) -> Dict[str, np.ndarray]:
    """scene/dataset_readers.py:303-312: uniform points in [-1.3, 1.3]^3, colours SH2RGB(u / 255), zero normals."""
    rng = rng or np.random.default_rng()
    xyz = rng.random((num_pts, 3)) * 2.6 - 1.3
    shs = rng.random((num_pts, 3)) / 255.0
    return dict(points=xyz, colors=shs * 0.28209479177387814 + 0.5, normals=np.zeros((num_pts, 3)))


def get_expon_lr_func(lr_init: float, lr_final: float, lr_delay_steps: int = 0, lr_delay_mult: float = 1.0,
                      max_steps: int = 1000000) -> Callable[[int], float]:
    """utils/general_utils.py:33-71: log-linear interpolation lr_init -> lr_final with an optional eased-in delay."""

    def helper(step: int) -> float:
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        delay = 1.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        t = np.clip(step / max_steps, 0, 1)
        return delay * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)

    return helper


# ---- COLMAP sparse models (scene/colmap_loader.py:27-274; scene/dataset_readers.py:81-214) ----------------------
import struct  # noqa: E402

# model id -> (name, number of parameters), COLMAP's src/base/camera_models.h as listed at scene/colmap_loader.py:27-39
COLMAP_CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5),
                        4: ("OPENCV", 8), 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5),
                        8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}


class ColmapCamera(NamedTuple):
    id: int
    model: str
    width: int
    height: int
    params: np.ndarray


class ColmapImage(NamedTuple):
    id: int
    qvec: np.ndarray
    tvec: np.ndarray
    camera_id: int
    name: str
    xys: np.ndarray
    point3D_ids: np.ndarray


def qvec2rotmat(q) -> np.ndarray:
    """scene/colmap_loader.py:46-65: rotation matrix of the (w, x, y, z) quaternion, not normalised."""
    w, x, y, z = (float(v) for v in q)
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def read_colmap_cameras_bin(path: str) -> Dict[int, ColmapCamera]:
    """cameras.bin (scene/colmap_loader.py:249-273): u64 count, then per camera i32 id, i32 model, u64 width, u64 height,
    f64 params[model]."""
    buf = open(path, "rb").read()
    (n,), off = struct.unpack_from("<Q", buf, 0), 8
    out = {}
    for _ in range(n):
        cam_id, model_id, width, height = struct.unpack_from("<iiQQ", buf, off)
        off += 24
        if model_id not in COLMAP_CAMERA_MODELS:
            raise ValueError(f"{path}: unknown COLMAP camera model id {model_id}")
        name, n_params = COLMAP_CAMERA_MODELS[model_id]
        params = np.frombuffer(buf, dtype="<f8", count=n_params, offset=off).copy()
        off += 8 * n_params
        out[cam_id] = ColmapCamera(cam_id, name, int(width), int(height), params)
    if len(out) != n:
        raise ValueError(f"{path}: duplicate camera ids")
    return out


def read_colmap_images_bin(path: str) -> Dict[int, ColmapImage]:
    """images.bin (scene/colmap_loader.py:207-246): u64 count, then per image i32 id, f64 qvec[4], f64 tvec[3], i32
    camera id, NUL-terminated name, u64 n, n x (f64 x, f64 y, i64 point3D id)."""
    buf = open(path, "rb").read()
    (n,), off = struct.unpack_from("<Q", buf, 0), 8
    rec = np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")])
    out = {}
    for _ in range(n):
        vals = struct.unpack_from("<idddddddi", buf, off)
        off += 64
        end = buf.index(b"\x00", off)
        name = buf[off:end].decode("utf-8")
        off = end + 1
        (m,) = struct.unpack_from("<Q", buf, off)
        off += 8
        pts = np.frombuffer(buf, dtype=rec, count=m, offset=off)
        off += 24 * m
        out[vals[0]] = ColmapImage(vals[0], np.array(vals[1:5]), np.array(vals[5:8]), vals[8], name,
                                   np.column_stack([pts["x"], pts["y"]]) if m else np.zeros((0, 2)),
                                   pts["id"].astype(np.int64))
    return out


def read_colmap_points3d_bin(path: str):
    """points3D.bin (scene/colmap_loader.py:149-177) -> (xyz [N,3] f64, rgb [N,3] f64, error [N,1] f64)."""
    buf = open(path, "rb").read()
    (n,), off = struct.unpack_from("<Q", buf, 0), 8
    xyz, rgb, err = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 1))
    for i in range(n):
        v = struct.unpack_from("<QdddBBBd", buf, off)
        off += 43
        (track,) = struct.unpack_from("<Q", buf, off)
        off += 8 + 8 * track
        xyz[i], rgb[i], err[i] = v[1:4], v[4:7], v[7]
    return xyz, rgb, err


def readColmapCameras(cam_extrinsics: Dict[int, ColmapImage], cam_intrinsics: Dict[int, ColmapCamera],
                      images_folder: str) -> List[CameraInfo]:
    """scene/dataset_readers.py:81-134: only undistorted (SIMPLE_PINHOLE / PINHOLE) cameras."""
    from PIL import Image
    infos = []
    for key in cam_extrinsics:
        extr = cam_extrinsics[key]
        intr = cam_intrinsics[extr.camera_id]
        R = np.transpose(qvec2rotmat(extr.qvec))
        T = np.array(extr.tvec)
        if intr.model == "SIMPLE_PINHOLE":
            fovy, fovx = focal2fov(intr.params[0], intr.height), focal2fov(intr.params[0], intr.width)
        elif intr.model == "PINHOLE":
            fovy, fovx = focal2fov(intr.params[1], intr.height), focal2fov(intr.params[0], intr.width)
        else:
            raise AssertionError("Colmap camera model not handled: only undistorted datasets (PINHOLE or SIMPLE_PINHOLE "
                                 "cameras) supported!")
        image_path = os.path.join(images_folder, os.path.basename(extr.name))
        infos.append(CameraInfo(uid=intr.id, R=R, T=T, FovY=fovy, FovX=fovx, image=Image.open(image_path),
                                image_path=image_path, image_name=os.path.basename(image_path).split(".")[0],
                                width=intr.width, height=intr.height))
    return infos


def readColmapSceneInfo(path: str, images: Optional[str], eval: bool, llffhold: int = 8) -> Dict:
    """scene/dataset_readers.py:175-222 for binary models: cameras sorted by image name, every llffhold-th one held out
    when eval, the normalisation of the training cameras, the sparse points as the initial cloud."""
    sparse = os.path.join(path, "sparse/0")
    extr = read_colmap_images_bin(os.path.join(sparse, "images.bin"))
    intr = read_colmap_cameras_bin(os.path.join(sparse, "cameras.bin"))
    cams = sorted(readColmapCameras(extr, intr, os.path.join(path, "images" if images is None else images)),
                  key=lambda c: c.image_name)
    train = [c for i, c in enumerate(cams) if not eval or i % llffhold != 0]
    test = [c for i, c in enumerate(cams) if eval and i % llffhold == 0]
    cloud = None
    pts = os.path.join(sparse, "points3D.bin")
    if os.path.exists(pts):
        xyz, rgb, _ = read_colmap_points3d_bin(pts)
        cloud = dict(points=xyz, colors=rgb / 255.0, normals=np.zeros_like(xyz))  # storePly / fetchPly round trip (:136-173)
    return dict(train_cameras=train, test_cameras=test, nerf_normalization=getNerfppNorm(train), point_cloud=cloud)
