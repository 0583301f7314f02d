"""Generates tests/golden/ref_reader.npz (run in the authoring container only).

REFERENCE-PINNED vectors for the host-side helpers of gi-gs_amd/dataset_readers.py, produced by importing the
reference's own utils/graphics_utils.py, utils/general_utils.py and utils/sh_utils.py (pure numpy / torch; importable on
CPU) on seeded inputs.  scene/dataset_readers.py itself cannot be imported (plyfile is absent): its pose conversion
(:236-244) is restated in the product and checked here through the reference's getWorld2View2 / projection functions.

    python tests/golden/make_reader_golden.py
"""
import math
import os
import sys

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, "/root/reference")
    from utils.general_utils import PILtoTorch, get_expon_lr_func  # noqa: E402
    from utils.graphics_utils import focal2fov, fov2focal, getProjectionMatrix, getWorld2View2  # noqa: E402
    from utils.sh_utils import SH2RGB  # noqa: E402

    rng = np.random.default_rng(77)
    out = {}
    fovx = 0.6911112070083618
    W, H = 40, 30
    frames = []
    for i in range(5):
        # a random rigid camera-to-world in Blender's convention
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        c2w = np.eye(4)
        c2w[:3, :3] = Q
        c2w[:3, 3] = rng.normal(size=3) * 3.0
        frames.append(c2w)
        # the reference's own pose handling (scene/dataset_readers.py:236-244), then its camera maths (scene/cameras.py:75-91)
        m = np.array(c2w)
        m[:3, 1:3] *= -1
        w2c = np.linalg.inv(m)
        R, T = np.transpose(w2c[:3, :3]), w2c[:3, 3]
        fovy = focal2fov(fov2focal(fovx, W), H)
        wvt = torch.tensor(getWorld2View2(R, T)).transpose(0, 1)
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = wvt.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
        out[f"cam{i}_viewmatrix"] = wvt.numpy()
        out[f"cam{i}_projmatrix"] = full.numpy()
        out[f"cam{i}_campos"] = wvt.inverse()[3, :3].numpy()
        out[f"cam{i}_fovy"] = np.float64(fovy)
    out["frames"] = np.stack(frames)
    out["fovx"] = np.float64(fovx)
    out["size"] = np.array([W, H])
    # PILtoTorch on an RGBA image, resized
    img = (rng.random((H, W, 4)) * 255).astype(np.uint8)
    out["image_rgba"] = img
    out["image_resized_20x15"] = PILtoTorch(Image.fromarray(img, "RGBA"), (20, 15)).numpy()
    # the xyz learning-rate schedule of training_setup (scene/gaussian_model.py:347-353; arguments defaults)
    f = get_expon_lr_func(lr_init=1.6e-4 * 3.7, lr_final=1.6e-6 * 3.7, lr_delay_mult=0.01, max_steps=30000)
    g = get_expon_lr_func(lr_init=1e-2, lr_final=1e-4, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=2000)
    steps = np.array([0, 1, 10, 499, 500, 1500, 2000, 29999, 30000, 40000])
    out["lr_steps"] = steps
    out["lr_xyz"] = np.array([f(int(s)) for s in steps], np.float64)
    out["lr_delayed"] = np.array([g(int(s)) for s in steps], np.float64)
    sh = rng.random((16, 3)) / 255.0
    out["sh_in"], out["sh2rgb"] = sh, SH2RGB(sh)
    np.savez_compressed(os.path.join(HERE, "ref_reader.npz"), **out)
    print("wrote ref_reader.npz")


if __name__ == "__main__":
    main()
