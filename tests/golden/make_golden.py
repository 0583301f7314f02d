"""Generates the golden fixtures under tests/golden/ (run in the authoring container only).

Two kinds of vectors:
  (1) REFERENCE-PINNED: produced by importing the reference's own Python
      (/root/reference/utils/sh_utils.py, utils/graphics_utils.py) on seeded inputs.  They
      pin the oracle's SH colour evaluation (forward.cu:22-80 restated in
      oracle/gigs_oracle.cpp::sh_to_rgb) and the camera-matrix conventions
      (gi-gs_amd/scenes.py).  The reference itself never travels: only inputs/outputs do.
  (2) ORACLE-FROZEN: small outputs of oracle/ itself on seeded scenes, so that an oracle
      rebuilt on another machine (the GPU box) can be checked against the one that was
      reviewed here.  These do NOT pin parity with the reference; they pin reproducibility.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gi-gs_amd"))


def reference_vectors():
    sys.path.insert(0, "/root/reference")
    from utils.graphics_utils import getProjectionMatrix, getWorld2View2  # noqa: E402
    from utils.sh_utils import RGB2SH, eval_sh  # noqa: E402

    rng = np.random.default_rng(1234)
    out = {}
    # ---- SH: reference layout is sh[..., C, (deg+1)^2]; kernel layout is [P, M, 3]
    P = 257
    means = rng.uniform(-2, 2, size=(P, 3)).astype(np.float32)
    campos = np.array([0.3, -4.0, 1.1], np.float32)
    dirs = means - campos[None]
    dirs = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    for deg in range(4):
        M = 16  # allocated coefficients; `deg` is the active degree
        shs = rng.normal(0, 0.6, size=(P, M, 3)).astype(np.float32)
        ref = eval_sh(deg, torch.from_numpy(shs).transpose(1, 2), torch.from_numpy(dirs.astype(np.float32)))
        col = torch.clamp_min(ref + 0.5, 0.0).numpy()  # gaussian_renderer/__init__.py:119-123
        out[f"sh_deg{deg}_shs"] = shs
        out[f"sh_deg{deg}_rgb"] = col.astype(np.float32)
        out[f"sh_deg{deg}_clamped"] = ((ref + 0.5) < 0).numpy()
    out["sh_means"] = means
    out["sh_campos"] = campos
    out["rgb2sh_in"] = rng.uniform(0, 1, size=(16, 3)).astype(np.float32)
    out["rgb2sh_out"] = RGB2SH(out["rgb2sh_in"]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_sh.npz"), **out)

    # ---- cameras
    cams = {}
    for i, (fx, fy) in enumerate([(0.6911, 0.6911), (1.2, 0.8), (0.4, 0.9)]):
        cams[f"proj_{i}_fov"] = np.array([fx, fy], np.float64)
        cams[f"proj_{i}"] = getProjectionMatrix(0.01, 100.0, fx, fy).numpy()
    import scenes
    for i in range(3):
        cam = scenes.orbit_camera(i, 3, 400, 300)
        W2C = cam["viewmatrix"].T
        R = W2C[:3, :3].T.astype(np.float64)  # reference stores R transposed (scene/cameras.py)
        T = W2C[:3, 3].astype(np.float64)
        cams[f"w2v_{i}_R"] = R
        cams[f"w2v_{i}_T"] = T
        cams[f"w2v_{i}"] = getWorld2View2(R, T)
    np.savez_compressed(os.path.join(HERE, "ref_camera.npz"), **cams)


def oracle_frozen():
    import scenes
    from oracle import oracle as orc

    orc.set_threads(1)
    sc = scenes.random_scene(P=600, sh_degree=2, seed=7, scale_mu=0.08)
    cam = scenes.orbit_camera(1, 5, 96, 80)
    r = orc.Rasterizer()
    out = r.forward(bg=np.array([0.1, 0.2, 0.3], np.float32), means3D=sc["means3D"], opacities=sc["opacities"],
                    normal=sc["normal"], albedo=sc["albedo"], roughness=sc["roughness"], metallic=sc["metallic"],
                    shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"], sh_degree=2,
                    viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], campos=cam["campos"],
                    tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], image_height=80, image_width=96)
    frozen = dict(radii=out["radii"], num_rendered=np.int64(out["num_rendered"]),
                  point_list=r.state("point_list"), ranges=r.state("ranges"), n_contrib=r.state("n_contrib"),
                  color=out["color"], depth=out["depth"], opacity=out["opacity"], normal=out["normal"])
    np.savez_compressed(os.path.join(HERE, "oracle_frozen_small.npz"), **frozen)


if __name__ == "__main__":
    reference_vectors()
    oracle_frozen()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
