"""Generates tests/golden/ref_colmap.npz (run in the authoring container only).

A small synthetic COLMAP sparse model (cameras.bin, images.bin, points3D.bin written here following COLMAP's binary
layout) is parsed by the reference's own scene/colmap_loader.py (numpy + struct only: importable on CPU); the file bytes
and what the reference read from them are stored.  Only data travels, never the reference.

    python tests/golden/make_colmap_golden.py
"""
import os
import struct
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    # scene/__init__.py pulls in plyfile (absent here); colmap_loader.py itself needs numpy / struct only, so the
    # module file is loaded on its own
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_colmap_loader", "/root/reference/scene/colmap_loader.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    qvec2rotmat, read_extrinsics_binary = mod.qvec2rotmat, mod.read_extrinsics_binary
    read_intrinsics_binary, read_points3D_binary = mod.read_intrinsics_binary, mod.read_points3D_binary
    rng = np.random.default_rng(5)
    cams = [(1, 1, 64, 48, [70.0, 72.5, 32.0, 24.0]), (2, 0, 50, 40, [55.0, 25.0, 20.0])]
    cam_bytes = struct.pack("<Q", len(cams))
    for cid, model, w, h, params in cams:
        cam_bytes += struct.pack("<iiQQ", cid, model, w, h) + struct.pack("<%dd" % len(params), *params)
    names = ["view_b.png", "view_a.jpg", "sub/view_c.png", "view_d.png", "view_e.png"]
    img_bytes = struct.pack("<Q", len(names))
    for i, name in enumerate(names):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        t = rng.normal(size=3) * 2
        m = int(rng.integers(0, 4))
        img_bytes += struct.pack("<idddddddi", 10 + i, *q, *t, 1 + (i % 2)) + name.encode() + b"\x00"
        img_bytes += struct.pack("<Q", m)
        for _ in range(m):
            img_bytes += struct.pack("<ddq", *rng.random(2) * 40, int(rng.integers(-1, 50)))
    n_pts = 17
    pt_bytes = struct.pack("<Q", n_pts)
    for i in range(n_pts):
        track = int(rng.integers(0, 4))
        pt_bytes += struct.pack("<QdddBBBd", 100 + i, *rng.normal(size=3), *[int(v) for v in rng.integers(0, 256, 3)],
                                float(rng.random()))
        pt_bytes += struct.pack("<Q", track) + struct.pack("<%di" % (2 * track), *[int(v) for v in rng.integers(0, 9, 2 * track)])
    out = {"cameras_bin": np.frombuffer(cam_bytes, np.uint8), "images_bin": np.frombuffer(img_bytes, np.uint8),
           "points3D_bin": np.frombuffer(pt_bytes, np.uint8)}
    with tempfile.TemporaryDirectory() as d:
        for fn, b in (("cameras.bin", cam_bytes), ("images.bin", img_bytes), ("points3D.bin", pt_bytes)):
            open(os.path.join(d, fn), "wb").write(b)
        intr = read_intrinsics_binary(os.path.join(d, "cameras.bin"))
        extr = read_extrinsics_binary(os.path.join(d, "images.bin"))
        xyz, rgb, err = read_points3D_binary(os.path.join(d, "points3D.bin"))
    out["cam_ids"] = np.array(sorted(intr))
    for cid, c in intr.items():
        out[f"cam{cid}_model"] = np.array(c.model)
        out[f"cam{cid}_wh"] = np.array([c.width, c.height])
        out[f"cam{cid}_params"] = c.params
    out["img_ids"] = np.array(list(extr))
    for iid, im in extr.items():
        out[f"img{iid}_qvec"], out[f"img{iid}_tvec"] = im.qvec, im.tvec
        out[f"img{iid}_camera_id"] = np.array(im.camera_id)
        out[f"img{iid}_name"] = np.array(im.name)
        out[f"img{iid}_xys"] = im.xys.reshape(-1, 2)
        out[f"img{iid}_p3d"] = im.point3D_ids
        out[f"img{iid}_R"] = qvec2rotmat(im.qvec)
    out["xyz"], out["rgb"], out["err"] = xyz, rgb, err
    np.savez_compressed(os.path.join(HERE, "ref_colmap.npz"), **out)
    print("wrote ref_colmap.npz")


if __name__ == "__main__":
    main()
