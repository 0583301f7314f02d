"""The reference's on-disk formats, so that scenes trained by either code load in the other.

    save_ply / load_ply            point_cloud.ply exactly as GaussianModel.save_ply / load_ply lay it out
                                   (scene/gaussian_model.py:397-465, 474-578): one `vertex` element of float32
                                   properties x y z, f_dc_*, f_rest_* (channel-major, i.e. the [P,K,3] features
                                   transposed to [P,3,K]), opacity, normal_0..2, albedo_0..2, roughness, metallic,
                                   scale_*, rot_*; binary little-endian as plyfile writes by default
    save_checkpoint / load_checkpoint   chkpntN.pth = {"gaussians": GaussianModel.capture() 18-tuple, "cubemap":
                                   state_dict, "light_optimizer": state_dict, "iteration"} (train.py:466-490,
                                   scene/gaussian_model.py:82-176)

Pure host code (numpy / torch serialization): no kernels, works on CPU tensors.  `plyfile` is not needed; the reader
parses the PLY header itself (ascii and binary little-endian vertex elements with scalar properties).  Checkpoints are
read with `torch.load(..., weights_only=True)`: nothing in the file is executed.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch

NAMES = ["xyz", "f_dc", "f_rest", "opacity", "normal", "albedo", "roughness", "metallic", "scaling", "rotation"]

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def attribute_names(n_dc: int, n_rest: int, n_scale: int = 3, n_rot: int = 4):
    """GaussianModel.construct_list_of_attributes (scene/gaussian_model.py:397-416)."""
    names = ["x", "y", "z"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"normal_{i}" for i in range(3)]
    names += [f"albedo_{i}" for i in range(3)]
    names += ["roughness", "metallic"]
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def _np(t) -> np.ndarray:
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save_ply(path: str, params: Dict[str, torch.Tensor]) -> None:
    """params: the ten raw (pre-activation) tensors by group name; f_dc [P,1,3], f_rest [P,K,3]."""
    xyz = _np(params["xyz"]).astype(np.float32)
    P = xyz.shape[0]
    f_dc = np.ascontiguousarray(_np(params["f_dc"]).transpose(0, 2, 1).reshape(P, -1), dtype=np.float32)
    f_rest = np.ascontiguousarray(_np(params["f_rest"]).transpose(0, 2, 1).reshape(P, -1), dtype=np.float32)
    cols = [xyz, f_dc, f_rest, _np(params["opacity"]).reshape(P, 1), _np(params["normal"]).reshape(P, 3),
            _np(params["albedo"]).reshape(P, 3), _np(params["roughness"]).reshape(P, 1),
            _np(params["metallic"]).reshape(P, 1), _np(params["scaling"]).reshape(P, -1),
            _np(params["rotation"]).reshape(P, -1)]
    table = np.ascontiguousarray(np.concatenate([c.astype(np.float32) for c in cols], axis=1), dtype="<f4")
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], cols[8].shape[1], cols[9].shape[1])
    assert table.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {P}"]
    header += [f"property float {n}" for n in names]
    header.append("end_header")
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(table.tobytes())


def read_ply_vertices(path: str) -> Dict[str, np.ndarray]:
    """-> {property name: array [P]} of the `vertex` element (scalar properties only)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements, cur = None, [], None
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                cur = {"name": tok[1], "count": int(tok[2]), "props": []}
                elements.append(cur)
            elif tok[0] == "property":
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported (element {cur['name']})")
                if tok[1] not in _PLY_TYPES:
                    raise ValueError(f"{path}: unknown property type {tok[1]}")
                cur["props"].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("binary_little_endian", "ascii"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        out = None
        for el in elements:
            dt = np.dtype([(n, "<" + t) for n, t in el["props"]])
            if fmt == "ascii":
                rows = np.loadtxt(f, max_rows=el["count"], ndmin=2) if el["count"] else np.zeros((0, len(el["props"])))
                data = {n: rows[:, i].astype(t) for i, (n, t) in enumerate(el["props"])}
            else:
                raw = f.read(dt.itemsize * el["count"])
                if len(raw) != dt.itemsize * el["count"]:
                    raise ValueError(f"{path}: truncated element {el['name']}")
                arr = np.frombuffer(raw, dtype=dt, count=el["count"])
                data = {n: arr[n] for n, _ in el["props"]}
            if el["name"] == "vertex":
                out = data
                break
        if out is None:
            raise ValueError(f"{path}: no vertex element")
        return out


def load_ply(path: str, max_sh_degree: int, device="cpu") -> Dict[str, torch.Tensor]:
    """GaussianModel.load_ply (scene/gaussian_model.py:474-578) -> the ten raw tensors by group name."""
    v = read_ply_vertices(path)

    def stack(prefix, count=None):
        names = sorted((n for n in v if n.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))
        if count is not None and len(names) != count:
            raise ValueError(f"{path}: expected {count} '{prefix}*' properties, found {len(names)}")
        return np.stack([np.asarray(v[n], dtype=np.float32) for n in names], axis=1)

    P = len(v["x"])
    K = (max_sh_degree + 1) ** 2
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1).astype(np.float32)
    f_dc = stack("f_dc_", 3).reshape(P, 3, 1).transpose(0, 2, 1)
    f_rest = stack("f_rest_", 3 * K - 3).reshape(P, 3, K - 1).transpose(0, 2, 1)  # the assert at :517
    out = dict(xyz=xyz, f_dc=f_dc, f_rest=f_rest, opacity=np.asarray(v["opacity"], np.float32)[:, None],
               normal=stack("normal_", 3), albedo=stack("albedo_", 3),
               roughness=np.asarray(v["roughness"], np.float32)[:, None],
               metallic=np.asarray(v["metallic"], np.float32)[:, None], scaling=stack("scale_"), rotation=stack("rot"))
    return {k: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device) for k, a in out.items()}


def capture(active_sh_degree: int, params: Dict[str, torch.Tensor], stats, optimizer, spatial_lr_scale: float) -> Tuple:
    """GaussianModel.capture (scene/gaussian_model.py:82-123); `stats` is a densify.DensifyState (or anything with its
    five attributes)."""
    return (active_sh_degree, params["xyz"], params["f_dc"], params["f_rest"], params["scaling"], params["rotation"],
            params["opacity"], params["normal"], params["albedo"], params["roughness"], params["metallic"],
            stats.max_radii2D, stats.xyz_gradient_accum, stats.xyz_gradient_accum_abs, stats.xyz_gradient_accum_abs_max,
            stats.denom, optimizer.state_dict(), spatial_lr_scale)


def restore(model_args: Tuple):
    """GaussianModel.restore (:125-176) -> (active_sh_degree, params dict, stats dict, optimizer state_dict,
    spatial_lr_scale); the caller builds its optimizer over `params` and calls load_state_dict."""
    if len(model_args) != 18:
        raise ValueError(f"checkpoint 'gaussians' tuple has {len(model_args)} entries, expected 18")
    (deg, xyz, f_dc, f_rest, scaling, rotation, opacity, normal, albedo, roughness, metallic, max_radii2D, accum, accum_abs,
     accum_abs_max, denom, opt_dict, spatial_lr_scale) = model_args
    params = dict(xyz=xyz, f_dc=f_dc, f_rest=f_rest, opacity=opacity, normal=normal, albedo=albedo, roughness=roughness,
                  metallic=metallic, scaling=scaling, rotation=rotation)
    stats = dict(max_radii2D=max_radii2D, xyz_gradient_accum=accum, xyz_gradient_accum_abs=accum_abs,
                 xyz_gradient_accum_abs_max=accum_abs_max, denom=denom)
    return int(deg), params, stats, opt_dict, float(spatial_lr_scale)


def save_checkpoint(path: str, gaussians: Tuple, cubemap_state: Dict, light_optimizer_state: Dict, iteration: int) -> None:
    """train.py:466-490."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    torch.save({"gaussians": gaussians, "cubemap": cubemap_state, "light_optimizer": light_optimizer_state,
                "iteration": iteration}, path)


def load_checkpoint(path: str, map_location: Optional[str] = "cpu") -> Dict:
    """train.py:223-234, but with the safe loader: tensors, numbers, tuples, dicts only."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    for k in ("gaussians", "iteration"):
        if k not in ckpt:
            raise ValueError(f"{path}: not a GI-GS checkpoint (no '{k}')")
    return ckpt


def ply_sh_degree(path: str) -> int:
    """The SH degree a point_cloud.ply was saved with, from its f_rest_* property count (3 * ((deg + 1)^2 - 1))."""
    v = read_ply_vertices(path)
    n_rest = sum(1 for n in v if n.startswith("f_rest_"))
    K = n_rest // 3 + 1
    deg = int(round(K ** 0.5)) - 1
    if (deg + 1) ** 2 != K or n_rest % 3:
        raise ValueError(f"{path}: {n_rest} f_rest_* properties do not make a whole SH degree")
    return deg


def load_scene(path: str) -> Dict[str, np.ndarray]:
    """A trained scene from disk -- `point_cloud.ply` (GaussianModel.save_ply, scene/gaussian_model.py:397-465) or
    `chkpntN.pth` (train.py:466-490) -- as the POST-ACTIVATION arrays the rasterizer takes (the getters of
    scene/gaussian_model.py:178-263 applied on the CPU): the dictionary layout of gi-gs_amd/scenes.py
    (means3D, shs [P,K,3], opacities, normal, albedo, roughness, metallic, scales, rotations, sh_degree) + "raw"."""
    if path.endswith(".ply"):
        deg = ply_sh_degree(path)
        raw = load_ply(path, deg)
    else:
        ckpt = load_checkpoint(path)
        _, raw, _, _, _ = restore(ckpt["gaussians"])
        raw = {k: (v.detach().float().cpu() if isinstance(v, torch.Tensor) else torch.as_tensor(v)) for k, v in raw.items()}
        K = 1 + int(raw["f_rest"].shape[1])
        deg = int(round(K ** 0.5)) - 1
    F = torch.nn.functional
    f32 = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)  # noqa: E731
    return dict(means3D=f32(raw["xyz"]), shs=f32(torch.cat((raw["f_dc"], raw["f_rest"]), dim=1)),
                opacities=f32(torch.sigmoid(raw["opacity"])), normal=f32(F.normalize(raw["normal"], dim=-1)),
                albedo=f32(torch.sigmoid(raw["albedo"])), roughness=f32(torch.sigmoid(raw["roughness"])),
                metallic=f32(torch.sigmoid(raw["metallic"])), scales=f32(torch.exp(raw["scaling"])),
                rotations=f32(F.normalize(raw["rotation"])), sh_degree=deg, raw={k: f32(v) for k, v in raw.items()})
