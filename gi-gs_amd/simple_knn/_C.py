"""`from simple_knn._C import distCUDA2` (scene/gaussian_model.py:22, 277-281) on libgigs_hip."""
from __future__ import annotations

import torch

import gigs_lib

_lib = gigs_lib.lib()


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points [P,3] float CUDA tensor -> [P] mean squared distance to the three nearest other points
    (submodules/simple-knn/spatial.cu:15-26)."""
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 needs a CUDA/HIP tensor: gigs-hip has no CPU path")
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("distCUDA2: points must be [P,3]")
    pts = points.contiguous().float()
    P = int(pts.shape[0])
    out = torch.full((P,), 0.0, dtype=torch.float32, device=pts.device)  # spatial.cu:20
    if P == 0:
        return out
    nbytes = int(_lib.gigs_dist2_scratch_bytes(P))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        gigs_lib.check(_lib.gigs_dist2(P, pts.data_ptr(), out.data_ptr(), scratch.data_ptr(), nbytes,
                                       torch.cuda.current_stream().cuda_stream), "dist2")
    return out
