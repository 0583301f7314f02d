"""CPU checker for distCUDA2.  TEST INFRASTRUCTURE ONLY.

The quantity is defined exactly by submodules/simple-knn/simple_knn.cu:117-185: for every point, the three smallest squared
Euclidean distances to the OTHER points (excluded by index, so exact duplicates count with distance 0), averaged; the Morton
order and the boxes only prune.  Two independent evaluations of that definition:
  dist2_brute  O(P^2) numpy in float32 with the kernel's operation order (dx*dx + dy*dy + dz*dz), small P
  dist2_tree   scipy.spatial.cKDTree (float64), any P
No fixture of the reference exists and the extension cannot be built here (CUDA): pinned by definition.
"""
import numpy as np


def dist2_brute(points: np.ndarray) -> np.ndarray:
    p = np.asarray(points, np.float32)
    P = p.shape[0]
    out = np.empty(P, np.float32)
    big = np.float32(np.finfo(np.float32).max)
    for i in range(P):
        d = p - p[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        d2[i] = big
        if P <= 3:
            d2 = np.concatenate([d2, np.full(4 - P, big, np.float32)])
        best = np.sort(d2)[:3]
        with np.errstate(over="ignore"):
            out[i] = (np.float32(best[0] + best[1]) + best[2]) / np.float32(3.0)
    return out


def dist2_tree(points: np.ndarray) -> np.ndarray:
    from scipy.spatial import cKDTree
    p = np.asarray(points, np.float32).astype(np.float64)
    d, _ = cKDTree(p).query(p, k=4)  # the point itself (distance 0) + three others
    return ((d[:, 1:] ** 2).sum(axis=1) / 3.0).astype(np.float32)
