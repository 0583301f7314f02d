"""CPU: the two evaluations of distCUDA2's definition agree with each other and with a hand-computed case."""
import numpy as np
import pytest

from oracle import knn_ref


def test_brute_force_and_kdtree_agree():
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(800, 3)).astype(np.float32)
    a, b = knn_ref.dist2_brute(pts), knn_ref.dist2_tree(pts)
    assert np.allclose(a, b, rtol=1e-5)
    line = np.zeros((6, 3), np.float32)
    line[:, 1] = [0, 1, 3, 6, 10, 15]
    assert np.allclose(knn_ref.dist2_brute(line), [(1 + 9 + 36) / 3, (1 + 4 + 25) / 3, (4 + 9 + 9) / 3, (9 + 16 + 25) / 3,
                                                   (16 + 25 + 49) / 3, (25 + 81 + 144) / 3])
    # fewer than four points: the FLT_MAX placeholders stay in the sum (simple_knn.cu:140, 184)
    assert np.all(knn_ref.dist2_brute(line[:3]) > 1e38) and np.all(np.isinf(knn_ref.dist2_brute(line[:2])))


def test_product_knn_has_no_cpu_path():
    import torch
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="no CPU path"):
        distCUDA2(torch.zeros(8, 3))
