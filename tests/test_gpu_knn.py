"""GPU: distCUDA2 (gi-gs_amd/simple_knn -> gigs_dist2) against the definition evaluated two ways on the CPU
(oracle/knn_ref.py).  Tolerance 1e-5 relative (fp32 squared distances; the kd-tree works in float64)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(pts):
    from simple_knn._C import distCUDA2
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return distCUDA2(torch.from_numpy(np.asarray(pts, np.float32)).cuda()).cpu().numpy()


@pytest.mark.parametrize("P", [4, 5, 63, 64, 65, 1000, 4097])
def test_small_sets_match_brute_force(P):
    from oracle import knn_ref
    rng = np.random.default_rng(P)
    pts = rng.normal(size=(P, 3)).astype(np.float32)
    got, want = _run(pts), knn_ref.dist2_brute(pts)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-12), np.abs(got - want).max()


def test_duplicates_clusters_and_degenerate_axes():
    from oracle import knn_ref
    rng = np.random.default_rng(1)
    a = rng.normal(size=(500, 3)).astype(np.float32)
    pts = np.concatenate([a, a[:100], a[:50], a[:50]])          # pairs and quadruples of identical points
    pts = np.concatenate([pts, rng.normal(size=(300, 3)).astype(np.float32) * 1e-3 + 5.0])  # a tight far cluster
    got = _run(pts)
    assert np.allclose(got, knn_ref.dist2_brute(pts), rtol=1e-5, atol=1e-12)
    assert np.all(got[:50] == 0.0)                              # four copies -> three neighbours at distance 0
    flat = rng.normal(size=(700, 3)).astype(np.float32)
    flat[:, 2] = 0.25                                           # zero extent along z
    assert np.allclose(_run(flat), knn_ref.dist2_brute(flat), rtol=1e-5, atol=1e-12)
    line = np.zeros((130, 3), np.float32)
    line[:, 0] = np.arange(130)
    got = _run(line)
    assert got[0] == np.float32((1 + 4 + 9) / 3) and got[64] == np.float32((1 + 1 + 4) / 3)


def test_fewer_than_four_points_and_empty():
    from simple_knn._C import distCUDA2
    assert distCUDA2(torch.zeros(0, 3).cuda()).shape == (0,)
    from oracle import knn_ref
    for P in (1, 2, 3):
        pts = np.arange(3 * P, dtype=np.float32).reshape(P, 3)
        got = _run(pts)                                         # FLT_MAX placeholders stay in the sum, as in the reference
        assert np.all(got > 1e38) and np.array_equal(got, knn_ref.dist2_brute(pts)), got


def test_full_size_matches_kdtree_and_scale_init():
    from oracle import knn_ref
    import scenes
    sc = scenes.surface_scene(P=300_000, sh_degree=0, seed=0)
    pts = sc["means3D"]
    got, want = _run(pts), knn_ref.dist2_tree(pts)
    assert np.allclose(got, want, rtol=2e-5, atol=1e-12), np.abs(got - want).max()
    # scene/gaussian_model.py:277-281: scales = log(sqrt(clamp_min(dist2, 1e-7)))
    s = np.log(np.sqrt(np.maximum(got, 1e-7)))
    assert np.isfinite(s).all()
    # order independence: a permutation of the input permutes the output
    perm = np.random.default_rng(3).permutation(pts.shape[0])
    assert np.array_equal(_run(pts[perm]), got[perm])
