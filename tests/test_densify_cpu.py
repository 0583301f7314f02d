"""CPU: the stepwise restatement of the reference's densification (oracle/densify_ref.py) on a case small enough to
follow by hand, and the product's refusal to run without the GPU."""
import pytest
import torch

from oracle import densify_ref as ref

NAMES = ref.NAMES
SHAPES = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (2, 3), "opacity": (1,), "normal": (3,), "albedo": (3,),
          "roughness": (1,), "metallic": (1,), "scaling": (3,), "rotation": (4,)}


def _tiny():
    P = 5
    params = {n: torch.arange(P, dtype=torch.float32).reshape((P,) + (1,) * len(SHAPES[n])).expand((P,) + SHAPES[n]).clone()
              for n in NAMES}
    params["rotation"] = torch.tensor([[1.0, 0, 0, 0]] * P)
    #            small+grad  big+grad   small,quiet  big,quiet  transparent
    params["scaling"] = torch.log(torch.tensor([[0.01] * 3, [0.2] * 3, [0.01] * 3, [0.2] * 3, [0.01] * 3]))
    params["opacity"] = torch.tensor([[2.0], [2.0], [2.0], [2.0], [-6.0]])
    zeros = lambda: {n: torch.zeros_like(params[n]) for n in NAMES}  # noqa: E731
    stats = dict(accum=torch.tensor([[1e-3], [1e-3], [0.0], [0.0], [0.0]]), accum_abs=torch.tensor([[2e-3], [2e-3], [0.0], [0.0], [0.0]]),
                 accum_abs_max=torch.zeros(P, 1), denom=torch.tensor([[1.0], [1.0], [1.0], [0.0], [1.0]]),
                 max_radii2D=torch.zeros(P))
    return dict(params=params, exp_avg=zeros(), exp_avg_sq=zeros(), stats=stats)


def test_restated_densify_on_a_hand_checked_case():
    m = _tiny()
    m["exp_avg"]["albedo"] += 7.0
    z = torch.ones(10, 3)
    ref.densify_and_prune(m, max_grad=2e-4, min_opacity=0.05, extent=4.0, max_screen_size=None, z_clone=z, z_split=z)
    # ratio = 2/5 -> Q = the 0.6-quantile of [0, 0, 0, 2e-3, 2e-3] = 8e-4: rows 0 and 1 are selected.
    # row 0 cloned, row 1 split in two (and removed), row 3 has denom 0 (NaN -> 0), row 4 pruned for opacity:
    # survivors = [0, 2, 3] + clone of 0 + two children of 1
    p = m["params"]
    assert p["xyz"].shape[0] == 6
    assert p["albedo"][:, 0].tolist() == [0.0, 2.0, 3.0, 0.0, 1.0, 1.0]
    assert m["exp_avg"]["albedo"][:, 0].tolist() == [7.0, 7.0, 7.0, 0.0, 0.0, 0.0]  # new rows start with zero moments
    # identity rotation: sample = scale * z
    assert torch.allclose(p["xyz"][3], torch.tensor([0.01] * 3))
    assert torch.allclose(p["xyz"][4], torch.tensor([1.2] * 3)) and torch.allclose(p["xyz"][5], torch.tensor([1.2] * 3))
    assert torch.allclose(torch.exp(p["scaling"][4]), torch.tensor([0.2 / 1.6] * 3))
    assert all(float(v.abs().sum()) == 0.0 for v in m["stats"].values()) and m["stats"]["denom"].shape == (6, 1)


def test_world_size_prune_only_with_a_screen_threshold():
    m = _tiny()
    m["params"]["scaling"][2] = torch.log(torch.tensor([0.5] * 3))  # > 0.1 * extent, quiet
    z = torch.zeros(10, 3)
    a = _tiny()
    a["params"]["scaling"][2] = m["params"]["scaling"][2]
    ref.densify_and_prune(m, 2e-4, 0.05, 4.0, None, z, z)
    ref.densify_and_prune(a, 2e-4, 0.05, 4.0, 20, z, z)
    assert m["params"]["xyz"].shape[0] == 6 and a["params"]["xyz"].shape[0] == 5


def test_product_densify_has_no_cpu_path():
    import densify
    st = densify.DensifyState(4, "cpu")
    with pytest.raises(RuntimeError, match="no CPU path"):
        densify.add_densification_stats(st, torch.zeros(4, 3), torch.ones(4, dtype=torch.int32))
