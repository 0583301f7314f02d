"""CPU: the restated training-loop glue (oracle/train_glue_ref.py) against what pins it.

  * ssim / l1 / the stage-1 colour loss and their gradients  <- tests/golden/ref_loss.npz, produced by the
    reference's own utils/loss_utils.py (tests/golden/make_loss_golden.py)
  * the Adam update                                         <- torch.optim.Adam(eps=1e-15), the optimizer the
    reference constructs (scene/gaussian_model.py:346)
  * the TV / masked-L1 restatements                         <- hand-computed cases (parity otherwise unpinned:
    train.py is not importable without kornia / nvdiffrast)
and the host-side logic of FusedAdam that needs no GPU.
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import train_glue_ref as ref

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_loss.npz"))
CASES = ["a", "b", "c", "d"]


@pytest.mark.parametrize("case", CASES)
def test_restated_ssim_and_l1_match_the_reference_fixture(case):
    x = torch.from_numpy(GOLD[f"{case}_img"]).requires_grad_(True)
    y = torch.from_numpy(GOLD[f"{case}_gt"])
    lam = float(GOLD["lambda"])
    loss = ref.l1_ssim_ref(x, y, lam)
    loss.backward()
    assert abs(ref.l1_ref(x, y).item() - float(GOLD[f"{case}_l1"])) < 1e-7
    assert abs(ref.ssim_ref(x, y).item() - float(GOLD[f"{case}_ssim"])) < 2e-6
    assert abs(loss.item() - float(GOLD[f"{case}_loss"])) < 1e-6
    g = GOLD[f"{case}_grad"]
    assert np.abs(x.grad.numpy() - g).max() < 1e-6 * max(1.0, np.abs(g).max() * 1e3)


def test_tv_restatement_on_a_hand_computed_case():
    # 1 prediction channel, 2x2; gt varies only along x in channel 0
    gt = torch.zeros(3, 2, 2)
    gt[0, :, 1] = 0.3
    pred = torch.tensor([[[1.0, 3.0], [2.0, 7.0]]])
    wx = math.exp(-0.3 / 3.0)  # mean over the 3 gt channels of |0.3 - 0|
    # vertical pairs (weight 1): (2-1)^2, (7-3)^2 -> mean 8.5; horizontal: (3-1)^2 wx, (7-2)^2 wx -> mean 14.5 wx
    want = 8.5 + 14.5 * wx
    assert abs(ref.tv_ref(gt, pred).item() - want) < 1e-5
    mask = torch.tensor([[[1.0, 1.0], [0.0, 1.0]]])
    # masked: vertical pair x=0 dropped, horizontal pair y=1 dropped; the means keep their denominators
    want_m = (16.0 / 2.0) + (4.0 * wx / 2.0)
    assert abs(ref.tv_ref(gt, pred, mask=mask).item() - want_m) < 1e-5
    assert ref.tv_ref(gt, torch.full((5, 2, 2), 0.7)).item() == 0.0


def test_masked_l1_restatement():
    a = torch.arange(12.0).reshape(3, 2, 2)
    b = torch.zeros(3, 2, 2)
    mask = torch.tensor([[True, False], [False, True]])
    # selected pixels 0 and 3 of each channel: (0+3) + (4+7) + (8+11) over 6 values
    assert abs(ref.masked_l1_ref(a, b, mask).item() - 33.0 / 6.0) < 1e-6
    assert math.isnan(ref.masked_l1_ref(a, b, torch.zeros(2, 2, dtype=torch.bool)).item())


def test_adam_restatement_follows_torch_adam():
    torch.manual_seed(0)
    p0 = torch.randn(1000)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p], lr=0.0, eps=1e-15)
    q, m, v = p0.clone(), torch.zeros(1000), torch.zeros(1000)
    for step in range(1, 6):
        g = torch.randn(1000) * (10.0 ** (step - 3))
        lr = 1e-2 / step
        opt.param_groups[0]["lr"] = lr
        p.grad = g.clone()
        opt.step()
        q, m, v = ref.adam_ref(q, g, m, v, step, lr)
        assert torch.allclose(q, p.detach(), rtol=2e-6, atol=1e-8)
        st = opt.state[p]
        # element-wise rounding differs where g and m cancel: compare against the tensor's scale
        assert (m - st["exp_avg"]).abs().max() <= 1e-6 * st["exp_avg"].abs().max()
        assert (v - st["exp_avg_sq"]).abs().max() <= 1e-6 * st["exp_avg_sq"].abs().max()


def test_product_losses_and_optimizer_have_no_cpu_path():
    import losses
    import optim
    x = torch.rand(3, 16, 16, requires_grad=True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        losses.l1_ssim_loss(x, torch.rand(3, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU path"):
        losses.get_tv_loss(torch.rand(3, 16, 16), x)
    with pytest.raises(RuntimeError, match="no CPU path"):
        losses.masked_l1_loss(x, torch.rand(3, 16, 16), torch.ones(16, 16, dtype=torch.bool))
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        optim.FusedAdam([p], lr=1e-3).step()
    with pytest.raises(ValueError):
        optim.FusedAdam([p], lr=-1.0)
    import gigs_lib
    assert gigs_lib.lib().gigs_loss_scratch_floats(3, 800, 800) >= 2 * 3 * 25 * 25
    assert gigs_lib.lib().gigs_loss_scratch_floats(0, 800, 800) == 0
