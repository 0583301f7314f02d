"""Generates tests/golden/ref_loss.npz (run in the authoring container only).

REFERENCE-PINNED vectors for the training-loop losses: the reference's own utils/loss_utils.py (`ssim`, `l1_loss`;
pure torch, importable on CPU) evaluated on seeded images, with the autograd gradient of the stage-1 colour loss
(train.py:320) w.r.t. the rendered image.  Only inputs and outputs are stored; the reference never travels.

    python tests/golden/make_loss_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, "/root/reference")
    from utils.loss_utils import l1_loss, ssim  # noqa: E402

    rng = np.random.default_rng(4321)
    out = {}
    cases = {"a": (3, 37, 53), "b": (3, 64, 64), "c": (1, 11, 9), "d": (3, 96, 130)}
    for name, (C, H, W) in cases.items():
        gt = rng.uniform(0, 1, size=(C, H, W)).astype(np.float32)
        # a render that resembles its target: smooth structure + noise, clipped like a real image
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        gt = np.clip(0.5 + 0.4 * np.sin(xx / 7.0 + yy / 11.0)[None] * gt, 0, 1).astype(np.float32)
        img = np.clip(gt + rng.normal(0, 0.08, size=gt.shape), 0, 1).astype(np.float32)
        if name == "c":
            img[:, :3] = gt[:, :3]  # exact ties: |x - y| = 0 has gradient 0
        x = torch.from_numpy(img).requires_grad_(True)
        y = torch.from_numpy(gt)
        lam = 0.2
        l1 = l1_loss(x, y)
        s = ssim(x, y)
        loss = (1.0 - lam) * l1 + lam * (1.0 - s)
        loss.backward()
        out[f"{name}_img"], out[f"{name}_gt"] = img, gt
        out[f"{name}_l1"] = np.float32(l1.item())
        out[f"{name}_ssim"] = np.float32(s.item())
        out[f"{name}_loss"] = np.float32(loss.item())
        out[f"{name}_grad"] = x.grad.numpy().astype(np.float32)
        x2 = torch.from_numpy(img).requires_grad_(True)
        ssim(x2, y).backward()
        out[f"{name}_grad_ssim"] = x2.grad.numpy().astype(np.float32)
    out["lambda"] = np.float32(0.2)
    np.savez_compressed(os.path.join(HERE, "ref_loss.npz"), **out)
    print("wrote ref_loss.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim > 0})


if __name__ == "__main__":
    main()
