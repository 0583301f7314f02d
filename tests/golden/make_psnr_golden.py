"""Generates tests/golden/ref_psnr.npz (run in the authoring container only).

REFERENCE-PINNED vectors for the "PSNR vs ref" half of BASELINE's metric: the reference's own
utils/image_utils.py `psnr` (pure torch, importable on CPU), called the way train.py:786 / render.py:379 call it
(`psnr(a, b).mean()` on [3,H,W] images), on the seeded image pairs already stored in ref_loss.npz.  Only inputs'
names and outputs are stored; the reference never travels.

    python tests/golden/make_psnr_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, "/root/reference")
    from utils.image_utils import psnr  # noqa: E402

    src = np.load(os.path.join(HERE, "ref_loss.npz"))
    out = {}
    for name in "abcd":
        a, b = torch.from_numpy(src[f"{name}_img"]), torch.from_numpy(src[f"{name}_gt"])
        per_channel = psnr(a, b)
        out[f"{name}_psnr_rows"] = per_channel.numpy().astype(np.float32)
        out[f"{name}_psnr_mean"] = np.float32(per_channel.mean().item())
    np.savez_compressed(os.path.join(HERE, "ref_psnr.npz"), **out)
    print({k: (v.tolist() if v.ndim else float(v)) for k, v in out.items()})


if __name__ == "__main__":
    main()
