"""Times distCUDA2 (gigs_dist2) on the C2 / C4 point counts.  python tools/knn_bench.py [--P 300000]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gi-gs_amd"))
import gigs_lib  # noqa: E402
import scenes  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, nargs="+", default=[300000, 3000000])
    a = ap.parse_args()
    out = {}
    for P in a.P:
        pts = torch.from_numpy(scenes.surface_scene(P=P, sh_degree=0, seed=0)["means3D"]).cuda()
        uni = torch.rand(P, 3, device="cuda") * 2.6 - 1.3  # the reference's random init cloud (dataset_readers.py:308)
        for name, x in (("surface", pts), ("uniform", uni)):
            distCUDA2(x)
            torch.cuda.synchronize()
            with gigs_lib.profile() as prof:
                for _ in range(5):
                    distCUDA2(x)
                torch.cuda.synchronize()
            ms = prof.stages["dist2"][0] / prof.stages["dist2"][1]
            out[f"{name}_{P}"] = {"ms": round(ms, 3), "Mpoints_per_s": round(P / ms / 1e3, 1)}
    print(json.dumps({"dist2": out}))


if __name__ == "__main__":
    main()
