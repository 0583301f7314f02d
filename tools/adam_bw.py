#!/usr/bin/env python
"""Bandwidth of the Adam launch on the ten Gaussian groups of a C4-sized model (3 M Gaussians, 67 floats each), with the
declared stage-2 gradient set (seven groups without a gradient tensor): streaming (non-temporal) against plain loads / stores."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("gi-gs_amd")
import torch  # noqa: E402

import gigs_lib  # noqa: E402


def main():
    lib = gigs_lib.lib()
    dev = torch.device("cuda:0")
    P = int(os.environ.get("P", 3_000_000))
    rows = dict(xyz=3, f_dc=3, f_rest=45, opacity=1, normal=3, albedo=3, roughness=1, metallic=1, scaling=3, rotation=4)
    with_grad = ("albedo", "roughness", "metallic")
    t = {}
    for k, r in rows.items():
        t[k] = [torch.randn(P, r, device=dev) for _ in range(2)] + [torch.rand(P, r, device=dev) * 1e-4]
        if k in with_grad:
            t[k].append(torch.randn(P, r, device=dev) * 1e-3)
    groups = (gigs_lib.AdamGroup * len(rows))(*[
        gigs_lib.AdamGroup(v[0].data_ptr(), v[3].data_ptr() if len(v) > 3 else None, v[1].data_ptr(), v[2].data_ptr(), v[0].numel(), 0.0, 0)
        for v in t.values()])
    tab = torch.tensor([[1e-3, 0.9]] * len(rows), dtype=torch.float32, device=dev)
    nbytes = sum(v[0].numel() * 4 * (6 + (1 if len(v) > 3 else 0)) for v in t.values())
    s = torch.cuda.current_stream().cuda_stream
    for flag, name in ((0x100, "plain"), (0, "streaming"), (0x100, "plain"), (0, "streaming")):
        for _ in range(3):
            gigs_lib.check(lib.gigs_adam_step_dyn(len(rows), C.cast(groups, C.c_void_p), 0.9, 0.999, 1e-15, flag, tab.data_ptr(), s), "adam")
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        n = 20
        for _ in range(n):
            gigs_lib.check(lib.gigs_adam_step_dyn(len(rows), C.cast(groups, C.c_void_p), 0.9, 0.999, 1e-15, flag, tab.data_ptr(), s), "adam")
        b.record()
        b.synchronize()
        ms = a.elapsed_time(b) / n
        print("%-9s loads / stores: %.3f ms per launch, %.2f GB moved -> %.2f TB/s" % (name, ms, nbytes / 1e9, nbytes / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
