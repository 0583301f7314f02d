#!/usr/bin/env python
"""Launches the SSAO kernel of the C2 bench view a few times in one march configuration (for rocprofv3 --pmc passes).
    python3 tools/gi_pmc_driver.py <exact|hoist_fma|proj> <cert 0|1>"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
importlib.import_module("gi-gs_amd")
import torch  # noqa: E402

import gi_variants  # noqa: E402
import scenes  # noqa: E402

mode, cert = sys.argv[1], sys.argv[2]
sc = scenes.surface_scene(P=300_000, sh_degree=2, seed=0)
cam = scenes.orbit_camera(5, 64, 800, 800, radius=3.5)
gi = scenes.GI_DEFAULTS
gb = gi_variants.gbuffer(sc, cam, gi, 2)
import gigs_lib  # noqa: E402
gigs_lib.set_options(gi_cert=int(cert))
_, t = gi_variants.run_mode(mode, gb, cam, gi, 3)
torch.cuda.synchronize()
print(mode, cert, t)
