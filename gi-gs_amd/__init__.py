"""gigs-hip: MI355X-native implementation of the GI-GS rasterizer hot path.

The directory is a source root rather than an importable package name (it contains a hyphen):
importing it (``importlib.import_module("gi-gs_amd")``) puts it on ``sys.path`` so that the
drop-in packages keep the reference's import names:

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, Gaussian_SSR
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
