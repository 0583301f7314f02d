"""Drop-in for the reference's `pbr` package (pbr/__init__.py:1-4): same names, HIP kernels behind."""
from .light import CubemapLight
from .shade import get_brdf_lut, pbr_shading, saturate_dot, linear_to_srgb, aces_film

__all__ = ["CubemapLight", "get_brdf_lut", "pbr_shading", "saturate_dot", "linear_to_srgb", "aces_film"]
