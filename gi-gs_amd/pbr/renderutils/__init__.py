"""Drop-in for `pbr.renderutils` restricted to what GI-GS imports from it (pbr/light.py:10)."""
from .ops import diffuse_cubemap, specular_cubemap

__all__ = ["diffuse_cubemap", "specular_cubemap"]
