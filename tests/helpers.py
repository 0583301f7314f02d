"""Shared helpers for the parity tests."""
import numpy as np

import scenes

GAUSS_KEYS = ["means3D", "opacities", "normal", "albedo", "roughness", "metallic", "shs", "scales", "rotations"]


def oracle_forward(orc, sc, cam, bg=(0.0, 0.0, 0.0), **kw):
    r = orc.Rasterizer()
    args = {k: sc[k] for k in GAUSS_KEYS}
    args.update(kw)
    out = r.forward(bg=np.asarray(bg, np.float32), sh_degree=sc["sh_degree"], viewmatrix=cam["viewmatrix"],
                    projmatrix=cam["projmatrix"], campos=cam["campos"], tanfovx=cam["tanfovx"],
                    tanfovy=cam["tanfovy"], image_height=cam["image_height"], image_width=cam["image_width"],
                    **args)
    return r, out


def random_pix_grads(rng, H, W):
    return dict(color=rng.normal(size=(3, H, W)).astype(np.float32),
                opacity=rng.normal(size=(1, H, W)).astype(np.float32),
                depth=rng.normal(size=(1, H, W)).astype(np.float32),
                normal=rng.normal(size=(3, H, W)).astype(np.float32),
                albedo=rng.normal(size=(3, H, W)).astype(np.float32),
                roughness=rng.normal(size=(1, H, W)).astype(np.float32),
                metallic=rng.normal(size=(1, H, W)).astype(np.float32))


def focal(cam):
    return (cam["image_width"] / (2.0 * cam["tanfovx"]), cam["image_height"] / (2.0 * cam["tanfovy"]))


def small_scene(P=300, sh_degree=1, seed=3, W=64, H=48, scale_mu=0.08, view=0):
    sc = scenes.random_scene(P=P, sh_degree=sh_degree, seed=seed, scale_mu=scale_mu)
    cam = scenes.orbit_camera(view, 4, W, H)
    return sc, cam


def set_options(monkeypatch, **kw):
    """Switch gigs_options for the rest of the test: the current library context of this thread is replaced by a derived
    one (gigs_lib.Context.derive) and restored by monkeypatch at teardown.  Option names are gigs_options members
    (include/gigs_hip.h); `async_binning=(capacity, counters)` / `False` and `blend_event=` are accepted too."""
    import gigs_lib
    monkeypatch.setattr(gigs_lib._tls, "ctx", gigs_lib.current().derive(**kw))
