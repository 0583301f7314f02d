"""ctypes binding of libgigs_hip.so (C ABI in include/gigs_hip.h).

This is the only place the product talks to native code.  There is NO fallback: if the
shared library is missing or a call fails, an exception is raised -- the oracle under
oracle/ is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GIGS_LIB", os.path.join(_HERE, "libgigs_hip.so"))  # GIGS_LIB: tuning builds

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)

_f = C.c_void_p  # device pointers are passed as integers
_i = C.c_int
_fl = C.c_float

# name -> (restype, argtypes); must list every symbol declared in include/gigs_hip.h
SIGNATURES = {
    "gigs_last_error": (C.c_char_p, []),
    "gigs_build_arch": (C.c_char_p, []),
    "gigs_required_geom": (C.c_size_t, [_i]),
    "gigs_required_image": (C.c_size_t, [_i, _i]),
    "gigs_required_binning": (C.c_size_t, [_i]),
    "gigs_forward": (_i, [C.c_void_p, ALLOC_FN, C.c_void_p, ALLOC_FN, C.c_void_p, ALLOC_FN, C.c_void_p,
                          _i, _i, _i, _f, _i, _i,            # P D M background width height
                          _f, _f, _f, _f, _f, _f, _f, _f,    # means3D shs colors opac normal albedo rough metal
                          _f, _fl, _f, _f,                   # scales scale_modifier rotations cov3D
                          _f, _f, _f, _fl, _fl,              # view proj campos tanfovx tanfovy
                          _i, _i, _i,                        # prefiltered argmax_depth inference
                          _f, _f, _f, _f, _f, _f, _f, _f, _f,  # 9 output planes
                          _f, _i, C.c_void_p]),              # radii debug stream
    "gigs_backward": (_i, [C.c_void_p, _i, _i, _i, _i, _f, _i, _i,       # P D M R background width height
                           _f, _f, _f, _f, _f, _f, _f,       # means3D shs colors normal albedo rough metal
                           _f, _f, _f, _f, _f, _f, _f,       # scales rotations cov3D view proj campos radii
                           _fl, _fl, _fl,                    # scale_modifier tanfovx tanfovy
                           _f, _f, _f,                       # geom binning image buffers
                           _f, _f, _f, _f, _f, _f, _f,       # 7 incoming grads
                           _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f,  # 14 outputs
                           _i, C.c_void_p]),
    "gigs_lite_forward": (_i, [C.c_void_p, ALLOC_FN, C.c_void_p, ALLOC_FN, C.c_void_p, ALLOC_FN, C.c_void_p,
                               _i, _i, _i, _f, _i, _i,            # P D M background width height
                               _f, _f, _f, _f,                    # means3D shs colors opacities
                               _f, _fl, _f, _f,                   # scales scale_modifier rotations cov3D
                               _f, _f, _f, _fl, _fl,              # view proj campos tanfovx tanfovy
                               _i, _i,                            # prefiltered argmax_depth
                               _f, _f, _f, _f, _i, C.c_void_p]),  # color opacity depth radii debug stream
    "gigs_ssr_backward": (_i, [_i, _i, _f, _f, _f, C.c_void_p]),
    "gigs_mark_visible": (_i, [_i, _f, _f, _f, _f, C.c_void_p]),
    "gigs_depth_to_normal": (_i, [_i, _i, _fl, _fl, _f, _f, _f, _f, C.c_void_p]),
    "gigs_derive_normal": (_i, [_i, _i, C.c_float, C.c_float, _f, _f, C.c_float, C.c_float, C.c_float, _f, _f, C.c_void_p]),
    "gigs_ssao": (_i, [_i, _i, _fl, _fl, _fl, _fl, _fl, _fl, _i, _i, _f, _f, _f, C.c_void_p]),
    "gigs_ssr": (_i, [_i, _i, _fl, _fl, _fl, _fl, _fl, _fl, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f,
                      C.c_void_p]),
    "gigs_gi_scratch_bytes": (C.c_size_t, [_i, _i]),
    "gigs_ssao_ex": (_i, [C.c_void_p, _i, _i, _fl, _fl, _fl, _fl, _fl, _fl, _i, _i, _f, _f, _f, _f, C.c_void_p]),
    "gigs_ssr_ex": (_i, [C.c_void_p, _i, _i, _fl, _fl, _fl, _fl, _fl, _fl, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f,
                         C.c_void_p]),
    "gigs_ssr_hits": (_i, [C.c_void_p, _i, _i, _fl, _fl, _fl, _fl, _fl, _fl, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f,
                           _i, _f, _f, _f, C.c_uint, _f, C.c_void_p]),
    "gigs_ssr_apply": (_i, [_i, _i, _fl, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_median3x3": (_i, [_i, _i, _i, _f, _f, C.c_void_p]),
    "gigs_median3x3_backward": (_i, [_i, _i, _i, _f, _f, _f, C.c_void_p]),
    "gigs_bilateral3x3": (_i, [_i, _i, _i, _fl, _fl, _fl, _f, _f, C.c_void_p]),
    "gigs_diffuse_cubemap_fwd": (_i, [_i, _f, _f, C.c_void_p]),
    "gigs_diffuse_cubemap_bwd": (_i, [_i, _f, _f, C.c_void_p]),
    "gigs_specular_bounds": (_i, [_i, _fl, _f, C.c_void_p]),
    "gigs_specular_cubemap_fwd": (_i, [_i, _f, _f, _fl, _fl, _f, C.c_void_p]),
    "gigs_specular_cubemap_bwd": (_i, [_i, _f, _f, _fl, _fl, _f, C.c_void_p]),
    "gigs_specular_weights": (_i, [_i, _f, _f, _fl, _fl, _i, _f, C.c_void_p]),
    "gigs_specular_weights_divide": (_i, [_i, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_specular_cubemap_fwd_w": (_i, [C.c_void_p, _i, _f, _f, _f, _f, _i, _f, _f, C.c_void_p]),
    "gigs_specular_cubemap_bwd_w": (_i, [C.c_void_p, _i, _f, _f, _f, _i, _f, _i, _f, C.c_void_p]),
    "gigs_specular_cubemap_multi_w": (_i, [C.c_void_p, _i, C.c_void_p, _i, C.c_void_p]),
    "gigs_cubemap_mip_fwd": (_i, [_i, _i, _f, _f, C.c_void_p]),
    "gigs_cubemap_mip_bwd_add": (_i, [_i, _f, _f, _f, C.c_void_p]),
    "gigs_cubemap_mip_bwd_add2": (_i, [_i, _f, _f, _f, _f, C.c_void_p]),
    "gigs_cubemap_mip_bwd": (_i, [_i, _f, _f, C.c_void_p]),
    "gigs_shade_fwd": (_i, [_i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, C.POINTER(C.c_void_p),
                            C.POINTER(C.c_int), _f, _i, _i, _i, _i, _f, _f, _f, _f, C.c_void_p]),
    "gigs_shade_bwd": (_i, [_i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, C.POINTER(C.c_void_p),
                            C.POINTER(C.c_int), _f, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f,
                            C.POINTER(C.c_void_p), C.c_void_p]),
    "gigs_shade_fwd_ex": (_i, [C.c_void_p, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, C.POINTER(C.c_void_p),
                               C.POINTER(C.c_int), _f, _i, _i, _i, _i, _f, _f, _f, _f, C.c_void_p, C.c_void_p]),
    "gigs_shade_bwd_ex": (_i, [C.c_void_p, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, C.POINTER(C.c_void_p),
                               C.POINTER(C.c_int), _f, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f,
                               C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "gigs_gbuffer_post": (_i, [_i, _i, _f, _f, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_gbuffer_post_bwd": (_i, [_i, _i, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_normalize_mask": (_i, [_i, _i, _f, _f, _f, C.c_void_p]),
    "gigs_nonzero_mask": (_i, [_i, _i, _f, _f, C.c_void_p]),
    "gigs_stage2_loss_fwd": (_i, [_i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_stage2_loss_bwd": (_i, [_i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_stage2_loss_fwd_grad": (_i, [_i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_cube_texture_fwd": (_i, [_i, _f, _i, _f, _f, _i, C.c_void_p]),
    "gigs_cube_texture_bwd": (_i, [_i, _i, _f, _f, _f, _i, C.c_void_p]),
    "gigs_cube_taps": (_i, [_i, _i, _f, _f, _f, C.c_void_p]),
    "gigs_cube_texture_bwd_gather": (_i, [_i, _i, _i, _f, _f, _f, _i, _i, _f, _f, _f, C.c_void_p]),
    "gigs_latlong_to_cubemap": (_i, [_i, _i, _i, _i, _i, _f, _f, C.c_void_p]),
    "gigs_loss_scratch_floats": (C.c_size_t, [_i, _i, _i]),
    "gigs_l1_ssim_fwd": (_i, [_i, _i, _i, _f, _f, C.c_float, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_l1_ssim_bwd": (_i, [_i, _i, _i, _f, _f, C.c_float, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_tv_loss_fwd": (_i, [_i, _i, _i, _i, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_tv_loss_bwd": (_i, [_i, _i, _i, _i, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_masked_l1_fwd": (_i, [_i, _i, _i, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_masked_l1_bwd": (_i, [_i, _i, _i, _f, _f, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_adam_step": (_i, [_i, C.c_void_p, C.c_double, C.c_double, C.c_double, _i, C.c_void_p]),
    "gigs_adam_step_dyn": (_i, [_i, C.c_void_p, C.c_double, C.c_double, C.c_double, _i, _f, C.c_void_p]),
    "gigs_adam_scalars": (None, [C.c_double, _i, C.c_double, C.c_double, C.POINTER(C.c_float)]),
    "gigs_adam_step_watch": (_i, [_i, C.c_void_p, C.c_double, C.c_double, C.c_double, _i, _f, C.c_char_p, _f, C.c_void_p]),
    "gigs_adam_step_guarded": (_i, [_i, C.c_void_p, C.c_double, C.c_double, C.c_double, _i, _f, C.c_char_p, _f, _f, C.c_void_p]),
    "gigs_ctx_set_reuse_binning": (_i, [C.c_void_p, _i]),
    "gigs_ctx_set_materials_only": (_i, [C.c_void_p, C.c_void_p]),
    "gigs_ctx_set_split_sh": (_i, [C.c_void_p, C.c_void_p]),
    "gigs_activate_fwd": (_i, [_i, _i, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gigs_activate_bwd": (_i, [_i, _i, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gigs_densify_stats": (_i, [_i, _f, _f, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_gather_rows": (_i, [_i, C.c_void_p, C.c_longlong, C.c_longlong, _f, _f, C.c_void_p]),
    "gigs_dist2_scratch_bytes": (C.c_size_t, [_i]),
    "gigs_dist2": (_i, [_i, _f, _f, _f, C.c_size_t, C.c_void_p]),
    "gigs_geom_offset": (C.c_longlong, [_i, _i]),
    "gigs_binning_offset": (C.c_longlong, [_i, _i]),
    "gigs_image_offset": (C.c_longlong, [_i, _i, _i]),
    "gigs_selftest_div2": (_i, [_i, _f, _f, _f, _f, _f, _f, C.c_void_p]),
    "gigs_selftest_round": (_i, [_f, C.c_void_p]),
    "gigs_ctx_set_blend_begin_event": (_i, [C.c_void_p, C.c_void_p]),
    "gigs_stream_delay": (_i, [C.c_uint, C.c_void_p]),
    "gigs_ctx_set_async_binning": (_i, [C.c_void_p, _i, C.c_void_p]),
    "gigs_ctx_create": (C.c_void_p, []),
    "gigs_ctx_destroy": (None, [C.c_void_p]),
    "gigs_ctx_get_options": (_i, [C.c_void_p, C.c_void_p]),
    "gigs_ctx_set_options": (_i, [C.c_void_p, C.c_void_p]),
    "gigs_profile_begin": (None, []),
    "gigs_profile_end": (_i, [C.POINTER(C.c_float), C.POINTER(C.c_int), _i]),
    "gigs_profile_stage_name": (C.c_char_p, [_i]),
}

class ShadeExt(C.Structure):
    """gigs_shade_ext of include/gigs_hip.h."""
    _fields_ = [("planar", C.c_int), ("rough_scale", C.c_float), ("rough_bias", C.c_float),
                ("out_F0", C.c_void_p), ("out_linear", C.c_void_p), ("out_roughness", C.c_void_p),
                ("g_albedo_mul_a", C.c_void_p), ("g_albedo_mul_b", C.c_void_p),
                ("g_roughness_add", C.c_void_p), ("g_metallic_add", C.c_void_p),
                ("g_scale", C.c_void_p), ("lamb_mask", C.c_void_p), ("lamb_acc4", C.c_void_p), ("part", C.c_int)]


class Options(C.Structure):
    """gigs_options of include/gigs_hip.h."""
    _fields_ = [(n, C.c_int) for n in (
        "struct_bytes", "binning_legacy", "bucket_max_mean", "long_lists", "bucket_target", "bin_bands", "blend_cull",
        "pre_bwd_sh_skip", "gi_march", "gi_cert", "gi_interleave", "gi_tile_log2w", "gi_zero_rays", "spec_max8",
        "spec_max16", "shade_lds_floats", "shade_bwd_blocks")]


OPTION_NAMES = tuple(n for n, _ in Options._fields_ if n != "struct_bytes")
GI_MARCHES = ("exact", "hoist", "hoist_fma", "proj_nr", "proj")


class SpecLevel(C.Structure):
    """gigs_spec_level of include/gigs_hip.h."""
    _fields_ = [("res", C.c_int), ("avg_window", C.c_int), ("src", C.c_void_p), ("bounds", C.c_void_p),
                ("offsets", C.c_void_p), ("weights", C.c_void_p), ("dst", C.c_void_p), ("wsum", C.c_void_p)]


class AdamGroup(C.Structure):
    """gigs_adam_group of include/gigs_hip.h."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_longlong), ("lr", C.c_double), ("step", C.c_int)]


class ActivationRaw(C.Structure):
    """gigs_activation_raw / gigs_activation_raw_grad of include/gigs_hip.h (nine pointers)."""
    _fields_ = [(n, C.c_void_p) for n in ("f_dc", "f_rest", "opacity", "normal", "albedo", "roughness", "metallic",
                                          "scaling", "rotation")]


class ActivationOut(C.Structure):
    """gigs_activation_out of include/gigs_hip.h (eight pointers)."""
    _fields_ = [(n, C.c_void_p) for n in ("shs", "opacities", "normal", "albedo", "roughness", "metallic", "scales",
                                          "rotations")]


class GatherTensor(C.Structure):
    """gigs_gather_tensor of include/gigs_hip.h."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_floats", C.c_int), ("zero_new", C.c_int)]


_lib = None


class GigsError(RuntimeError):
    pass


def lib():
    """Load libgigs_hip.so (built by gi-gs_amd/build.py); raises if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python gi-gs_amd/build.py` "
                "(hipcc, --offload-arch=gfx950). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str) -> int:
    if rc < 0:
        msg = lib().gigs_last_error().decode("utf-8", "replace")
        raise GigsError(f"{what} failed ({rc}): {msg}")
    return rc


class profile:
    """Context manager around gigs_profile_begin/end; `.stages` = {name: (total_ms, launches)}."""

    def __enter__(self):
        lib().gigs_profile_begin()
        self.stages = {}
        return self

    def __exit__(self, *exc):
        n = 64
        ms = (C.c_float * n)()
        cnt = (C.c_int * n)()
        k = lib().gigs_profile_end(ms, cnt, n)
        for i in range(min(k, n)):
            if cnt[i]:
                self.stages[lib().gigs_profile_stage_name(i).decode()] = (float(ms[i]), int(cnt[i]))
        return False


# ------------------------------------------------------------------------------------------
# contexts (gigs_ctx): per-instance state of the library -- options, asynchronous binning, scheduling event
# ------------------------------------------------------------------------------------------
class Context:
    """An immutable gigs_ctx.  Contexts are interned by their complete settings (`Context.get`): asking twice for the same
    options / asynchronous-binning buffer / event yields the same native object, so deriving one per call costs a
    dictionary lookup.  `ptr` is what the C entry points take first (None = the library's default context)."""

    _cache: dict = {}
    _lock = threading.Lock()

    def __init__(self, opts: tuple, async_capacity: int, async_counters, blend_event, reuse_binning: bool = False,
                 materials_only=None, sh_rest=None):
        l = lib()
        self.opts, self.async_capacity = tuple(opts), int(async_capacity)
        self.async_counters, self.blend_event = async_counters, blend_event  # kept alive with the context
        self.reuse_binning = bool(reuse_binning)
        self.materials_only = materials_only  # device int32[1] violation counter (gigs_ctx_set_materials_only), kept alive
        self.sh_rest = sh_rest  # the optimizer's _features_rest tensor [P, M-1, 3] (gigs_ctx_set_split_sh), kept alive
        self.ptr = l.gigs_ctx_create()
        if not self.ptr:
            raise GigsError("gigs_ctx_create failed")
        o = Options()
        o.struct_bytes = C.sizeof(Options)
        for n, v in zip(OPTION_NAMES, self.opts):
            setattr(o, n, int(v))
        check(l.gigs_ctx_set_options(self.ptr, C.byref(o)), "gigs_ctx_set_options")
        if self.async_capacity > 0:
            check(l.gigs_ctx_set_async_binning(self.ptr, self.async_capacity,
                                               None if async_counters is None else async_counters.data_ptr()),
                  "gigs_ctx_set_async_binning")
        if blend_event is not None:
            check(l.gigs_ctx_set_blend_begin_event(self.ptr, blend_event.cuda_event), "gigs_ctx_set_blend_begin_event")
        if self.reuse_binning:
            check(l.gigs_ctx_set_reuse_binning(self.ptr, 1), "gigs_ctx_set_reuse_binning")
        if materials_only is not None:
            check(l.gigs_ctx_set_materials_only(self.ptr, materials_only.data_ptr()), "gigs_ctx_set_materials_only")
        if sh_rest is not None:
            check(l.gigs_ctx_set_split_sh(self.ptr, sh_rest.data_ptr()), "gigs_ctx_set_split_sh")

    def __del__(self):
        try:
            if self.ptr:
                lib().gigs_ctx_destroy(self.ptr)  # host memory only; queued work does not reference the context
                self.ptr = None
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    @classmethod
    def get(cls, opts: tuple, async_capacity: int = 0, async_counters=None, blend_event=None, reuse_binning=False,
            materials_only=None, sh_rest=None) -> "Context":
        key = (tuple(int(v) for v in opts), int(async_capacity),
               None if async_counters is None else async_counters.data_ptr(), None if blend_event is None else id(blend_event),
               bool(reuse_binning), None if materials_only is None else materials_only.data_ptr(),
               None if sh_rest is None else (sh_rest.data_ptr(), tuple(sh_rest.shape)))
        with cls._lock:
            c = cls._cache.get(key)
            if c is None:
                if len(cls._cache) > 512:  # contexts of long-gone buffers: start over (live ones are re-created on demand)
                    cls._cache.clear()
                c = cls._cache[key] = cls(key[0], async_capacity, async_counters, blend_event, reuse_binning, materials_only, sh_rest)
        return c

    def derive(self, async_binning=None, blend_event="keep", reuse_binning=None, materials_only="keep", sh_rest="keep",
               **options) -> "Context":
        """The context with some settings changed: option names of gigs_options (gi_march also by name), `async_binning` =
        (capacity, counters tensor) or False to switch it off, `blend_event` = a torch.cuda.Event or None, `reuse_binning`
        = True / False (gigs_ctx_set_reuse_binning), `materials_only` = a device int32[1] violation counter or None
        (gigs_ctx_set_materials_only), `sh_rest` = the optimizer's _features_rest tensor or None (gigs_ctx_set_split_sh)."""
        opts = list(self.opts)
        for k, v in options.items():
            if k == "gi_march" and isinstance(v, str):
                v = GI_MARCHES.index(v)
            opts[OPTION_NAMES.index(k)] = int(v)
        cap, cnt = self.async_capacity, self.async_counters
        if async_binning is False:
            cap, cnt = 0, None
        elif async_binning is not None:
            cap, cnt = async_binning
        ev = self.blend_event if isinstance(blend_event, str) else blend_event
        mo = self.materials_only if isinstance(materials_only, str) else materials_only
        rest = self.sh_rest if isinstance(sh_rest, str) else sh_rest
        return Context.get(tuple(opts), cap, cnt, ev, self.reuse_binning if reuse_binning is None else reuse_binning, mo, rest)

    def option(self, name: str) -> int:
        return self.opts[OPTION_NAMES.index(name)]


_default_opts = None
_tls = threading.local()

# Objects with independent work the rasterizer's forward may start beside its blend kernel (pbr.light.CubemapLight registers
# itself: its pre-filter is prefetched on a side stream, see CubemapLight.prefetch).  The protocol: `wants_prefetch()` ->
# bool, `prefetch(step_event, blend_event)`.
import weakref  # noqa: E402

prefetchers = weakref.WeakSet()


def default_options() -> tuple:
    """The library's default options: the GIGS_* environment as it was when the library first read it."""
    global _default_opts
    if _default_opts is None:
        o = Options()
        o.struct_bytes = C.sizeof(Options)
        check(lib().gigs_ctx_get_options(None, C.byref(o)), "gigs_ctx_get_options")
        _default_opts = tuple(getattr(o, n) for n in OPTION_NAMES)
    return _default_opts


def current() -> Context:
    """The context the Python operators of this thread pass to the library."""
    c = getattr(_tls, "ctx", None)
    if c is None:
        c = _tls.ctx = Context.get(default_options())
    return c


def ctx_ptr():
    return current().ptr


class use:
    """`with use(ctx):` -- the operators called inside (on this thread) run with `ctx`."""

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def __enter__(self):
        self._prev = current()
        _tls.ctx = self.ctx
        return self.ctx

    def __exit__(self, *exc):
        _tls.ctx = self._prev
        return False


def options(**kw) -> use:
    """`with options(gi_march="exact", blend_cull=0):` -- the current context with these gigs_options changed."""
    return use(current().derive(**kw))


def set_options(**kw) -> Context:
    """The same, for good (scripts and tools): replaces this thread's current context by the derived one."""
    _tls.ctx = current().derive(**kw)
    return _tls.ctx


def with_forward_context(backward):
    """Decorator for `torch.autograd.Function.backward` (below `@staticmethod`).  autograd runs backward nodes on its own
    device thread, where this module's per-thread current context is not the caller's: a forward stores
    `ctx.lib_ctx = gigs_lib.current()` and the decorated backward runs under `use(ctx.lib_ctx)` -- the backward belongs to the
    library context its forward ran with, whichever thread executes it."""
    @functools.wraps(backward)
    def wrapper(ctx, *grads):
        lc = getattr(ctx, "lib_ctx", None)
        if lc is None:
            return backward(ctx, *grads)
        with use(lc):
            return backward(ctx, *grads)
    return wrapper
