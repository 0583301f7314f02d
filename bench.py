#!/usr/bin/env python
"""bench.py -- train-step renders/sec (fwd+bwd, G-buffer + indirect) at 800x800; PSNR vs the CPU oracle.

    python bench.py --gpus 1 --steps 20 --warmup 5                       # BASELINE configs[1] (C2), the headline
    python bench.py --config c3                                          # configs[2]: relight inference, PBR + indirect
    python bench.py --config c4                                          # configs[3]: 3 M Gaussians, SH 3, 1237x822
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W           # configs[4] with --config c5 (1297x840)

One "step" is one stage-2 training iteration of the reference (train.py:266-422) on one
camera view per GPU: rasterizer forward (preprocess, scan, duplicate, radix sort, ranges,
G-buffer blend), the in-operator filters + depth->normal + SSAO, the deferred shade,
SSR (indirect diffuse), the L1 loss, and the full backward (shade/SSR closed form, blend
backward, preprocess backward).  No optimizer, no data loading.

Workload = BASELINE.json configs[1] stand-in (the real TensoIR data is not available
offline): ~300k synthetic Gaussians on surfaces, 800x800, SH degree 2, GI step=16
delta=0.0625, start=8 (the CLI default of the reference; --start 64 gives the README
setting where the march loop is empty).  With N GPUs every rank renders a different view of
the replicated scene and the parameter gradients are summed with one RCCL all-reduce per
step (weak scaling: one view per GPU per step).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# torch, numpy and the package are imported by run() -- AFTER the arguments are parsed and, for --gpus N > 1 without a
# launcher, after the N ranks have been started as a child torchrun (no GPU call may precede that)
np = torch = dp = gigs_lib = pipeline = scenes = None

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
PARAM_KEYS = ["means3D", "opacities", "normal", "albedo", "roughness", "metallic", "shs", "scales", "rotations"]


def _imports():
    global np, torch, dp, gigs_lib, pipeline, scenes
    importlib.import_module("gi-gs_amd")
    import numpy as _np
    import torch as _torch

    import dp as _dp
    import gigs_lib as _gigs_lib
    import pipeline as _pipeline
    import scenes as _scenes
    np, torch, dp, gigs_lib, pipeline, scenes = _np, _torch, _dp, _gigs_lib, _pipeline, _scenes


def algorithmic_bytes(P, V, R, N, M, T):
    """Compulsory HBM bytes per launch of each stage."""
    return {
        "preprocess_fwd": 52 * P + V * (12 * M + 79),
        "scan": 8 * P,
        "duplicate": 20 * V + 12 * R,
        "sort": 24 * R,
        "tile_ranges": 8 * R + 8 * T,
        "blend_fwd": 88 * R + 84 * N,
        "blend_bwd": 40 * R + 60 * N,
        "preprocess_bwd": V * (259 + 24 * M),
        "depth_to_normal": 28 * N,
        "ssao": 28 * N,
        "ssr": 88 * N,
        "median3x3": 8 * N,        # per plane-launch of N pixels: read + write (per channel)
        "bilateral3x3": 24 * N,
        "median3x3_bwd": 12 * N,
        "shade_fwd": 80 * N,
        "shade_bwd": 96 * N,
        # light: 10 launches per build_mips (4 mip, 1 diffuse, 5 GGX); per launch on average: textures
        # read + written (6.3 MB each way over the chain) and the GGX window bounds (96 B per texel)
        "cubemap_fwd": (2 * 6.3e6 + 96 * 6 * (256 ** 2 + 128 ** 2 + 64 ** 2 + 32 ** 2 + 16 ** 2)) / 10,
        "cubemap_bwd": (2 * 6.3e6 + 96 * 6 * (256 ** 2 + 128 ** 2 + 64 ** 2 + 32 ** 2 + 16 ** 2)) / 10,
    }


def pmc_traffic(stage: str):
    """HBM bytes per launch of `stage` from the newest committed rocprofv3 --pmc summary under
    profiles/ (separate FETCH_SIZE / WRITE_SIZE passes; (2*FETCH_SIZE + WRITE_SIZE) * 1024 per
    MI355X_MICROARCH.md -- the x2 is calibrated for wide streaming reads only, so for the
    gather-dominated GI kernels this is an upper estimate).  None if no summary is committed."""
    k = _pmc_entry(stage)
    try:
        return int((2 * k["FETCH_SIZE"]["mean_per_launch"] + k["WRITE_SIZE"]["mean_per_launch"]) * 1024)
    except Exception:  # noqa: BLE001
        return None


_PMC_NAMES = {"ssao": "ssao_kernel", "ssr": "ssr_kernel", "blend_fwd": "blend_fwd_kernel", "blend_bwd": "blend_bwd_kernel",
              "shade_bwd": "shade_bwd_kernel", "shade_fwd": "shade_fwd_kernel", "preprocess_fwd": "preprocess_fwd_kernel",
              "preprocess_bwd": "preprocess_bwd_kernel", "sort": "bin_sort_kernel"}


_PMC_CONFIG = "c2"  # set by run(): the committed PMC summary of THIS configuration is the one quoted


def _pmc_path():
    """Newest committed PMC summary (profiles/rNN/pmc_summary*.json, by name order), preferring the one collected on this
    configuration (…_c2.json / …_c4.json; c3 / c5 run c2's / c4's kernels on the same sizes)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_summary*.json")))
    want = {"c3": "c2", "c5": "c4"}.get(_PMC_CONFIG, _PMC_CONFIG)
    mine = [f for f in files if f.endswith("_%s.json" % want)]
    return (mine or files or [None])[-1]


def _pmc_file():
    f = _pmc_path()
    return os.path.relpath(f, ROOT) if f else None


def _pmc_entry(stage: str):
    """Counters of `stage`'s kernel from the newest committed summary (profiles/rNN/pmc_summary*.json, by name order);
    template instantiations of one kernel are matched by prefix and the one with the most launches is taken."""
    f = _pmc_path()
    if f is None or stage not in _PMC_NAMES:
        return {}
    try:
        d = json.load(open(f))
    except Exception:  # noqa: BLE001
        return {}
    cands = [v for k, v in d.items() if k.startswith(_PMC_NAMES[stage])]
    if not cands:
        return {}
    return max(cands, key=lambda v: max((c.get("launches", 0) for c in v.values()), default=0))


def pmc_valu_busy(stage: str):
    """Fraction of the kernel's cycles in which the VALU pipes were issuing, from the same committed PMC
    summary: SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs, GRBM_GUI_ACTIVE cycles
    summed over the 8 XCDs.  This -- not the HBM fraction -- is the roofline that binds the GI march."""
    k = _pmc_entry(stage)
    try:
        cycles = k["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0
        return round(4.0 * k["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / (1024.0 * cycles), 3)
    except Exception:  # noqa: BLE001
        return None


def pmc_light_roofline():
    """The GGX pre-filter launch (all five levels in one kernel, forward and backward) against the HBM roofline, from the committed
    PMC pass of this configuration: HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md), duration =
    GRBM_GUI_ACTIVE / 8 XCDs at 2.4 GHz.  The kernel streams its cached pair-weight tables (0.74 GB per direction at base 256),
    which is what makes it HBM-bound; the bytes it could not avoid even by recomputing the weights (textures + window bounds)
    are `alg_bytes_without_tables`."""
    f = _pmc_path()
    if f is None:
        return None
    try:
        d = json.load(open(f))
    except Exception:  # noqa: BLE001
        return None
    out = {"pmc_summary": os.path.relpath(f, ROOT), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "alg_bytes_without_tables": int(2 * 6.3e6 + 96 * 6 * (256 ** 2 + 128 ** 2 + 64 ** 2 + 32 ** 2 + 16 ** 2))}
    for tag, key in (("fwd", "specular_apply_multi_kernel<false>"), ("bwd", "specular_apply_multi_kernel<true>")):
        cands = [v for k, v in d.items() if k.startswith(key)]
        if not cands:
            continue
        v = cands[0]
        try:
            byts = (2 * v["FETCH_SIZE"]["mean_per_launch"] + v["WRITE_SIZE"]["mean_per_launch"]) * 1024
            ms = v["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0 / 2.4e6
            out[tag] = {"hbm_bytes": int(byts), "ms_alone": round(ms, 4), "achieved": round(byts / (ms * 1e-3) / 1e9, 1),
                        "frac": round(byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "valu_busy": round(4.0 * v["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / (1024.0 * v["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0), 3)}
        except Exception:  # noqa: BLE001
            pass
    return out if ("fwd" in out or "bwd" in out) else None


def make_light(device, shade: str = "hip"):
    import pbr
    light = pbr.CubemapLight(base_res=256).to(device)
    return light, pbr.get_brdf_lut().to(device)


def cpu_baseline_and_parity(sc, cam, gi, sh_degree, light, brdf_lut, stepper, config, single_thread_res=0):
    """The checker leg (rank 0, N = 1 only): ONE view of the timed workload on the CPU oracle -- rasterizer fwd, the
    operator's filters + SSAO, light pre-filter, shade, SSR, sRGB/median, rasterizer bwd (oracle/stage2_ref.py; no
    shade / light backward on the CPU, so the CPU time is a lower bound of a full step) -- timed on all host cores,
    and the same view from the timed GPU path compared with it (oracle/parity.py): integer state bit for bit,
    every fp plane's mean L1 and the share of its elements beyond 1e-4 / 1e-5, rasterizer-backward gradients, the PSNR
    (utils/image_utils.py:31) of the final stage-2 image, K (evaluated pixel-Gaussian pairs) and the covered-pixel
    fraction; beside it the same per-pixel figures for the product's EXACT march and for the oracle against its
    FMA-contracted twin (parity.march_noise).  `single_thread_res` > 0 adds a single-thread timing on a bounded sample:
    the same scene and camera at that (square) resolution."""
    from oracle import oracle as orc
    from oracle import parity
    orc.build()
    cores = orc.max_threads()
    orc.set_threads(cores)
    only = ("albedo", "roughness", "metallic")  # the planes stage 2 differentiates
    dev = str(light.base.device)
    gpu = parity.gpu_capture(sc, cam, gi, sh_degree, light=light, brdf_lut=brdf_lut, stepper=stepper, grads_only=only, dev=dev)
    ref, t = parity.oracle_capture(orc, sc, cam, gi, sh_degree, light_base=gpu.get("light_base"), grads_only=only)
    rep = parity.compare(gpu, ref)
    rep["gi_per_pixel"] = parity.march_noise(orc, sc, cam, gi, sh_degree, gpu, ref, light=light, brdf_lut=brdf_lut, dev=dev)
    cpu = dict(value=round(1.0 / t["total"], 4), unit="renders/s", cores=cores, kind="port",
               sample="1 step of the same workload (%s, 1 view) on the oracle: " % config
                      + ", ".join("%s %.2fs" % (k, v) for k, v in t.items() if k != "total")
                      + "; backward blend is single-threaded (deterministic double sums), no shade/light backward on the CPU")
    if single_thread_res:
        # bounded single-thread sample (SURVEY 8(d) asks for both timings; the full-size view would take minutes on one
        # core): the SAME scene through the same oracle sequence at a reduced square resolution, light pre-filter
        # (resolution-independent: 6 x 256^2 texels, minutes on one core) excluded and said so
        import numpy as _np
        if "eye_target" in cam:  # a --scene view: the same pose at the reduced resolution
            small = scenes.look_at_camera(cam["eye_target"][0], cam["eye_target"][1], single_thread_res, single_thread_res, cam["fovx"])
        else:
            small = scenes.orbit_camera(cam.get("index", 0), cam.get("n_views", 64), single_thread_res, single_thread_res, radius=3.5)
        orc.set_threads(1)
        t1 = {}
        t0 = time.perf_counter()
        raw = parity.stage2_ref.operator_forward(orc, sc, small, gi, sh_degree, keep_state=True, timings=t1)
        tb = time.perf_counter()
        pg = parity.pixel_grads(single_thread_res, single_thread_res, only=only)
        raw["rasterizer"].backward(**{"grad_" + k: v for k, v in pg.items()})
        t1["backward"] = time.perf_counter() - tb
        F0 = _np.full((3, single_thread_res, single_thread_res), 0.04, _np.float32)
        ts = time.perf_counter()
        fx, fy = parity.stage2_ref.focal(small)
        orc.ssr(single_thread_res, single_thread_res, fx, fy, gi["radius"], gi["bias"], gi["thick"], gi["delta"], gi["step"],
                gi["start"], raw["out_normal_view"], raw["depth_pos"], raw["render"], raw["albedo_map"], raw["roughness_map"],
                raw["metallic_map"], F0)
        t1["ssr"] = time.perf_counter() - ts
        total = time.perf_counter() - t0
        orc.set_threads(cores)
        cpu["single_thread"] = dict(
            value=round(1.0 / total, 4), unit="renders/s", cores=1,
            sample="the same scene, 1 view at %dx%d (%.1f %% of the pixels; all %d Gaussians preprocessed and binned), 1 thread: "
                   % (single_thread_res, single_thread_res, 100.0 * single_thread_res ** 2 / (cam["image_width"] * cam["image_height"]),
                      sc["means3D"].shape[0])
                   + ", ".join("%s %.2fs" % (k, v) for k, v in t1.items())
                   + "; light pre-filter and shade not included (resolution-independent / negligible)")
    return cpu, rep


def cpu_baseline_and_parity_c3(sc, cam, gi, sh_degree, relighter, light, g, view_dirs, cam_t):
    """C3 checker leg: one relit view on the oracle (mips built once, outside the timed view -- relight.py:141)."""
    from oracle import oracle as orc
    from oracle import stage2_ref
    orc.build()
    cores = orc.max_threads()
    orc.set_threads(cores)
    base = light.base.detach().cpu().numpy()
    t0 = time.perf_counter()
    diffuse, spec = stage2_ref.build_mips(orc, base)
    t_mips = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = stage2_ref.relight_view(orc, sc, cam, gi, sh_degree, diffuse, spec, metallic=relighter.metallic)
    t_view = time.perf_counter() - t0
    out = relighter(cam_t, g, view_dirs)
    torch.cuda.synchronize()
    planes = {}
    for k in ("render_direct", "IRR", "render_rgb", "occlusion"):
        a, b = out[k].cpu().numpy(), ref[k]
        d = np.abs(np.nan_to_num(a.astype(np.float64)) - np.nan_to_num(b.astype(np.float64)))
        planes[k] = {"mean_l1": float(d.mean()), "max": float(d.max()),
                     "nan_pattern_equal": bool(np.array_equal(np.isnan(a), np.isnan(b)))}
    rep = dict(planes=planes, worst_plane_mean_l1=max(v["mean_l1"] for v in planes.values()),
               psnr_render_rgb=round(stage2_ref.psnr(np.nan_to_num(out["render_rgb"].cpu().numpy()),
                                                     np.nan_to_num(ref["render_rgb"])), 2))
    cpu = dict(value=round(1.0 / t_view, 4), unit="renders/s", cores=cores, kind="port",
               sample="1 relit view of the same workload on the oracle: %.2fs (light pre-filter, once per run: %.2fs, not "
                      "counted -- the GPU path builds it once outside the timed region too)" % (t_view, t_mips))
    return cpu, rep


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL = _Null()

CONFIGS = {
    # name: (Gaussians, width, height, SH degree, what BASELINE.json calls it)
    "c2": (300_000, 800, 800, 2, "configs[1] TensoIR lego stand-in"),
    "c3": (300_000, 800, 800, 2, "configs[2] TensoIR hotdog relight stand-in (inference, synthetic HDR envmap)"),
    "c4": (3_000_000, 1237, 822, 3, "configs[3] Mip-NeRF360 bicycle images_4 stand-in (--metallic --indirect)"),
    "c5": (3_000_000, 1297, 840, 3, "configs[4] Mip-NeRF360 garden images_4 stand-in, one view per GPU"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed blocks of --steps steps each: the first is the contract's measurement (`value`), all of them "
                         "give `repeats` = median / min / max ms per step")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="BASELINE.json configuration (c2 = the headline metric's; c3 relight inference; c4/c5 3 M Gaussians "
                         "at the Mip-NeRF360 images_4 resolutions)")
    ap.add_argument("--gaussians", type=int, default=None)
    ap.add_argument("--res", type=int, default=None, help="square image (overrides the config's size)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--sh-degree", type=int, default=None)
    ap.add_argument("--start", type=int, default=8, help="GI march start (8 = reference CLI default, 64 = README)")
    ap.add_argument("--shade", choices=["auto", "hip"], default="auto", help="(kept for older command lines; always hip)")
    ap.add_argument("--graphs", choices=["on", "off"], default="on",
                    help="capture the launch-bound glue segments of the step into hipGraphs")
    ap.add_argument("--fused", choices=["on", "off"], default="on",
                    help="run the tensor glue between rasterizer and loss.backward() as the fused stage-2 node "
                         "(gi-gs_amd/stage2_fused.py) instead of op-by-op torch")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle leg (cpu_baseline and parity)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements (drop_in_step, full_grad_step, iteration) that follow the timed region")
    ap.add_argument("--iteration", action="store_true",
                    help="print the complete training iteration (it/s, `iteration` / `iteration_stage1`) beside the metric even "
                         "under --no-extras (it is part of the default line)")
    ap.add_argument("--scene", default=None,
                    help="a trained scene instead of the synthetic stand-in: point_cloud.ply (GaussianModel.save_ply) or "
                         "chkpntN.pth (train.py); Gaussian count and SH degree come from the file (SURVEY 8(d): real "
                         "TensoIR / Mip-NeRF360 data if present on the box)")
    ap.add_argument("--cameras", default=None,
                    help="with --scene: a Blender / NeRF-synthetic transforms_*.json whose poses and field of view replace "
                         "the orbit cameras (rendered at the config's resolution)")
    ap.add_argument("--cpu-single-res", type=int, default=200,
                    help="resolution of the bounded single-thread CPU sample (0 = skip)")
    args = ap.parse_args(argv)
    cP, cW, cH, cD, cname = CONFIGS[args.config]
    args.gaussians = cP if args.gaussians is None else args.gaussians
    args.sh_degree = cD if args.sh_degree is None else args.sh_degree
    args.W = args.width or args.res or cW
    args.H = args.height or args.res or cH
    args.cname = cname
    return args


def load_workload(args, W, H):
    """(scene arrays, camera list, `data` tag).  Default: the synthetic stand-ins of SURVEY 8(d) (seed 0, 64 orbit views).
    --scene: a trained point_cloud.ply / chkpntN.pth through gi-gs_amd/scene_io.py (the reference's own formats,
    scene/gaussian_model.py:397-465, train.py:466-490); --cameras: the poses of a transforms_*.json
    (scene/dataset_readers.py:223-281), else 64 orbit views around the cloud's median at 2.5 x its 90 % extent."""
    import numpy as np
    import scenes
    if not args.scene:
        sc = scenes.surface_scene(P=args.gaussians, sh_degree=args.sh_degree, seed=0)
        n_views = 64
        cams = [dict(scenes.orbit_camera(i, n_views, W, H, radius=3.5), index=i, n_views=n_views) for i in range(n_views)]
        return sc, cams, "synthetic"
    import dataset_readers
    import scene_io
    if not os.path.exists(args.scene):
        raise SystemExit("bench.py --scene: %s does not exist" % args.scene)
    sc = scene_io.load_scene(args.scene)
    args.sh_degree = int(sc["sh_degree"])
    args.gaussians = int(sc["means3D"].shape[0])
    if args.cameras:
        cams = dataset_readers.cameras_from_transforms(args.cameras, W, H)
        tag = "%s, poses of %s (synthetic ground-truth images)" % (args.scene, args.cameras)
    else:
        centre = np.median(sc["means3D"], axis=0)
        extent = float(np.quantile(np.linalg.norm(sc["means3D"] - centre, axis=1), 0.9))
        n_views = 64
        cams = []
        for i in range(n_views):
            az = 2.0 * np.pi * (i + 0.25) / n_views
            eye = centre + 2.5 * extent * np.array([np.cos(0.5) * np.cos(az), np.cos(0.5) * np.sin(az), np.sin(0.5)])
            cams.append(dict(scenes.look_at_camera(eye, centre, W, H, 0.6911), eye_target=(eye, centre)))
        tag = "%s, 64 orbit views (synthetic ground-truth images)" % args.scene
    n = len(cams)
    cams = [dict(c, index=i, n_views=n) for i, c in enumerate(cams)]
    return sc, cams, tag


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` (N > 1) without a launcher around it: start the N ranks as a CHILD
    `python -m torch.distributed.run` (this process has made no GPU call and never will), relay rank 0's JSON line.
    Fails loudly when fewer than N GPUs are visible (GIGS_BENCH_BACKEND=gloo, the rehearsal mode, lets ranks share cards)."""
    backend = os.environ.get("GIGS_BENCH_BACKEND", "nccl")
    import torch as _torch  # device_count() only: it does not initialise the GPU (and nothing is exec'ed from here)
    n = _torch.cuda.device_count()
    if n < 1 or (backend == "nccl" and n < args.gpus):
        raise SystemExit("bench.py --gpus %d: %d GPU(s) visible; one rank per GPU over RCCL needs %d "
                         "(GIGS_BENCH_BACKEND=gloo rehearses the rank logic on fewer cards)" % (args.gpus, n, args.gpus))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, GIGS_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0 or line is None:
        raise SystemExit("bench.py --gpus %d: the %d-rank run failed (exit code %d, %s JSON line)"
                         % (args.gpus, args.gpus, rc, "no" if line is None else "a"))
    print(line, flush=True)
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, argv)
    # everything a library prints to stdout (RCCL's version banner, ...) goes to stderr; the JSON line alone to stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    line = run(args)
    if line is not None:
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    return 0


def run(args):
    global _PMC_CONFIG
    _PMC_CONFIG = args.config
    _imports()
    W, H, cname = args.W, args.H, args.cname
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or let bench.py start them: "
                         "`python bench.py --gpus N`)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path in the product)")
    # GIGS_BENCH_BACKEND=gloo with fewer GPUs than ranks is a rehearsal mode (ranks share cards, the all-reduce
    # goes through the host); the measured configuration is one rank per GPU over RCCL
    backend = os.environ.get("GIGS_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible -- one rank per GPU over RCCL" % (world, n_dev))
    dev_index = local_rank if backend == "nccl" else local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # GIGS_BENCH_FORCE_DIST=1: run the distributed code path (process group, gradient all-reduce, barriers) with a
    # single rank -- a one-GPU rehearsal of everything but the inter-GPU traffic
    force_coll = os.environ.get("GIGS_BENCH_FORCE_DIST", "0") == "1"
    use_dist = world > 1 or force_coll
    dist = None
    ranks_seen = 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

    def init_dist():
        # Called AFTER the first (hipGraph-capturing) step: RCCL's proxy thread issues HIP calls of its own, and a
        # HIP call from another thread while a stream is capturing in the default (global) mode fails the capture.
        nonlocal ranks_seen
        if not use_dist:
            return
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        ranks_seen = dist.get_world_size()
        if ranks_seen != world:
            raise SystemExit("bench.py: the process group has %d ranks, WORLD_SIZE says %d" % (ranks_seen, world))

    shade = "hip"
    gi = dict(scenes.GI_DEFAULTS, start=args.start)
    sc, cams, data_tag = load_workload(args, W, H)
    P, M = sc["means3D"].shape[0], sc["shs"].shape[1]
    g = {k: torch.from_numpy(sc[k]).to(dev).requires_grad_(True) for k in PARAM_KEYS}
    n_views = len(cams)
    cams_t = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, W, device=dev), indexing="ij")
    gt_image = torch.stack([0.5 + 0.3 * torch.sin(6 * xx), 0.5 + 0.3 * torch.cos(5 * yy), 0.4 + 0.2 * xx * yy])
    light, brdf_lut = make_light(dev)
    rays = pipeline.canonical_rays(cams[0], dev)
    view_dirs = [pipeline.view_dirs_for(c, rays, dev) for c in cams_t]  # per-camera constants, like gt_image
    params_light = [p for p in light.parameters()]
    # slab order: the stage-2 trainable set (materials, then the light) last and adjacent, so that it is ONE collective
    SLAB_KEYS = ["means3D", "opacities", "normal", "shs", "scales", "rotations", "albedo", "roughness", "metallic"]
    SINK_NAME = {"opacities": "opacity", "shs": "sh"}
    flat_params = [g[k] for k in SLAB_KEYS] + params_light
    inference = args.config == "c3"
    # What is reduced.  A stage-2 iteration DECLARES its non-zero gradient set: the loss reaches albedo / roughness /
    # metallic and the light only (train.py:330-420; the blend weights are detached from the material planes, SURVEY
    # App. D), every other gradient is an exact zero on every rank -- so that stretch of the slab is the collective
    # (5 of 46 floats per Gaussian + the light), and the premise is VERIFIED on the first two reduced steps
    # (check_rest_zero: one device read each, outside the timed region when --warmup >= 2).  GIGS_BENCH_REDUCE=all
    # reduces every gradient (what a stage-1 iteration needs).
    reduce_mode = os.environ.get("GIGS_BENCH_REDUCE", "trainable")
    reduce_only = None
    if reduce_mode == "trainable" and not inference:
        reduce_only = list(range(SLAB_KEYS.index("albedo"), len(flat_params)))
    # multi-rank: every gradient lives in one persistent flat slab; the rasterizer's backward writes into it directly
    # (dgr.grad_sink), so the all-reduce bucket needs no packing pass
    slab = sink = dgr_mod = None
    if use_dist:
        import diff_gaussian_rasterization as dgr_mod
        slab = dp.GradSlab(flat_params)
        slab.timing = True
        sink = slab.sink([SINK_NAME.get(k, k) for k in SLAB_KEYS])
    stepper = relighter = g_inf = None
    if inference:
        # configs[2]: relight.py -- HDR latlong map -> 256^2 cubemap, build_mips ONCE, then per view
        # render(inference=True) -> pbr_shading -> Gaussian_SSR -> sRGB -> median -> sum (gi-gs_amd/relight.py)
        import relight
        hdri = torch.from_numpy(scenes.synthetic_envmap(512, 1024, seed=1)).to(dev)
        light = relight.make_light(hdri, res=256)
        relighter = relight.Relighter(light, gi, args.sh_degree, metallic=False, fused=(args.fused == "on"), brdf_lut=brdf_lut,
                                      graphs=(args.graphs == "on"))
        g_inf = {k: v.detach() for k, v in g.items()}
    else:
        stepper = pipeline.Stage2Step(light, brdf_lut, gi, args.sh_degree, graphs=(args.graphs == "on"),
                                      fused=(args.fused == "on"))
    checked = [0]

    def one_step(i):
        vi = dp.view_for(i, rank, world, n_views)
        cam = cams_t[vi]
        if inference:
            return relighter(cam, g_inf, view_dirs[vi])
        for p in flat_params:
            p.grad = None
        with (dgr_mod.grad_sink(sink) if sink is not None else _NULL):
            out = stepper(cam, g, gt_image, view_dirs[vi])
        if use_dist and dist.is_initialized():
            # one flat bucket on the communication stream: xGMI is point-to-point, a single large all-reduce keeps every
            # link busy.  The step ends where an optimizer would read the reduced gradients, so the wait is part of it
            # (exact, no one-step gradient delay).
            check = reduce_only is not None and checked[0] < 2
            checked[0] += 1
            slab.allreduce_async(force=force_coll, only=reduce_only, check_rest_zero=check)
            slab.wait()
        return out

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_block(first):
        """EXACTLY --steps steps between two (barrier + synchronize) brackets; the maximum over ranks."""
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            one_step(first + i)
        barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    if use_dist:
        one_step(0)  # captures the hipGraphs (untimed, before the process group exists)
        torch.cuda.synchronize()
        init_dist()
    for i in range(args.warmup):
        one_step(i)
    if slab is not None:
        torch.cuda.synchronize()
        slab.comm_stats()  # drop the warm-up collectives' events
    with gigs_lib.profile() as prof:
        elapsed = timed_block(args.warmup)  # the contract's measurement
    comm = None
    if slab is not None:
        comm = slab.comm_stats()
        if comm is not None:
            comm["reduced"] = ("stage-2 non-zero set: albedo, roughness, metallic, light (rest verified zero on the first 2 steps)"
                               if reduce_only is not None else "every gradient")
            comm["floats_per_gaussian"] = round((comm["bytes"] / 4 - sum(p.numel() for p in params_light)) / P, 2)
            comm["backend"] = backend
    blocks = [elapsed]
    for b in range(1, max(1, args.repeats)):
        blocks.append(timed_block(args.warmup + b * args.steps))
    per_step = sorted(1e3 * x / args.steps for x in blocks)
    repeats = {"blocks": len(blocks), "steps_per_block": args.steps, "ms_per_step_median": round(per_step[len(per_step) // 2], 4),
               "ms_per_step_min": round(per_step[0], 4), "ms_per_step_max": round(per_step[-1], 4),
               "value_median": round(args.steps * world / (1e-3 * per_step[len(per_step) // 2] * args.steps), 3)}

    # multi-rank: the COMPLETE training iterations under data parallelism, on every rank (their collectives need all of
    # them): Stage2Trainer.data_parallel() reduces the stage-2 gradient set, Stage1Trainer.data_parallel() every gradient
    # (what GIGS_BENCH_REDUCE=all measures for the metric's step) -- each with its own `comm`
    dp_extras = {}
    if use_dist and not inference and (not args.no_extras or args.iteration):
        import train_iteration
        for p_ in flat_params:
            p_.grad = None
        if stepper is not None:
            stepper.close()  # the headline step's graphs: their memory pool is not needed beside the trainers'
        kw = dict(steps=max(20, args.steps), warmup=5, data_parallel=True, force=force_coll, rank=rank, world=world)
        # a failure here (the same on every rank: SPMD) must not cost the headline line above
        try:
            dp_extras["iteration"] = train_iteration.bench_iteration(sc, light, brdf_lut, gi, args.sh_degree, cams_t, view_dirs,
                                                                     gt_image, **kw)
        except Exception as ex:  # noqa: BLE001
            dp_extras["iteration"] = {"error": repr(ex)[:400]}
        try:
            dp_extras["iteration_stage1"] = train_iteration.bench_stage1_iteration(sc, gi, args.sh_degree, cams_t, gt_image, **kw)
        except Exception as ex:  # noqa: BLE001
            dp_extras["iteration_stage1"] = {"error": repr(ex)[:400]}
    line = None
    if rank == 0:
        # V (visible), R (instances) averaged over the views of the timed region (outside it)
        import diff_gaussian_rasterization as dgr
        Vs, Rs = [], []
        e = torch.Tensor([])
        with torch.no_grad():
            for i in range(min(args.steps, 8)):
                cam = cams_t[((args.warmup + i) * world) % n_views]
                res = dgr._C.rasterize_gaussians(
                    torch.zeros(3, device=dev), g["means3D"], e, g["opacities"], g["normal"], g["albedo"], g["roughness"],
                    g["metallic"], g["scales"], g["rotations"], e, g["shs"], cam["campos"], cam["viewmatrix"],
                    cam["projmatrix"], 1.0, cam["tanfovx"], cam["tanfovy"], H, W, args.sh_degree, False, False, False, False)
                Rs.append(int(res[0]))
                Vs.append(int((res[2] > 0).sum()))
                del res
        V, R = float(np.mean(Vs)), float(np.mean(Rs))
        N, T = H * W, ((W + 15) // 16) * ((H + 15) // 16)
        ab = algorithmic_bytes(P, V, R, N, M, T)
        kernels = {}

        def add_stages(stages, steps, source):
            for name, (ms, n) in stages.items():
                if name in kernels or n == 0:
                    continue
                avg = ms / n
                rec = {"launches_per_step": n / steps, "avg_ms": round(avg, 4), "ms_per_step": round(ms / steps, 4),
                       "timed": source}
                if name in ab:
                    rec["alg_bytes_per_launch"] = int(ab[name])
                    rec["achieved_GBs"] = round(ab[name] / (avg * 1e-3) / 1e9, 2)
                kernels[name] = rec

        add_stages(prof.stages, args.steps, "live")  # hipEvents over the timed region
        eager = None
        if args.graphs == "on" and world == 1 and not inference:
            # stages replayed from a hipGraph carry no events: time them on a few extra eager steps of the same
            # workload OUTSIDE the timed region (marked "eager-extra"; the live entries above are not touched)
            eager = pipeline.Stage2Step(light, brdf_lut, gi, args.sh_degree, graphs=False, fused=(args.fused == "on"))
            extra = 4
            for i in range(2):
                eager(cams_t[i % n_views], g, gt_image, view_dirs[i % n_views])
            torch.cuda.synchronize()
            with gigs_lib.profile() as prof2:
                for i in range(extra):
                    eager(cams_t[(2 + i) % n_views], g, gt_image, view_dirs[(2 + i) % n_views])
                torch.cuda.synchronize()
            add_stages(prof2.stages, extra, "eager-extra")
        # light pre-filter: its real traffic is the cached pair-weight tables, not the textures (PMC FETCH_SIZE of the
        # committed pass when there is one)
        for nm in ("cubemap_fwd", "cubemap_bwd"):
            if nm in kernels:
                kernels[nm]["note"] = ("per launch of the %d launches per step (mips, diffuse, merged GGX levels); the GGX launch "
                                       "streams its cached pair weights: see roofline_light" % round(kernels[nm]["launches_per_step"]))
        # the roofline entry: the kernel with the largest time per step among those with an event timing -- taken live
        # over the timed region where the stage is launched eagerly, otherwise from the eager steps right after it (a
        # step replayed from hipGraphs, the default, has no events inside its kernel nodes)
        live = [k for k in kernels if "achieved_GBs" in kernels[k] and kernels[k]["timed"] == "live"]
        with_bytes = live or [k for k in kernels if "achieved_GBs" in kernels[k]]
        dom = max(with_bytes, key=lambda k: kernels[k]["ms_per_step"]) if with_bytes else None
        roofline = None
        if dom is not None and "achieved_GBs" in kernels[dom]:
            a = kernels[dom]["achieved_GBs"]
            roofline = {"kernel": dom, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(a / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(dom), "timed": kernels[dom]["timed"]}
            vb = pmc_valu_busy(dom)
            if vb is not None:
                # What actually binds the kernel when it is not HBM: the share of its cycles with the vector ALUs issuing.
                # FLAT fields (the driver's parser keeps only those).  The instruction stream's VALU-active cycles come
                # from the committed PMC pass named in `pmc_summary` (PMC needs rocprofv3: not collectable inside this
                # run), the duration they are divided by is THIS run's live average: `frac` = VALU-active SIMD cycles per
                # second against 1024 SIMDs x 2.4 GHz.  `bound` names the larger of the two fractions; the HBM figures the
                # contract asks for stay beside it as hbm_*.
                roofline["valu_busy"] = vb
                roofline["pmc_summary"] = _pmc_file()
                active = _pmc_entry(dom).get("SQ_ACTIVE_INST_VALU", {}).get("mean_per_launch")
                dur_s = kernels[dom]["avg_ms"] * 1e-3
                valu_frac = (4.0 * active / (1024.0 * 2.4e9 * dur_s)) if active else vb
                roofline["limiter_resource"] = "valu_issue" if valu_frac > roofline["frac"] else "hbm"
                if valu_frac > roofline["frac"]:
                    roofline.update({"hbm_achieved": a, "hbm_peak": HBM_PEAK_GBS, "hbm_unit": "GB/s", "hbm_frac": roofline["frac"],
                                     "bound": "valu", "achieved": round(valu_frac * 1024 * 2.4, 1), "peak": round(1024 * 2.4, 1),
                                     "unit": "G VALU-active SIMD cycles/s", "frac": round(valu_frac, 4)})
            if dom in ("ssao", "ssr"):
                roofline["note"] = ("dominant kernel by time; bound / achieved / peak / frac are the roof that binds it, fp32 VALU "
                                    "issue (VALU-active SIMD cycles of its instruction stream, from the committed PMC pass, over this "
                                    "run's live duration, against 1024 SIMDs x 2.4 GHz); its HBM roofline (algorithmic bytes / duration: "
                                    "hbm_*) does not: the z-plane lookups are served by L2 / L1 (the plane is 2.5 MB) and most are "
                                    "skipped by the certification table in LDS (DESIGN.md section 5)")
        cpu = parity_rep = None
        vi0 = args.warmup % n_views
        if not args.no_cpu_baseline and world == 1:
            if inference:
                cpu, parity_rep = cpu_baseline_and_parity_c3(sc, cams[vi0], gi, args.sh_degree, relighter, light, g_inf,
                                                             view_dirs[vi0], cams_t[vi0])
            else:
                cpu, parity_rep = cpu_baseline_and_parity(sc, cams[vi0], gi, args.sh_degree, light, brdf_lut, stepper,
                                                          args.config.upper(), single_thread_res=args.cpu_single_res)
        extras = {}
        if (not args.no_extras or args.iteration) and world == 1 and not inference:
            extras = secondary_steps(args, g, light, brdf_lut, gi, cams_t, view_dirs, gt_image, n_views, sc)
        what = ("relight renders/sec (inference, PBR+indirect, mips built once)" if inference
                else "train-step renders/sec (fwd+bwd, G-buffer+indirect)")
        dense = bool(getattr(stepper, "_dense", False))
        line = {
            "metric": "%s at %dx%d; PSNR vs the CPU oracle" % (what, W, H),
            "value": round(args.steps * world / elapsed, 3), "unit": "renders/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": data_tag,
            "config": {"workload": "%s (%s): %dk %s Gaussians, %dx%d, SH deg %d, GI step=%d start=%d delta=%g"
                                   % (args.config.upper(), cname, P // 1000, "trained (--scene)" if args.scene else "surface",
                                      W, H, args.sh_degree, gi["step"], gi["start"], gi["delta"]),
                       "P": P, "V": round(V), "R": round(R), "N": N, "M": M, "shade": shade, "hip_graphs": args.graphs, "fused_glue": args.fused,
                       "gi_march": gigs_lib.GI_MARCHES[gigs_lib.current().option("gi_march")],
                       "rasterizer": ("one hipGraph per view" if inference and args.graphs == "on" else
                                      "hipGraph (GIGS_RASTER_GRAPH=1)" if os.environ.get("GIGS_RASTER_GRAPH", "0") == "1"
                                      else "whole step = 2 hand-captured hipGraphs (fwd, bwd), asynchronous binning"
                                      if (stepper is not None and getattr(stepper, "whole", None) is not None)
                                      else "eager launches, synchronous binning (one read-back of the instance count%s)"
                                      % ("; dense scene: global radix sort" if dense else "")
                                      if (args.graphs == "off" or args.fused == "off" or dense
                                          or os.environ.get("GIGS_RASTER_ASYNC", "1") != "1")
                                      else "eager launches, asynchronous binning (no host read-back)"),
                       "parallelism": "view-parallel dp%d, 1 view/GPU/step, flat grad all-reduce%s"
                                      % (world, " of the stage-2 non-zero gradient set" if reduce_only else " of every gradient")},
            "ranks_seen": ranks_seen, "comm": comm, "repeats": repeats,
            "roofline": roofline, "roofline_light": None if inference else pmc_light_roofline(), "cpu_baseline": cpu,
            "kernels": kernels,
        }
        line.update(extras)
        line.update(dp_extras)
        if parity_rep is not None:
            # the timed GPU path vs the CPU oracle on one view of this very workload (oracle/parity.py)
            line["psnr_vs_oracle_db"] = parity_rep.get("psnr_render_rgb")
            for k in ("K_pairs_evaluated", "K_pairs_contributing", "covered_px_frac"):
                if k in parity_rep:
                    line["config"][k] = parity_rep[k]
            line["parity_" + args.config] = parity_rep
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return line


def _time_steps(fn, warm, n):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(warm + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return {"ms_per_step": round(1e3 * dt, 3), "renders_per_s": round(1.0 / dt, 2), "steps": n}


def secondary_steps(args, g, light, brdf_lut, gi, cams_t, view_dirs, gt_image, n_views, sc):
    """Secondary measurements of the same workload, after the timed region (rank 0, one GPU):

    drop_in_step    what an UNMODIFIED train.py gets through the reference's own API, op by op: GaussianRasterizer ->
                    gbuffer post-processing -> pbr_shading -> Gaussian_SSR -> L1 -> loss.backward(), eager, torch glue
                    (Stage2Step(graphs=False, fused=False)): the drop-in claim's own figure
    full_grad_step  rasterizer forward (+ in-op filters + SSAO) and backward with ALL SEVEN incoming gradient planes
                    live (colour, opacity, depth, normal, albedo, roughness, metallic: the blend backward's dL/dalpha
                    chain runs, which the stage-2 pattern of the headline step skips); no shade / SSR
    iteration       a COMPLETE stage-2 training iteration of train.py:247-523: activations of the raw parameter groups,
                    the headline step, BRDF TV + lamb + envmap TV, backward, Adam on Gaussians and light, clamp
    iteration_cached_geometry  the same iteration over eight revisited views with pipeline.GeometryCache: tile lists,
                    occlusion plane and the indirect-light hit list are reused per view while the optimizer reports no
                    change of a geometry bit (stage 2 updates materials and light only); same updates, fewer kernels.
                    A secondary figure: `value`, `drop_in_step` and `iteration` run everything
    iteration_stage1  a complete stage-1 iteration (train.py:266-331): L1 + D-SSIM + normal losses, full backward, Adam
    """
    out = {}
    leaves = list(g.values()) + list(light.parameters())

    def clear():
        for p in leaves:
            p.grad = None

    if not args.no_extras:
        _drop_in_and_full_grad(args, g, light, brdf_lut, gi, cams_t, view_dirs, gt_image, n_views, out, clear)
    clear()
    try:
        import train_iteration
        out["iteration"] = train_iteration.bench_iteration(sc, light, brdf_lut, gi, args.sh_degree, cams_t, view_dirs, gt_image,
                                                           steps=max(20, args.steps), warmup=5)
        out["iteration_cached_geometry"] = train_iteration.bench_iteration(sc, light, brdf_lut, gi, args.sh_degree, cams_t, view_dirs,
                                                                           gt_image, steps=max(20, args.steps), warmup=5,
                                                                           geometry_cache=True)
        out["iteration_stage1"] = train_iteration.bench_stage1_iteration(sc, gi, args.sh_degree, cams_t, gt_image,
                                                                         steps=max(20, args.steps), warmup=5)
        lean = train_iteration.bench_stage1_iteration(sc, gi, args.sh_degree, cams_t, gt_image, steps=max(20, args.steps),
                                                      warmup=5, compute_occlusion=False)
        out["iteration_stage1"]["without_the_unused_ssao_march"] = {k: lean[k] for k in ("iterations_per_s", "ms_per_iteration")}
    except ImportError:
        pass
    return out


def _drop_in_and_full_grad(args, g, light, brdf_lut, gi, cams_t, view_dirs, gt_image, n_views, out, clear):
    drop = pipeline.Stage2Step(light, brdf_lut, gi, args.sh_degree, graphs=False, fused=False)

    def drop_step(i):
        clear()
        drop(cams_t[i % n_views], g, gt_image, view_dirs[i % n_views])
    out["drop_in_step"] = dict(_time_steps(drop_step, 3, 12), what="op-by-op through GaussianRasterizer / pbr_shading / "
                               "Gaussian_SSR, eager, torch glue (--fused off --graphs off)")
    del drop
    torch.manual_seed(11)
    H, W = args.H, args.W
    wts = [torch.randn(c, H, W, device=gt_image.device) / (H * W) for c in (3, 1, 1, 3, 3, 1, 1)]
    bg = torch.zeros(3, device=gt_image.device)

    def full_step(i):
        clear()
        o, _m2d, _ = pipeline.rasterize(cams_t[i % n_views], g, args.sh_degree, bg, gi)
        planes = (o[0], o[2], o[3], o[5], o[7], o[8], o[9])
        torch.autograd.backward(planes, wts)
    with gigs_lib.profile() as prof:
        rec = _time_steps(full_step, 3, 12)
    st = {k: round(ms / n, 4) for k, (ms, n) in prof.stages.items() if n and k in ("blend_fwd", "blend_bwd", "preprocess_bwd")}
    out["full_grad_step"] = dict(rec, kernels_avg_ms=st, what="rasterizer + in-op filters + SSAO forward, backward with all seven "
                                 "incoming gradient planes live, eager")


if __name__ == "__main__":
    sys.exit(main())
