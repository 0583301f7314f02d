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
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
importlib.import_module("gi-gs_amd")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import dp  # noqa: E402
import gigs_lib  # noqa: E402
import pipeline  # noqa: E402
import scenes  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
PARAM_KEYS = ["means3D", "opacities", "normal", "albedo", "roughness", "metallic", "shs", "scales", "rotations"]


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


def _pmc_entry(stage: str):
    """Counters of `stage`'s kernel from the newest committed summary (profiles/rNN/pmc_summary*.json, by name order);
    template instantiations of one kernel are matched by prefix and the one with the most launches is taken."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_summary*.json")))
    if not files or stage not in _PMC_NAMES:
        return {}
    try:
        d = json.load(open(files[-1]))
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


def make_light(device, shade: str):
    if shade == "hip":
        import pbr
        light = pbr.CubemapLight(base_res=256).to(device)
        return light, pbr.get_brdf_lut().to(device)
    return None, None


def stub_step(cam, g, sh_degree, gi, gt_image):
    """Step without the deferred shade (used only until the HIP shade lands; flagged in the
    JSON line as config.shade = "none")."""
    dev = g["means3D"].device
    bg = torch.zeros(3, device=dev)
    res = pipeline.render(cam, g, sh_degree, bg, gi, derive_normal=True)
    H, W = cam["image_height"], cam["image_width"]
    albedo_map, metallic_map = res["albedo_map"], res["metallic_map"]
    roughness_map = res["roughness_map"] * (1.0 - 0.04) + 0.04
    occ = res["occlusion_map"].detach()
    render_direct = torch.where(res["normal_mask"], (albedo_map * occ).clamp(0, 1), bg[:, None, None])
    ssr = pipeline.Gaussian_SSR(cam["tanfovx"], cam["tanfovy"], W, H, gi["radius"], gi["bias"], gi["thick"],
                                gi["delta"], gi["step"], gi["start"])
    F0 = (1.0 - metallic_map) * 0.04 + albedo_map * metallic_map
    (IRR, _) = ssr(res["out_normal_view"].detach(), res["depth_pos"].detach(),
                   pipeline.srgb_to_linear(render_direct).detach(), albedo_map, roughness_map, metallic_map, F0)
    IRR = pipeline.filters.median_blur(pipeline.linear_to_srgb(IRR)[None], (3, 3))[0]
    loss = torch.abs(render_direct + IRR - gt_image).mean()
    loss = loss + 0.001 * ((1.0 - roughness_map[res["normal_mask"]]).mean() + metallic_map[res["normal_mask"]].mean())
    loss.backward()
    return dict(loss=loss.detach(), radii=res["radii"])


def cpu_baseline_and_parity(sc, cam, gi, sh_degree, light, brdf_lut, stepper, config):
    """The checker leg (rank 0, N = 1 only): ONE view of the timed workload on the CPU oracle -- rasterizer fwd, the
    operator's filters + SSAO, light pre-filter, shade, SSR, sRGB/median, rasterizer bwd (oracle/stage2_ref.py; no
    shade / light backward on the CPU, so the CPU time is a lower bound of a full step) -- timed on all host cores,
    and the same view from the timed GPU path compared with it (oracle/parity.py): integer state bit for bit,
    every fp plane's mean L1, rasterizer-backward gradients, and the PSNR (utils/image_utils.py:31) of the final
    stage-2 image."""
    from oracle import oracle as orc
    from oracle import parity
    orc.build()
    cores = orc.max_threads()
    orc.set_threads(cores)
    only = ("albedo", "roughness", "metallic")  # the planes stage 2 differentiates
    gpu = parity.gpu_capture(sc, cam, gi, sh_degree, light=light, brdf_lut=brdf_lut, stepper=stepper, grads_only=only,
                             dev=str(light.base.device) if light is not None else "cuda:0")
    ref, t = parity.oracle_capture(orc, sc, cam, gi, sh_degree, light_base=gpu.get("light_base"), grads_only=only)
    rep = parity.compare(gpu, ref)
    cpu = dict(value=round(1.0 / t["total"], 4), unit="renders/s", cores=cores, kind="port",
               sample="1 step of the same workload (%s, 1 view) on the oracle: " % config
                      + ", ".join("%s %.2fs" % (k, v) for k, v in t.items() if k != "total")
                      + "; backward blend is single-threaded (deterministic double sums), no shade/light backward on the CPU")
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="BASELINE.json configuration (c2 = the headline metric's; c3 relight inference; c4/c5 3 M Gaussians "
                         "at the Mip-NeRF360 images_4 resolutions)")
    ap.add_argument("--gaussians", type=int, default=None)
    ap.add_argument("--res", type=int, default=None, help="square image (overrides the config's size)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--sh-degree", type=int, default=None)
    ap.add_argument("--start", type=int, default=8, help="GI march start (8 = reference CLI default, 64 = README)")
    ap.add_argument("--shade", choices=["auto", "hip", "none"], default="auto")
    ap.add_argument("--graphs", choices=["on", "off"], default="on",
                    help="capture the launch-bound glue segments of the step into hipGraphs")
    ap.add_argument("--fused", choices=["on", "off"], default="on",
                    help="run the tensor glue between rasterizer and loss.backward() as the fused stage-2 node "
                         "(gi-gs_amd/stage2_fused.py) instead of op-by-op torch")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle leg (cpu_baseline and parity)")
    args = ap.parse_args()
    cP, cW, cH, cD, cname = CONFIGS[args.config]
    args.gaussians = cP if args.gaussians is None else args.gaussians
    args.sh_degree = cD if args.sh_degree is None else args.sh_degree
    W = args.width or args.res or cW
    H = args.height or args.res or cH

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path in the product)")
    # GIGS_BENCH_BACKEND=gloo with fewer GPUs than ranks is a rehearsal mode (ranks share cards, the all-reduce
    # goes through the host); the measured configuration is one rank per GPU over RCCL
    backend = os.environ.get("GIGS_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # GIGS_BENCH_FORCE_DIST=1: run the distributed code path (process group, gradient all-reduce, barriers) with a
    # single rank -- a one-GPU rehearsal of everything but the inter-GPU traffic
    use_dist = world > 1 or os.environ.get("GIGS_BENCH_FORCE_DIST", "0") == "1"
    dist = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

    def init_dist():
        # Called AFTER the first (hipGraph-capturing) step: RCCL's proxy thread issues HIP calls of its own, and a
        # HIP call from another thread while a stream is capturing in the default (global) mode fails the capture.
        if not use_dist:
            return
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, (world, args.gpus)

    shade = args.shade
    if shade == "auto":
        shade = "hip" if os.path.exists(os.path.join(ROOT, "gi-gs_amd", "pbr", "__init__.py")) else "none"

    gi = dict(scenes.GI_DEFAULTS, start=args.start)
    sc = scenes.surface_scene(P=args.gaussians, sh_degree=args.sh_degree, seed=0)
    P, M = sc["means3D"].shape[0], sc["shs"].shape[1]
    g = {k: torch.from_numpy(sc[k]).to(dev).requires_grad_(True) for k in PARAM_KEYS}
    n_views = 64
    cams = [scenes.orbit_camera(i, n_views, W, H, radius=3.5) for i in range(n_views)]
    cams_t = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams]
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, W, device=dev), indexing="ij")
    gt_image = torch.stack([0.5 + 0.3 * torch.sin(6 * xx), 0.5 + 0.3 * torch.cos(5 * yy), 0.4 + 0.2 * xx * yy])
    light, brdf_lut = make_light(dev, shade)
    rays = pipeline.canonical_rays(cams[0], dev)
    view_dirs = [pipeline.view_dirs_for(c, rays, dev) for c in cams_t]  # per-camera constants, like gt_image
    if light is not None:
        params_light = [p for p in light.parameters()]
    else:
        params_light = []
    # slab order: the stage-2 trainable set (materials, then the light) last and adjacent, so that
    # GIGS_BENCH_REDUCE=trainable can reduce it with ONE collective
    SLAB_KEYS = ["means3D", "opacities", "normal", "shs", "scales", "rotations", "albedo", "roughness", "metallic"]
    SINK_NAME = {"opacities": "opacity", "shs": "sh"}
    flat_params = [g[k] for k in SLAB_KEYS] + params_light
    reduce_only = None
    if os.environ.get("GIGS_BENCH_REDUCE", "all") == "trainable" and args.config != "c3":
        # opt-in (NOT the measured default): a stage-2 iteration's gradients w.r.t. everything but albedo / roughness /
        # metallic and the light are identically zero on every rank (dp.GradSlab.allreduce_async(only=...))
        reduce_only = list(range(SLAB_KEYS.index("albedo"), len(flat_params)))
    # multi-rank: every gradient lives in one persistent flat slab; the rasterizer's backward writes into it directly
    # (dgr.grad_sink), so the all-reduce bucket needs no packing pass
    slab = sink = None
    if use_dist:
        import diff_gaussian_rasterization as dgr_mod
        slab = dp.GradSlab(flat_params)
        sink = slab.sink([SINK_NAME.get(k, k) for k in SLAB_KEYS])
    force_coll = os.environ.get("GIGS_BENCH_FORCE_DIST", "0") == "1"
    stepper = relighter = None
    inference = args.config == "c3"
    if inference:
        # configs[2]: relight.py -- HDR latlong map -> 256^2 cubemap, build_mips ONCE, then per view
        # render(inference=True) -> pbr_shading -> Gaussian_SSR -> sRGB -> median -> sum (gi-gs_amd/relight.py)
        import relight
        if shade != "hip":
            raise SystemExit("--config c3 needs the HIP shade")
        hdri = torch.from_numpy(scenes.synthetic_envmap(512, 1024, seed=1)).to(dev)
        light = relight.make_light(hdri, res=256)
        relighter = relight.Relighter(light, gi, args.sh_degree, metallic=False, fused=(args.fused == "on"), brdf_lut=brdf_lut,
                                      graphs=(args.graphs == "on"))
        g_inf = {k: v.detach() for k, v in g.items()}
    elif shade == "hip":
        stepper = pipeline.Stage2Step(light, brdf_lut, gi, args.sh_degree, graphs=(args.graphs == "on"),
                                      fused=(args.fused == "on"))

    def one_step(i):
        vi = dp.view_for(i, rank, world, n_views)
        cam = cams_t[vi]
        if inference:
            return relighter(cam, g_inf, view_dirs[vi])
        for p in flat_params:
            p.grad = None
        with (dgr_mod.grad_sink(sink) if sink is not None else _NULL):
            if shade == "hip":
                out = stepper(cam, g, gt_image, view_dirs[vi])
            else:
                out = stub_step(cam, g, args.sh_degree, gi, gt_image)
        if use_dist and dist.is_initialized():
            # one flat bucket on the communication stream: xGMI is point-to-point, a single large all-reduce keeps every
            # link busy.  The step ends where an optimizer would read the reduced gradients, so the wait is part of it
            # (exact, no one-step gradient delay).
            slab.allreduce_async(force=force_coll, only=reduce_only)
            slab.wait()
        return out

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist:
        one_step(0)  # captures the hipGraphs (untimed, before the process group exists)
        torch.cuda.synchronize()
        init_dist()
    for i in range(args.warmup):
        one_step(i)
    barrier()
    with gigs_lib.profile() as prof:
        t0 = time.perf_counter()
        for i in range(args.steps):
            one_step(args.warmup + i)
        barrier()
        t1 = time.perf_counter()
    elapsed = t1 - t0
    if use_dist:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

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
        if shade == "hip" and args.graphs == "on" and world == 1 and not inference:
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
                        "frac": round(a / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(dom),
                        "valu_busy": pmc_valu_busy(dom), "timed": kernels[dom]["timed"],
                        "note": "dominant kernel by time; its z-plane lookups are served by L2/L1 (the plane is 2.5 MB) and "
                                "most of them are skipped by the certification table in LDS, so it is bound by fp32 VALU "
                                "issue, not by HBM (valu_busy = share of its cycles with the vector ALUs issuing, from the "
                                "committed PMC pass; DESIGN.md section 5)"}
        cpu = parity_rep = None
        vi0 = args.warmup % n_views
        if not args.no_cpu_baseline and world == 1 and shade == "hip":
            if inference:
                cpu, parity_rep = cpu_baseline_and_parity_c3(sc, cams[vi0], gi, args.sh_degree, relighter, light, g_inf,
                                                             view_dirs[vi0], cams_t[vi0])
            else:
                cpu, parity_rep = cpu_baseline_and_parity(sc, cams[vi0], gi, args.sh_degree, light, brdf_lut, stepper,
                                                          args.config.upper())
        what = ("relight renders/sec (inference, PBR+indirect, mips built once)" if inference
                else "train-step renders/sec (fwd+bwd, G-buffer+indirect)")
        line = {
            "metric": "%s at %dx%d; PSNR vs the CPU oracle" % (what, W, H),
            "value": round(args.steps * world / elapsed, 3), "unit": "renders/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s (%s): %dk surface Gaussians, %dx%d, SH deg %d, GI step=%d start=%d delta=%g"
                                   % (args.config.upper(), cname, P // 1000, W, H, args.sh_degree, gi["step"], gi["start"],
                                      gi["delta"]),
                       "P": P, "V": round(V), "R": round(R), "N": N, "M": M, "shade": shade, "hip_graphs": args.graphs, "fused_glue": args.fused,
                       "gi_march": os.environ.get("GIGS_GI_MARCH", "proj (default)"),
                       "rasterizer": ("one hipGraph per view" if inference and args.graphs == "on" else
                                      "hipGraph (GIGS_RASTER_GRAPH=1)" if os.environ.get("GIGS_RASTER_GRAPH", "0") == "1"
                                      else "whole step = 2 hand-captured hipGraphs (fwd, bwd), asynchronous binning"
                                      if (stepper is not None and getattr(stepper, "whole", None) is not None)
                                      else "eager launches, synchronous binning (one read-back of the instance count%s)"
                                      % ("; dense scene: global radix sort" if getattr(stepper, "_dense", False) else "")
                                      if (args.graphs == "off" or args.fused == "off" or getattr(stepper, "_dense", False)
                                          or os.environ.get("GIGS_RASTER_ASYNC", "1") != "1")
                                      else "eager launches, asynchronous binning (no host read-back)"),
                       "parallelism": "view-parallel dp%d, 1 view/GPU/step, flat grad all-reduce%s"
                                      % (world, " of the stage-2 trainable set only (GIGS_BENCH_REDUCE=trainable)" if reduce_only else "")},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
        }
        if parity_rep is not None:
            # the timed GPU path vs the CPU oracle on one view of this very workload (oracle/parity.py)
            line["psnr_vs_oracle_db"] = parity_rep.get("psnr_render_rgb")
            line["parity_" + args.config] = parity_rep
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
