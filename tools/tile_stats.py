"""Per-tile instance-list statistics of the bench view + a list-scheduling estimate (in-order vs longest-first)."""
import os, sys, heapq
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gi-gs_amd"))
import numpy as np, torch
import scenes, pipeline, gigs_lib
import diff_gaussian_rasterization as dgr
dev = "cuda:0"
sc = scenes.surface_scene(P=300_000, sh_degree=2, seed=0)
lib = gigs_lib.lib()
for vi in (0, 3):
    cam = scenes.orbit_camera(vi, 8, 800, 800)
    g = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if hasattr(v, "dtype")}
    e = torch.Tensor([])
    camt = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    st = pipeline.make_settings(camt, 2, torch.zeros(3, device=dev), scenes.GI_DEFAULTS, dev)
    res = dgr._C.rasterize_gaussians(st.bg, g["means3D"], e, g["opacities"], g["normal"], g["albedo"], g["roughness"],
                                     g["metallic"], g["scales"], g["rotations"], e, g["shs"], st.campos, st.viewmatrix,
                                     st.projmatrix, 1.0, st.tanfovx, st.tanfovy, 800, 800, 2, False, False, False, False)
    torch.cuda.synchronize()
    R, binning, img = res[0], res[4], res[5]
    T = 50 * 50
    off = lib.gigs_image_offset(800, 800, 2)
    ranges = img[off:off + 8 * T].cpu().numpy().view(np.uint32).reshape(T, 2)
    ln = (ranges[:, 1] - ranges[:, 0]).astype(np.int64)
    hm_off = None
    print("view", vi, "R", R, "len mean %.0f median %.0f p90 %.0f p99 %.0f max %d empty %d" % (
        ln.mean(), np.median(ln), np.percentile(ln, 90), np.percentile(ln, 99), ln.max(), (ln == 0).sum()))
    def sched(order, slots=1024, a=300.0, b=1.0):
        h = [0.0] * slots
        heapq.heapify(h)
        for t in order:
            s = heapq.heappop(h)
            heapq.heappush(h, s + a + b * ln[t])
        return max(h)
    ideal = (300.0 * T + ln.sum()) / 1024
    print("  makespan in-order %.0f  longest-first %.0f  ideal %.0f  (max single %.0f)" % (
        sched(range(T)), sched(np.argsort(-ln)), ideal, 300 + ln.max()))

# contribution density of the longest tiles (per quadrant), from the forward's hit bytes
for vi in (0,):
    off_b = lib.gigs_binning_offset(R, 3)
    # hit_mask follows the sort space in the binning chunk; locate it through its size: 4R bytes after point_list etc.
    # (diagnostic only: recompute by brute force instead) -> use n_contrib as a proxy for per-pixel depth of the walk
    ncon = img[lib.gigs_image_offset(800, 800, 1):lib.gigs_image_offset(800, 800, 1) + 4 * 640000].cpu().numpy().view(np.uint32).reshape(800, 800)
    order = np.argsort(-ln)[:5]
    for t in order:
        ty, tx = divmod(int(t), 50)
        blk = ncon[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]
        q = [blk[:8, :8].max(), blk[:8, 8:].max(), blk[8:, :8].max(), blk[8:, 8:].max()]
        print("  tile", t, "len", ln[t], "last contributor per quadrant (max over pixels)", q, "mean", int(blk.mean()))

# walk length per (tile, quadrant): a quadrant stops early only when all its 64 pixels are saturated
fT = img[lib.gigs_image_offset(800, 800, 0):lib.gigs_image_offset(800, 800, 0) + 4 * 640000].cpu().numpy().view(np.float32).reshape(800, 800)
walk = []
for t in range(T):
    ty, tx = divmod(t, 50)
    for qy in (0, 1):
        for qx in (0, 1):
            sl = (slice(ty * 16 + qy * 8, ty * 16 + qy * 8 + 8), slice(tx * 16 + qx * 8, tx * 16 + qx * 8 + 8))
            # a saturated pixel stopped at a Gaussian after its last contributor: T * (1 - alpha) < 1e-4
            sat = (fT[sl] < 0.02).all()
            walk.append(min(ln[t], int(ncon[sl].max()) + 64) if sat else ln[t])
walk = np.array(walk)
print("walked instances per quadrant: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d ; sum %d (vs 4R = %d)" % (
    walk.mean(), np.median(walk), np.percentile(walk, 90), np.percentile(walk, 99), walk.max(), walk.sum(), 4 * ln.sum()))

# backward work per (tile, quadrant): instances the forward recorded as blended there
hoff = lib.gigs_binning_offset(R, 4)
hm = binning[hoff:hoff + 4 * R].cpu().numpy().reshape(R, 4)
hits = []
for t in range(T):
    seg = hm[ranges[t, 0]:ranges[t, 1]]
    hits.extend(seg.sum(0).tolist() if len(seg) else [0, 0, 0, 0])
hits = np.array(hits)
print("blended instances per quadrant: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d ; sum %d" % (
    hits.mean(), np.median(hits), np.percentile(hits, 90), np.percentile(hits, 99), hits.max(), hits.sum()))
