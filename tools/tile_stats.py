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
