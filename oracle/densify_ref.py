"""CPU restatement of the reference's densification.  TEST INFRASTRUCTURE ONLY.

Follows scene/gaussian_model.py step by step, on plain tensors instead of a GaussianModel:
  add_densification_stats :933-945 (+ train.py:494-498), densify_and_clone :750-786, densify_and_split :708-748,
  densification_postfix :660-706, prune_points / _prune_optimizer :595-633, densify_and_prune :907-931,
  reset_opacity :467-472, build_rotation utils/general_utils.py:89-110.
PARITY UNPINNED beyond this restatement: scene/gaussian_model.py cannot be imported here (plyfile, simple_knn,
pytorch3d are absent) and the reference holds no fixture for it.  The random draws of torch.normal are passed in as
standard-normal arrays (`z_clone`, `z_split`; a draw with std s is s*z) so that a test can feed both sides the same.
"""
import torch

NAMES = ["xyz", "f_dc", "f_rest", "opacity", "normal", "albedo", "roughness", "metallic", "scaling", "rotation"]


def build_rotation(r):
    q = r / torch.sqrt((r * r).sum(dim=1))[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.zeros((q.shape[0], 3, 3), dtype=r.dtype)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - w * z)
    R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y)
    R[:, 2, 1] = 2 * (y * z + w * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def add_stats(stats, grad2d, radii):
    """stats: dict accum, accum_abs, accum_abs_max, denom [P,1], max_radii2D [P]; in place."""
    vis = radii > 0
    stats["max_radii2D"][vis] = torch.max(stats["max_radii2D"][vis], radii[vis].to(stats["max_radii2D"].dtype))
    stats["accum"][vis] += torch.norm(grad2d[vis, :2], dim=-1, keepdim=True)
    a = torch.norm(grad2d[vis, :1].abs() + grad2d[vis, 1:2].abs(), dim=-1, keepdim=True)
    stats["accum_abs"][vis] += a
    stats["accum_abs_max"][vis] = torch.max(stats["accum_abs_max"][vis], a)
    stats["denom"][vis] += 1


def _cat(model, ext):
    """cat_tensors_to_optimizer + densification_postfix: parameters grow, moments grow by zeros, stats reset."""
    for n in NAMES:
        model["exp_avg"][n] = torch.cat((model["exp_avg"][n], torch.zeros_like(ext[n])), dim=0)
        model["exp_avg_sq"][n] = torch.cat((model["exp_avg_sq"][n], torch.zeros_like(ext[n])), dim=0)
        model["params"][n] = torch.cat((model["params"][n], ext[n]), dim=0)
    P = model["params"]["xyz"].shape[0]
    model["stats"] = dict(accum=torch.zeros(P, 1), accum_abs=torch.zeros(P, 1), accum_abs_max=torch.zeros(P, 1),
                          denom=torch.zeros(P, 1), max_radii2D=torch.zeros(P))


def _prune(model, mask):
    keep = ~mask
    for n in NAMES:
        model["params"][n] = model["params"][n][keep]
        model["exp_avg"][n] = model["exp_avg"][n][keep]
        model["exp_avg_sq"][n] = model["exp_avg_sq"][n][keep]
    for k in model["stats"]:
        model["stats"][k] = model["stats"][k][keep]


def densify_and_prune(model, max_grad, min_opacity, extent, max_screen_size, z_clone, z_split, percent_dense=0.01, N=2):
    """model: dict(params, exp_avg, exp_avg_sq: name -> tensor; stats).  z_clone [n_clone,3], z_split [N*n_split,3]
    are consumed in order (surplus rows ignored).  In place."""
    st = model["stats"]
    grads = st["accum"] / st["denom"]
    grads[grads.isnan()] = 0.0
    grads_abs = st["accum_abs"] / st["denom"]
    grads_abs[grads_abs.isnan()] = 0.0
    ratio = (torch.norm(grads, dim=-1) >= max_grad).float().mean()
    Q = torch.quantile(grads_abs.reshape(-1), 1 - ratio)
    p = model["params"]
    # clone (:750-786): small Gaussians with a large gradient; the copy gets a fresh position sample
    sel = (torch.norm(grads, dim=-1) >= max_grad) | (torch.norm(grads_abs, dim=-1) >= Q)
    sel = sel & (torch.exp(p["scaling"]).max(dim=1).values <= percent_dense * extent)
    stds = torch.exp(p["scaling"][sel])
    samples = stds * z_clone[: stds.shape[0]]
    ext = {n: p[n][sel] for n in NAMES}
    ext["xyz"] = torch.bmm(build_rotation(p["rotation"][sel]), samples.unsqueeze(-1)).squeeze(-1) + p["xyz"][sel]
    _cat(model, ext)
    p = model["params"]
    # split (:708-748): gradients padded with zeros for the clones just appended
    n_now = p["xyz"].shape[0]
    pg = torch.zeros(n_now)
    pg[: grads.shape[0]] = grads.squeeze()
    pga = torch.zeros(n_now)
    pga[: grads_abs.shape[0]] = grads_abs.squeeze()
    sel = (pg >= max_grad) | (pga >= Q)
    sel = sel & (torch.exp(p["scaling"]).max(dim=1).values > percent_dense * extent)
    stds = torch.exp(p["scaling"][sel]).repeat(N, 1)
    samples = stds * z_split[: stds.shape[0]]
    rots = build_rotation(p["rotation"][sel]).repeat(N, 1, 1)
    ext = {n: p[n][sel].repeat(N, *([1] * (p[n].dim() - 1))) for n in NAMES}
    ext["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + p["xyz"][sel].repeat(N, 1)
    ext["scaling"] = torch.log(torch.exp(p["scaling"][sel]).repeat(N, 1) / (0.8 * N))
    _cat(model, ext)
    _prune(model, torch.cat((sel, torch.zeros(N * int(sel.sum()), dtype=torch.bool))))
    p = model["params"]
    # prune (:919-927); max_radii2D was reset by the postfix above, so the screen-size test never fires here
    prune_mask = (torch.sigmoid(p["opacity"]) < min_opacity).squeeze(-1)
    if max_screen_size:
        big_vs = model["stats"]["max_radii2D"] > max_screen_size
        big_ws = torch.exp(p["scaling"]).max(dim=1).values > 0.1 * extent
        prune_mask = prune_mask | big_vs | big_ws
    _prune(model, prune_mask)


def reset_opacity(model):
    """:467-472 + replace_tensor_to_optimizer :580-593: opacity <- logit(min(sigmoid(opacity), 0.01)), moments zeroed."""
    o = torch.sigmoid(model["params"]["opacity"])
    o = torch.min(o, torch.ones_like(o) * 0.01)
    model["params"]["opacity"] = torch.log(o / (1 - o))
    model["exp_avg"]["opacity"] = torch.zeros_like(o)
    model["exp_avg_sq"]["opacity"] = torch.zeros_like(o)
