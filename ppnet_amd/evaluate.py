"""Success-rate / path-length evaluation of batched plans (BASELINE config 5's PPNet column).

Reference: EDaGe-PP/process_map.py:452-506 (extract_path_image: a problem is solved when extract_path succeeds and no
consecutive-waypoint segment collides; the solution's cost is the polyline length) and the OMPL harness's stopping rule
experiments/ompl_experiments/updated_geometric_planner.py:260-277,349-354 (a planner is done once its best cost is within
(1 + epsilon) of the target path's length).  Everything here runs on the device batch `PPNet.plan` returned; nothing is
read back per problem.
"""
import torch

from . import edage


def plan_lengths(waypoints, counts):
    """Polyline length of each plan: waypoints [B,M,2] (row, col), counts [B] valid points (0 = no plan) -> [B] f64."""
    M = waypoints.shape[1]
    seg = (waypoints[:, 1:] - waypoints[:, :-1]).pow(2).sum(dim=2).sqrt()
    valid = torch.arange(M - 1, device=waypoints.device)[None, :] < (counts[:, None].to(torch.int64) - 1)
    return (seg * valid).sum(dim=1)


def evaluate_plans(result, target_length_px, epsilon=0.1):
    """result: the dict of PPNet.plan / plan_tail; target_length_px [B]: the target path's Length in pixels
    (Path.Length * R / map_size).  Returns a dict of Python floats:
      extract_ok     fraction with a waypoint chain reaching the goal            (process_map.py:486-490)
      collision_free fraction of those whose segments all pass the circle test   (:491-495)
      success        fraction solved = ok and no collision                       (:496-503)
      length_ratio   mean (plan length / target length) over solved problems
      within_eps     fraction of ALL problems solved with length <= (1+epsilon) * target  (the harness's criterion)"""
    ok, coll = result["ok"], result["collision"]
    succ = ok & ~coll
    length = plan_lengths(result["waypoints"], result["counts"])
    ratio = length / target_length_px.to(length.dtype)
    n_ok = int(ok.sum())
    n_s = int(succ.sum())
    B = ok.numel()
    return {"extract_ok": n_ok / B, "collision_free": (int((ok & ~coll).sum()) / n_ok) if n_ok else 0.0,
            "success": n_s / B, "length_ratio": float(ratio[succ].mean()) if n_s else None,
            "within_eps": int((succ & (ratio <= 1.0 + epsilon)).sum()) / B, "epsilon": epsilon, "problems": B}


def label_heatmaps(paths, maps, placements, sigma=2.0, bound=None):
    """8-bit heat maps [n,R,R] with a ridge along each map's label path: GenNet's training target (mask_path,
    process_map.py:148-163, every 5th label point) blurred with a Gaussian and min-max normalised per sample as
    predict.py:95-102 does — what a trained GenNet is fitted to produce (GenNet/train.py: MSE against mask_path).
    No trained weights ship with the reference; these maps let the planner tail be exercised on plans that exist."""
    mask_path, _ = edage.label_masks(paths, maps, placements, bound=bound, want_path=True, want_space=False)
    x = (mask_path > 0).to(torch.float32).unsqueeze(1)
    r = max(1, int(3 * sigma + 0.5))
    t = torch.arange(-r, r + 1, device=x.device, dtype=torch.float32)
    k = torch.exp(-t * t / (2 * sigma * sigma))
    k = k / k.sum()
    # separable blur as sums of shifted slices (no library convolution: PPNet switches MIOpen's exhaustive search on, and a
    # search over these one-off 1 x (2r+1) shapes can take minutes)
    H, W = x.shape[-2:]
    xp = torch.nn.functional.pad(x, (r, r, 0, 0))
    x = sum(k[i] * xp[..., :, i:i + W] for i in range(2 * r + 1))
    xp = torch.nn.functional.pad(x, (0, 0, r, r))
    x = sum(k[i] * xp[..., i:i + H, :] for i in range(2 * r + 1))
    from .gennet import normalize_heatmap_u8
    return normalize_heatmap_u8(x)
