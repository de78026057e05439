"""Planner tail of the PPNet inference path on the GPU (reference: EDaGe-PP/process_map.py:293-425):
batched waypoint extraction from GenNet heat maps and batched circle-segment collision checks."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import rng


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _sp(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def resize_bilinear_u8(img, out_h, out_w):
    """PIL.Image.resize((out_w, out_h), BILINEAR) on u8 images [n,H,W] (device), bit-exact."""
    n, H, W = img.shape
    tmp = torch.empty(n, H, out_w, dtype=torch.uint8, device=img.device)
    out = torch.empty(n, out_h, out_w, dtype=torch.uint8, device=img.device)
    with torch.cuda.device(img.device):
        rc = L.lib.ppn_resize_bilinear_u8(_ptr(img.contiguous()), n, H, W, out_h, out_w, _ptr(tmp), _ptr(out), _sp(img.device))
    L.check(rc, "ppn_resize_bilinear_u8")
    return out


def extract_paths(heat_u8, init_state, end_state, down_sample_rate=2, max_wp=L.MAX_WAYPOINTS):
    """extract_path (process_map.py:293-365) for n heat maps [n,H,W] u8 on the device; init/end [n,2] f64 (full
    resolution).  Returns (ok [n] bool, waypoints [n,max_wp+2,2] f64 incl. start and goal, counts [n])."""
    dev = heat_u8.device
    n, H, W = heat_u8.shape
    h2, w2 = int(H / down_sample_rate), int(W / down_sample_rate)
    small = resize_bilinear_u8(heat_u8, h2, w2)
    heat = small.to(torch.float32) / 255.0                                 # T.ToTensor, process_map.py:302
    init = (init_state.to(torch.float64) / down_sample_rate).contiguous()
    end = (end_state.to(torch.float64) / down_sample_rate).contiguous()
    wp = torch.zeros(n, max_wp, 2, dtype=torch.float64, device=dev)
    wp_n = torch.empty(n, dtype=torch.int32, device=dev)
    ok = torch.empty(n, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib.ppn_extract_paths(_ptr(heat.contiguous()), n, h2, w2, _ptr(init), _ptr(end), max_wp, _ptr(wp), _ptr(wp_n),
                                     _ptr(ok), _sp(dev))
    L.check(rc, "ppn_extract_paths")
    # [init_state] + waypoints * rate + [end_state]  (process_map.py:355-359), one kernel
    full = torch.empty(n, max_wp + 2, 2, dtype=torch.float64, device=dev)
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    i64, e64 = init_state.to(torch.float64).contiguous(), end_state.to(torch.float64).contiguous()
    with torch.cuda.device(dev):
        rc = L.lib.ppn_assemble_paths(_ptr(wp), _ptr(wp_n), _ptr(ok), _ptr(i64), _ptr(e64), float(down_sample_rate), n, max_wp, _ptr(full),
                                      _ptr(counts), _sp(dev))
    L.check(rc, "ppn_assemble_paths")
    return ok.bool(), full, counts


def plan_collision(waypoints, counts, obstacles, n_obstacles, clearance, bound=224.0):
    """collision [B] bool: does any consecutive-waypoint segment of plan b (waypoints [B,M,2] f64, counts [B] i32 valid points) hit
    one of its first n_obstacles[b] obstacle rows obstacles[b] ([S,3]: ox, oy, size; float32 or float64)?  One launch for the
    whole batch (process_map.py:491-495)."""
    B, M = waypoints.shape[0], waypoints.shape[1]
    obs = obstacles if obstacles.dtype in (torch.float32, torch.float64) else obstacles.to(torch.float32)
    obs = obs.contiguous()
    S = obs.shape[1]
    out = torch.empty(B, dtype=torch.uint8, device=waypoints.device)
    with torch.cuda.device(waypoints.device):
        rc = L.lib.ppn_plan_collision(_ptr(waypoints.contiguous()), _ptr(counts.to(torch.int32).contiguous()), _ptr(obs),
                                      1 if obs.dtype == torch.float64 else 0, _ptr(n_obstacles.to(torch.int32).contiguous()), B, M, S,
                                      float(clearance), float(bound), _ptr(out), _sp(waypoints.device))
    L.check(rc, "ppn_plan_collision")
    return out.bool()


def collision_segments(s, e, prob, obs, obs_off, clearance, bound=224.0):
    """collision_check_circle_edge for n segments (device f32 [n,2] each); prob [n] i32 problem index;
    obs [m,3] f32 rows (ox, oy, size); obs_off [P+1] i32 CSR offsets. Returns hit [n] bool.
    bound: the map's resolution for the out-of-map test (224 = the reference's constant, process_map.py:384-387)."""
    n = s.shape[0]
    hit = torch.empty(n, dtype=torch.uint8, device=s.device)
    with torch.cuda.device(s.device):
        rc = L.lib.ppn_collision_segments_bound(_ptr(s.contiguous()), _ptr(e.contiguous()), _ptr(prob.contiguous()), n,
                                                _ptr(obs.contiguous()), _ptr(obs_off.contiguous()), float(clearance),
                                                float(bound), _ptr(hit), _sp(s.device))
    L.check(rc, "ppn_collision_segments_bound")
    return hit.bool()


def collision_check_single(s, e, obs, clearance):
    dev = torch.device(rng.device())
    st = torch.tensor([[float(s[0]), float(s[1])]], dtype=torch.float32, device=dev)
    et = torch.tensor([[float(e[0]), float(e[1])]], dtype=torch.float32, device=dev)
    ob = torch.tensor([[float(o[0]), float(o[1]), float(o[2])] for o in obs], dtype=torch.float32, device=dev).reshape(-1, 3)
    off = torch.tensor([0, ob.shape[0]], dtype=torch.int32, device=dev)
    prob = torch.zeros(1, dtype=torch.int32, device=dev)
    if ob.shape[0] == 0:
        ob = torch.zeros(1, 3, dtype=torch.float32, device=dev)
    return bool(collision_segments(st, et, prob, ob, off, clearance)[0])


def extract_path_pil(mask, init_state, end_state, down_sample_rate=8):
    dev = torch.device(rng.device())
    a = torch.tensor(np.asarray(mask, dtype=np.uint8)[None], device=dev)
    i0 = torch.tensor(np.asarray(init_state, dtype=np.float64).reshape(1, 2), device=dev)
    e0 = torch.tensor(np.asarray(end_state, dtype=np.float64).reshape(1, 2), device=dev)
    ok, full, cnt = extract_paths(a, i0, e0, down_sample_rate)
    if not bool(ok[0]):
        return False, None
    return True, full[0, :int(cnt[0])].cpu()
