"""The planner-side functions of EDaGe-PP/process_map.py that sit on the hot path, with the reference's
signatures: add_init_end_single (:119-145), extract_path (:293-365), collision_check_circle_edge (:383-425)."""
import numpy as np
import torch

from ppnet_amd import _lib as L
from ppnet_amd import edage, plan, rng


def add_init_end_single(image, init, end):
    """image: float tensor [3,R,R] (values in [0,1]) or u8 grid [R,R]; paints the 7x7 start/goal squares."""
    assert len(image.shape) in (2, 3), "Image shape incorrect"
    assert init is not None, "Init is None"
    assert end is not None, "End is None"
    dev = torch.device(rng.device())
    i0 = torch.tensor(np.asarray(init, dtype=np.float64).reshape(1, 2), device=dev)
    e0 = torch.tensor(np.asarray(end, dtype=np.float64).reshape(1, 2), device=dev)
    if image.dim() == 2:
        g = image.to(dev).contiguous().unsqueeze(0)
        return edage.paint_markers(g, i0, e0)[0]
    R = image.shape[1]
    mark = edage.paint_markers(torch.zeros(1, R, R, dtype=torch.uint8, device=dev), i0, e0)[0] == L.GRID_MARK
    red = torch.tensor([255.0, 0.0, 0.0], device=image.device, dtype=image.dtype)      # process_map.py:120
    image[:, mark.to(image.device)] = red[:, None]
    return image


def extract_path(mask, init_state, end_state, down_sample_rate=8):
    """mask: PIL 'L' image (the GenNet heat map). Returns (bool, tensor [n+2,2] | None)."""
    return plan.extract_path_pil(mask, init_state, end_state, down_sample_rate)


def collision_check_circle_edge(s, e, obs, clearance):
    return plan.collision_check_single(s, e, obs, clearance)
