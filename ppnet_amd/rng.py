"""Random-stream policy of the drop-in classes.

"mt19937" (default): the host replays the reference's own global streams — `np.random.random` for path
    samples / placements / obstacles and `torch.rand` for pocket obstacles — in the reference's order and
    feeds the draws to the kernels, so `np.random.seed(s); torch.manual_seed(s)` scripts keep their meaning.
    The torch stream also advances by the one draw torchvision 0.12's `RandomRotation` takes per path
    (Path.py:160-161) and per placed map (MapGenerate.py:103-104).
    Pocket obstacles consume `torch.rand` isle by isle in the order of Qhull's vertex list (Path.py:388-395,463-537): Path
    computes Qhull's first vertex with scipy on the host and hands it to the kernel (`hull_start`, ppn_edage_paths_ex2), so
    the obstacles and the torch stream position are the reference's.
"philox": every draw is Philox4x32-10 keyed by (seed, stream, instance id, index) and generated on the
    device — the throughput mode, independent of launch order and of the number of GPUs.
"""
_state = {"mode": "mt19937", "seed": 0, "next_path": 0, "next_map": 0, "device": "cuda:0"}


def set_mode(mode, seed=0, device=None):
    if mode not in ("mt19937", "philox"):
        raise ValueError(mode)
    _state.update(mode=mode, seed=int(seed), next_path=0, next_map=0)
    if device is not None:
        _state["device"] = device


def mode():
    return _state["mode"]


def seed():
    return _state["seed"]


def device():
    return _state["device"]


def take_path_ids(n):
    first = _state["next_path"]
    _state["next_path"] += n
    return first


def take_map_ids(n):
    first = _state["next_map"]
    _state["next_map"] += n
    return first
