"""PPNet inference path, batched over planning problems (reference: SegNet/test.py -> GenNet/predict.py ->
EDaGe-PP/process_map.extract_path_image): occupancy grid -> SegNet free-space mask -> GenNet waypoint heat map ->
greedy waypoint extraction -> circle-segment collision check.  Everything stays on the device."""
import os

import torch

from . import _lib as L
from . import edage, plan
from .gennet import AEViT, normalize_heatmap_u8
from .segnet import SegNet, normalize_images


_TUNED = os.environ.get("PPNET_TUNED_TABLE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gemm_gfx950_b256.csv")


def _use_tuned_gemms():
    """hipBLASLt solution picks (PyTorch TunableOp, recorded on MI355X, ROCm 7.2) for the vendor GEMMs of the prepared bfloat16
    batch — since round 4 the projections of the NAT levels at C >= 512 (segnet.LIBRARY_GEMM_FROM_C: addmm_ with beta = 1,
    _addmm_activation with the GELU epilogue) at batch 256 of 256 x 256 and of 512 x 512 maps —, for the A/B knobs that send more
    of the network to the library (PPNET_LIBRARY_GEMM, PPNET_NO_LN_FOLD's fallbacks) and for the float32 leg.  Read-only: no tuning
    at run time; shapes not in the table, or a table whose validators do not match the installed libraries, fall back to the
    default heuristics."""
    try:
        import torch.cuda.tunable as tn
        if os.path.exists(_TUNED) and torch.cuda.is_available():
            import tempfile
            tn.enable(True)
            retune = os.environ.get("PPNET_TUNE_GEMMS")          # maintainer knob: path of a table to (re)record
            tn.tuning_enable(bool(retune))
            # TunableOp saves its table to its filename at exit: point that at a scratch path, never at the shipped table
            tn.set_filename(retune or os.path.join(tempfile.gettempdir(), f"ppnet_amd_tunableop_{os.getpid()}.csv"))
            tn.read_file(_TUNED)
    except Exception as e:                                   # an optional speed-up must never break inference
        import sys
        print(f"ppnet_amd: tuned GEMM table not used ({e})", file=sys.stderr)


class CapturedPlan:
    """One recorded batch of PPNet.capture(): static inputs (grid, init, end, obstacles, n_obstacles, tail_heat), the HIP graph,
    and the buffers its kernels write (mask, heat, result = plan_tail's dict).  replay() enqueues the graph on the current
    stream and returns `result`; the buffers are overwritten by the next replay."""
    graph = None
    mask = heat = result = None

    def replay(self):
        self.graph.replay()
        return self.result


class PPNet(torch.nn.Module):
    def __init__(self, resolution=256, segnet=None, gennet=None, amp_dtype=None, weights_dtype=torch.bfloat16):
        """weights_dtype=bfloat16 (default): both networks hold bf16 weights and activations (fp32 accumulation inside
        the GEMM / conv / attention kernels, fp32 statistics inside LayerNorm); amp_dtype=bfloat16 with
        weights_dtype=None keeps fp32 weights under autocast; both None = fp32 everywhere."""
        super().__init__()
        # float32 weights keep the library convolutions on the path: exhaustive MIOpen search for their few shapes (a one-off at the
        # first batch; measured 64 -> 57 ms per 256-problem batch against the heuristic pick).  The bfloat16 default runs no library
        # convolution at all, and the flag is process-global (it would also send a later training step's backward convolutions through
        # the search), so it is not touched there.
        if weights_dtype is None:
            torch.backends.cudnn.benchmark = True
        _use_tuned_gemms()
        self.resolution = resolution
        self.segnet = segnet if segnet is not None else SegNet()
        self.gennet = gennet if gennet is not None else AEViT(1, 1, resolution, 24)      # predict.py:36,46
        self.amp_dtype = amp_dtype
        self.weights_dtype = weights_dtype
        self.segnet.prepare_inference()          # BN folded into convs (float32), channels_last weights
        self.gennet.prepare_inference()
        if weights_dtype is not None:
            self.segnet.to(weights_dtype)
            self.gennet.to(weights_dtype)

    @torch.no_grad()
    def segment(self, grid_u8):
        """u8 occupancy codes [B,R,R] -> free-space class mask [B,R,R] int64 (SegNet argmax)."""
        rgb = edage.grid_to_rgb(grid_u8) * 255.0                             # the JPEG the reference would read back
        x = normalize_images(rgb)
        if self.weights_dtype is not None:
            x = x.to(self.weights_dtype)
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            return self.segnet(x)

    @torch.no_grad()
    def segment_u8(self, grid_u8):
        """segment() as u8 labels: the normalised input straight from the occupancy codes (ppn_grid_to_image) and the
        fused output tail (SegNet.labels_u8)."""
        from .segnet import IMG_MEAN, IMG_STD
        from . import fused
        if self.amp_dtype is None and self.segnet.backbone.patch_embed.takes_codes(grid_u8):
            return self.segnet.labels_u8(grid_u8)              # the tokenizer reads the codes themselves (palette convolution)
        x = fused.grid_to_image(grid_u8, IMG_MEAN, IMG_STD, self.weights_dtype or torch.float32)
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            return self.segnet.labels_u8(x)

    @torch.no_grad()
    def heatmap(self, mask):
        """class mask [B,R,R] -> 8-bit waypoint heat map [B,R,R] (GenNet + per-sample min-max, predict.py:88-102)."""
        x = mask.to(self.weights_dtype or torch.float32).unsqueeze(1)        # my_dataset.py:15: values {0,1}
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            y = self.gennet(x)
        from . import fused
        return fused.heatmap_u8(y) if y.is_cuda else normalize_heatmap_u8(y)   # one kernel for predict.py:95-102's min-max

    @torch.no_grad()
    def plan(self, grid_u8, init, end, obstacles, n_obstacles, clearance=None, down_sample_rate=2):
        """Full pipeline for B problems. init/end [B,2] f64 (row, col); obstacles [B,S,3] f64 rows [col,row,r] with
        n_obstacles [B] valid. clearance in pixels: default 1/50 * resolution, the reference's call site
        (process_map.py:491-495: `1/50*224` on its 224-pixel maps) at this model's resolution; the out-of-map test uses
        the resolution as well.  Returns dict(ok, waypoints, counts, collision, success)."""
        heat = self.heatmap(self.segment(grid_u8) if os.environ.get("PPNET_NO_FUSED_TAIL") else self.segment_u8(grid_u8))
        return self.plan_tail(heat, init, end, obstacles, n_obstacles, clearance, down_sample_rate)

    @torch.no_grad()
    def generate_and_plan(self, paths, maps, placements, first_path_id=0, first_map_id=0, seed=0, obstacles_size=5, obstacles_num=20,
                          clearance=None, down_sample_rate=2, mark=None, gennet_input="segnet"):
        """BASELINE config 5's chain on the current stream, nothing read back in between: stage A into `paths` (a PathsBatch) ->
        stage B into `maps` (paths.n x placements maps) -> SegNet labels of those grids -> GenNet heat map of those labels ->
        extract_path + collision check on that heat map, start / goal / obstacles straight from `maps`
        (EDaGe-PP/MapGenerate.py:40-124 -> SegNet/test.py -> GenNet/predict.py -> process_map.py:452-506).
        mark: optional callable(stage name) called between the stages (bench.py records timing events with it).
        gennet_input: "segnet" — GenNet reads SegNet's labels, the reference's chain — or "labels": GenNet reads the generator's own
        mask_space of the same maps (ppn_label_masks, process_map.py:165-191: what SegNet is TRAINED to emit and GenNet was trained
        on).  SegNet still segments every grid — same kernels, same time — its mask is returned, and nothing downstream reads it: for
        a chain whose SegNet holds no trained weights (none ship with the reference, and a DiNAT-B checkpoint does not fit a repo).
        Returns dict(mask, heat, result)."""
        mark = mark or (lambda name: None)
        mark("start")
        edage.generate_paths(paths.n, paths.R, paths.map_size, paths.clearance, seed=seed, first_path_id=first_path_id,
                             device=paths.device, out=paths)
        edage.generate_maps(paths, placements, obstacles_size, obstacles_num, seed=seed, first_map_id=first_map_id, out=maps)
        mark("generated")
        mask = self.segment_u8(maps.grid)
        mark("segmented")
        if gennet_input == "labels":
            _, space = edage.label_masks(paths, maps, placements, want_path=False, want_space=True)
            heat = self.heatmap(space)
        else:
            heat = self.heatmap(mask)
        mark("heatmap")
        init, end = maps.segpoint[:, 0].contiguous(), maps.segpoint[:, 10].contiguous()
        result = self.plan_tail(heat, init, end, maps.obstacles, maps.n_obstacles[:, 0].contiguous(), clearance, down_sample_rate)
        mark("planned")
        return dict(mask=mask, heat=heat, result=result)

    @torch.no_grad()
    def capture(self, grid_u8, init, end, obstacles, n_obstacles, tail_heat=None, clearance=None, down_sample_rate=2, warmup=2):
        """The whole batch — segment_u8 -> heatmap -> plan_tail, ~290 kernel launches — recorded once as ONE HIP graph
        (hipStreamBeginCapture through torch.cuda.graph) for this batch shape; returns a CapturedPlan whose replay() is a single
        hipGraphLaunch.  The arguments become the graph's static input buffers (copy a new batch into `.grid`, `.init`, ... before
        replay()); the results live in `.mask`, `.heat`, `.result`.  tail_heat: 8-bit heat maps the planner tail walks instead of
        the network's own output (bench.py: ridge maps, see there).  Nothing in the path synchronises or allocates outside the
        caching allocator, so the recorded launches are exactly the eager ones (tests/test_ppnet_config3.py compares the two)."""
        cp = CapturedPlan()
        cp.grid, cp.init, cp.end, cp.obstacles, cp.n_obstacles, cp.tail_heat = grid_u8, init, end, obstacles, n_obstacles, tail_heat

        def body():
            cp.mask = self.segment_u8(cp.grid)
            cp.heat = self.heatmap(cp.mask)
            cp.result = self.plan_tail(cp.heat if cp.tail_heat is None else cp.tail_heat, cp.init, cp.end, cp.obstacles,
                                       cp.n_obstacles, clearance, down_sample_rate)
        side = torch.cuda.Stream(grid_u8.device)
        side.wait_stream(torch.cuda.current_stream(grid_u8.device))
        with torch.cuda.stream(side):                     # persistent buffers, library workspaces and solution picks: outside the capture
            for _ in range(max(1, warmup)):
                body()
        torch.cuda.current_stream(grid_u8.device).wait_stream(side)
        torch.cuda.synchronize(grid_u8.device)
        cp.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cp.graph):
            body()
        return cp

    @torch.no_grad()
    def plan_tail(self, heat, init, end, obstacles, n_obstacles, clearance=None, down_sample_rate=2, max_wp=L.MAX_WAYPOINTS, side_stream=None):
        """extract_path + collision_check_circle_edge over the consecutive waypoints (process_map.py:486-503) for B
        8-bit heat maps [B,R,R].  max_wp: the walk's step cap (stands in for the reference's 1 s timeout).
        side_stream: run the tail's kernels on that HIP stream behind everything the current stream has enqueued so far (an event),
        and return at once — the greedy walk is one wave per problem and as long as its longest walk (2 048 steps for a problem whose
        walk never arrives: 2.5 ms per batch with trained weights), a latency chain that fills 1 / 16 of the chip; on its own stream the
        NEXT batch's networks run beside it instead of behind it.  The result tensors then belong to `side_stream`: synchronise with it
        (or wait on an event recorded there) before reading them from another stream."""
        if side_stream is not None:
            cur = torch.cuda.current_stream(heat.device)
            ev = torch.cuda.Event()
            ev.record(cur)
            with torch.cuda.stream(side_stream):
                side_stream.wait_event(ev)
                for t in (heat, init, end, obstacles, n_obstacles):
                    t.record_stream(side_stream)
                return self.plan_tail(heat, init, end, obstacles, n_obstacles, clearance, down_sample_rate, max_wp)
        if clearance is None:
            clearance = 1 / 50 * self.resolution
        ok, wp, cnt = plan.extract_paths(heat, init, end, down_sample_rate, max_wp)
        # consecutive-waypoint segments of every problem against its own obstacles: one launch (process_map.py:491-495)
        collision = plan.plan_collision(wp, cnt, obstacles, n_obstacles, clearance, bound=self.resolution)
        return dict(ok=ok, waypoints=wp, counts=cnt, collision=collision, success=ok & ~collision)
