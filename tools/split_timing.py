"""Times the two stage-B kernels apart, back to back and pipelined over streams (config 2).  Diagnostic only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd import edage

dev = torch.device("cuda", 0)
PATHS, PL, R, K = 100, 100, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 20
pbs = [edage.generate_paths(PATHS, R, 50, 3, seed=0, first_path_id=i * PATHS, device=dev) for i in range(4)]
mbs = [edage.MapsBatch(PATHS * PL, R, K, dev) for _ in range(2)]
torch.cuda.synchronize()


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print("place  alone %.4f ms" % timed(lambda: edage.generate_maps(pbs[0], PL, 5, K, out=mbs[0], phase="place")))
print("raster alone %.4f ms" % timed(lambda: edage.generate_maps(pbs[0], PL, 5, K, out=mbs[0], phase="raster")))
print("fused kernel %.4f ms" % timed(lambda: edage.generate_maps(pbs[0], PL, 5, K, out=mbs[0])))
print("place then raster, one stream %.4f ms" % timed(lambda: (edage.generate_maps(pbs[0], PL, 5, K, out=mbs[0], phase="place"), edage.generate_maps(pbs[0], PL, 5, K, out=mbs[0], phase="raster"))))

sp, sr, sa = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def pipeline(n, with_paths):
    placed = [None, None]; rastered = [None, None]; pready = [None] * 4; pfree = [None] * 4
    for it in range(n):
        mbi, pbi = it % 2, it % 4
        if with_paths:
            with torch.cuda.stream(sa):
                if pfree[pbi] is not None:
                    sa.wait_event(pfree[pbi])
                edage.generate_paths(PATHS, R, 50, 3, seed=0, first_path_id=it * PATHS, device=dev, out=pbs[pbi])
                pready[pbi] = torch.cuda.Event(); pready[pbi].record(sa)
        with torch.cuda.stream(sp):
            if rastered[mbi] is not None:
                sp.wait_event(rastered[mbi])
            if pready[pbi] is not None:
                sp.wait_event(pready[pbi])
            edage.generate_maps(pbs[pbi], PL, 5, K, first_map_id=it * PATHS * PL, out=mbs[mbi], phase="place")
            placed[mbi] = torch.cuda.Event(); placed[mbi].record(sp)
        with torch.cuda.stream(sr):
            sr.wait_event(placed[mbi])
            edage.generate_maps(pbs[pbi], PL, 5, K, out=mbs[mbi], phase="raster")
            rastered[mbi] = torch.cuda.Event(); rastered[mbi].record(sr)
            pfree[pbi] = rastered[mbi]


for wp in (False, True):
    pipeline(6, wp); torch.cuda.synchronize()
    t = time.perf_counter(); pipeline(40, wp); th = time.perf_counter() - t; torch.cuda.synchronize()
    print("pipelined place|raster%s: %.4f ms / batch (host enqueue %.4f ms / batch)" % (" + paths" if wp else "", (time.perf_counter() - t) / 40 * 1e3, th / 40 * 1e3))
    # GPU-side only: hold all three streams behind a sleep while the host enqueues, then time with events
    gate = torch.cuda.Event(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(400_000_000)
    e0.record(); gate.record()
    for st in (sp, sr, sa):
        st.wait_event(gate)
    pipeline(40, wp)
    torch.cuda.current_stream().wait_stream(sr); torch.cuda.current_stream().wait_stream(sp); torch.cuda.current_stream().wait_stream(sa)
    e1.record(); torch.cuda.synchronize()
    print("   gated (no host in the loop): %.4f ms / batch" % (e0.elapsed_time(e1) / 40))
