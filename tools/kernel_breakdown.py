#!/usr/bin/env python3
"""Per-kernel time of the LAST iteration in a rocprofv3 --kernel-trace CSV (steady state, past library warm-up).
usage: kernel_breakdown.py <dir> <marker-kernel-substring> <marker-calls-per-iteration>"""
import collections, csv, glob, sys
d, marker, per_iter = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
# per_iter > 0: the tail of the trace from 12 dispatches before the iteration's first marker call (the marker closes a PPNet batch);
# per_iter == 0: everything between the last two marker calls (one whole step of a loop that ends with the marker)
sel = rows[max(idx[-per_iter] - 12, 0):] if per_iter > 0 else rows[idx[-2] + 1:idx[-1] + 1]
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    k = r["Kernel_Name"][:100]
    agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"last iteration: wall {(int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e6:.2f} ms, kernels {tot / 1e6:.2f} ms, {len(sel)} dispatches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[4]) if len(sys.argv) > 4 else 25]:
    print(f"{v[0] / 1e6:8.2f} ms {100 * v[0] / tot:5.1f}% x{v[1]:4d}  {k}")
