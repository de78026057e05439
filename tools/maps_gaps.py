"""From a rocprofv3 kernel trace of bench.py: the idle time between consecutive stage-B (edage_maps) kernels (diagnostic)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if "edage_maps_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]                       # the timed half
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
gap = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
gap.sort()
n = len(gap)
print(f"{len(rows)} launches: kernel avg {sum(dur) / len(dur) / 1e3:.2f} us; gap avg {sum(gap) / n / 1e3:.2f} us, median {gap[n // 2] / 1e3:.2f}, p10 {gap[n // 10] / 1e3:.2f}, p90 {gap[9 * n // 10] / 1e3:.2f}")
