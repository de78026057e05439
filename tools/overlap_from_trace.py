"""Reads a rocprofv3 kernel-trace CSV and reports how much of each kernel's time ran concurrently with other kernels."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
rows = rows[-240:]
ov = collections.defaultdict(lambda: [0, 0, 0])
for i, (s, e, n, q, st) in enumerate(rows):
    o = 0
    for j, (s2, e2, n2, _, _) in enumerate(rows):
        if i != j:
            o += max(0, min(e, e2) - max(s, s2))
    ov[(n, q, st)][0] += e - s; ov[(n, q, st)][1] += o; ov[(n, q, st)][2] += 1
for k, (d, o, c) in ov.items():
    print(f"{k}: n={c} avg {d / c / 1e3:.1f} us, overlapped with others {o / c / 1e3:.1f} us")
t0 = rows[0][0]
for s, e, n, q, st in rows[-24:]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f}  q{q} s{st} {n}")
