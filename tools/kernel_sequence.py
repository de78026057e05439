#!/usr/bin/env python3
"""The last iteration of a rocprofv3 --kernel-trace CSV as an ordered list (start offset, duration, name): shows WHERE the small
framework kernels sit between the hand-written ones.  usage: kernel_sequence.py <dir> <marker-substring> <marker-calls-per-iteration>"""
import csv, glob, re, sys
d, marker, per_iter = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
sel = rows[max(idx[-per_iter] - 12, 0):]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    n = r["Kernel_Name"]
    n = re.sub(r"at::native::", "", n)
    m = re.search(r"(\w+Functor\w*|\w+_kernel_cuda\w*|direct_copy_kernel\w*|\w+_kernel)\b", n)
    print("%9.3f %8.3f  %s | %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                                   n[:70], n[70:260] if "elementwise" in n else ""))
