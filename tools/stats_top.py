"""Prints the top rows of a rocprofv3 kernel_stats.csv: calls, total ms, average us, share, name (diagnostic)."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms" % (tot / 1e6))
for r in rows[:n]:
    print("%6d  %9.3f ms  %9.1f us  %5.1f%%  %s" % (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                 float(r["Percentage"]), r["Name"][:110]))
