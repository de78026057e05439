#!/usr/bin/env python3
"""Reduce two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — separate runs, MI355X_MICROARCH.md §HBM) of
`bench.py` to HBM bytes per edage_maps_kernel launch and write profiles/traffic_maps_kernel.json.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write <round-tag>

gfx950 corrections: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> doubled;
WRITE_SIZE reads the bytes exactly for 16-byte-per-lane stores.  Both counters are in KiB.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(d, counter, kernel="edage_maps_kernel"):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_dir, write_dir, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch_kib, nf = mean_counter(fetch_dir, "FETCH_SIZE")
    write_kib, nw = mean_counter(write_dir, "WRITE_SIZE")
    out = {"round": tag, "kernel": "edage_maps_kernel", "launches_averaged": [nf, nw],
           "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
           "hbm_bytes_per_launch": round(2 * fetch_kib * 1024 + write_kib * 1024),
           "note": "FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request); per launch = 10000 maps at R=256, K=20"}
    with open(os.path.join(ROOT, "profiles", "traffic_maps_kernel.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
