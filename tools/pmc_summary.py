#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per dispatch, per kernel."""
import csv
import glob
import sys
from collections import defaultdict

def main(pattern):
    files = glob.glob(pattern, recursive=True)
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                short = k.split("(")[0][-70:]
                c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0) or 0)
                a = agg[short][c]; a[0] += v; a[1] += 1
    for k in sorted(agg):
        print(k)
        for c, (s, n) in sorted(agg[k].items()):
            print(f"    {c:34s} mean/dispatch {s / n:16.1f}   dispatches {n}")

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc/**/*counter_collection.csv")
