#!/usr/bin/env python3
"""Kernel durations and the gaps between consecutive kernels from a rocprofv3 --kernel-trace CSV (last N rows)."""
import csv
import sys

path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{r['Kernel_Name'][:70]:70s} dur {(e - s) / 1e3:8.2f} us  gap {gap:7.2f} us")
    prev_end = e
