#!/usr/bin/env python3
"""Idle time between kernels in a rocprofv3 kernel trace (csv): per burst of back-to-back kernels, its span, the
summed kernel time, and the largest gaps with the kernels either side."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000        # a gap longer than this (ns) ends a burst
bursts, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > thr:
        bursts.append(cur)
        cur = []
    cur.append(b)
bursts.append(cur)
short = lambda n: n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:40]
for k, bu in enumerate(bursts):
    span = int(bu[-1]["End_Timestamp"]) - int(bu[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in bu)
    gaps = sorted(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"]), short(a["Kernel_Name"]), short(b["Kernel_Name"]))
                   for a, b in zip(bu, bu[1:])), reverse=True)
    print(f"burst {k}: {len(bu)} kernels, span {span / 1e3:.0f} us, busy {busy / 1e3:.0f} us, idle {(span - busy) / 1e3:.0f} us")
    for g in gaps[:4]:
        print(f"     gap {g[0] / 1e3:.1f} us after {g[1]} before {g[2]}")
