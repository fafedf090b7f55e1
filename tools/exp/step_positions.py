#!/usr/bin/env python3
"""Average duration of each kernel of a greedy decode step, by its position in the step, from a rocprofv3 kernel
trace (csv) of bench.py."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
bursts, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 100000:
        bursts.append(cur)
        cur = []
    cur.append(b)
bursts.append(cur)
big = [b for b in bursts if len(b) > 300][-1]
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
steps, cur = [], []
for r in big:
    cur.append(r)
    if "cached_advance" in r["Kernel_Name"]:
        steps.append(cur)
        cur = []
n = len(steps[5])
pos = collections.defaultdict(list)
for st in steps[3:18]:
    for k, r in enumerate(st[-n:]):
        pos[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), short(r["Kernel_Name"])))
tot = 0
for k in sorted(pos):
    v = pos[k]
    avg = sum(d for d, _ in v) / len(v) / 1e3
    tot += avg
    print(f"{k:2d} {v[0][1]:50s} {avg:6.1f}")
print(f"   step total {tot:.1f} us")
