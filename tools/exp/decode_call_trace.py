#!/usr/bin/env python3
"""Kernel timeline of the last complete decode call in a rocprofv3 --kernel-trace database (rocpd) of bench.py's decode
leg: everything that is not one of the per-step pair (layers, vocabulary), with start offsets, plus the call's span."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
sfx = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0][len("rocpd_kernel_dispatch"):]
ks = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from rocpd_info_kernel_symbol{sfx}")}
rows = list(db.execute(f"select kernel_id, start, end, grid_size_x, workgroup_size_x from rocpd_kernel_dispatch{sfx} order by start"))
idx = [i for i, r in enumerate(rows) if "cached_init" in ks[r[0]]]
a, b = idx[-2], idx[-1]
# step back to the feature head of that call
while a > 0 and "vocab_reduce" not in ks[rows[a - 1][0]] and a > idx[-2] - 8:
    a -= 1
t0 = rows[a][1]
prev = t0
for r in rows[a:b]:
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", ks[r[0]])[:56]
    if "decode_layers" in n or "vocab_reduce" in n:
        prev = max(prev, r[2])
        continue
    print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f} gap{(r[1] - prev) / 1e3:6.1f} g{r[3] // max(r[4], 1):6d} {n}")
    prev = max(prev, r[2])
print(f"span {(rows[b][1] - t0) / 1e3:.1f} us")
