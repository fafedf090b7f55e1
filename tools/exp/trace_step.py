#!/usr/bin/env python3
"""Prints the kernel timeline of the last complete step of a rocprofv3 --kernel-trace database (rocpd format):
start offset, duration, gap to the previous kernel's end, stream, grid, name.  A step starts at `embed_train_fwd_kernel`."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "embed_train_fwd"
tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
sfx = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0][len("rocpd_kernel_dispatch"):]
ks = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from rocpd_info_kernel_symbol{sfx}")}
rows = list(db.execute(f"select kernel_id, start, end, stream_id, grid_size_x, grid_size_y, workgroup_size_x "
                       f"from rocpd_kernel_dispatch{sfx} order by start"))
idx = [i for i, r in enumerate(rows) if marker in ks[r[0]]]
a, b = idx[-2], idx[-1]
t0 = rows[a][1]
prev_end = t0
busy = 0
for r in rows[a:b]:
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", ks[r[0]])
    n = re.sub(r"\.kd$", "", n)[:60]
    print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f} gap{(r[1] - prev_end) / 1e3:6.1f} s{r[3]} "
          f"g{r[4] // r[6]:6d}x{r[5]:<3d} {n}")
    prev_end = max(prev_end, r[2])
    busy += r[2] - r[1]
print(f"kernels {b - a}, span {(rows[b][1] - t0) / 1e3:.1f} us, sum of durations {busy / 1e3:.1f} us")
