#!/usr/bin/env python3
"""Durations of the one-launch decoder layers and of the vocabulary kernel per decode step, from a rocprofv3
--kernel-trace database (rocpd) of a run of decodes of 20 steps: the mean over the decodes for each step index."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
sfx = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0][len("rocpd_kernel_dispatch"):]
ks = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from rocpd_info_kernel_symbol{sfx}")}
rows = list(db.execute(f"select kernel_id, start, end from rocpd_kernel_dispatch{sfx} order by start"))
for pat in ("decode_layers_team_kernel", "vocab_reduce_kernel"):
    d = [(r[2] - r[1]) / 1e3 for r in rows if pat in ks[r[0]]]
    n = len(d) // steps
    d = d[len(d) - n * steps:]
    per = [sum(d[i * steps + t] for i in range(n)) / n for t in range(steps)]
    print(pat, f"{n} decodes; us per step index:", " ".join(f"{v:.0f}" for v in per), f"| mean {sum(per) / steps:.1f}")
