#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) of `bench.py --steps 1 --warmup 0
--train-steps 0 --no-cpu-baseline` into profiles/vocab_traffic.json.

    tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [kernel substring] [name of the measurement set]

Both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes,
so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv
import json
import os
import sys


def per_launch(path, counter, needle):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and needle in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_csv, write_csv = sys.argv[1], sys.argv[2]
    needle = sys.argv[3] if len(sys.argv) > 3 else "vocab_reduce_kernel"
    f, n = per_launch(fetch_csv, "FETCH_SIZE", needle)
    w, _ = per_launch(write_csv, "WRITE_SIZE", needle)
    out = {"kernel": needle, "set": sys.argv[4] if len(sys.argv) > 4 else "unnamed", "launches": n, "fetch_size_kib": f, "write_size_kib": w,
           "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
           "note": "FETCH_SIZE x2 (gfx950 counts 128-B read requests as 64 B) + WRITE_SIZE, KiB -> bytes"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "vocab_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
