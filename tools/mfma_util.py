#!/usr/bin/env python3
"""MFMA utilisation per kernel from two rocprofv3 passes over tools/mfma_workload.py (--output-format csv):

    tools/mfma_util.py <pmc counter_collection.csv> <kernel_trace.csv> > profiles/<tag>_mfma_util.md

Launches are grouped by kernel and grid size (one template instance serves several shapes).

PMC pass: SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE.
  MFMA busy  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): busy cycles summed over the chip's 1024 SIMDs
               against the dispatch's active cycles (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs) -- the expression of
               the derived metric MfmaUtil.
  MFMA FLOPs = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512 (the counter's unit is 512 FLOP).
Kernel-trace pass (no counters, so its durations are undisturbed): TFLOP/s = MFMA FLOPs / average duration.
v_mfma_f32_32x32x2_f32 issues once per 64 cycles per SIMD (4096 FLOP): 100 % busy at 2.4 GHz is 157.3 TFLOP/s, so
TFLOP/s / 157.3 and the busy fraction agree when the clock holds 2.4 GHz and sit apart by the clock ratio when it does
not (the chip runs sustained fp32 MFMA loops at about 2.1 GHz)."""
import csv
import sys
from collections import defaultdict

PEAK = 157.3


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def main():
    pmc, trace = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    for r in csv.DictReader(open(pmc)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    dur = {}
    for r in csv.DictReader(open(trace)):
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        d = dur.setdefault((short(r["Kernel_Name"]), grid), [0.0, 0])
        d[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d[1] += 1
    print("| kernel | grid (threads) | launches | MFMA busy (PMC) | MFMA GFLOP / launch (PMC) | avg duration (kernel trace) | TFLOP/s | of 157.3 |")
    print("|---|---|---|---|---|---|---|---|")
    rows = []
    for k, c in acc.items():
        mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
        if mops <= 0:
            continue
        calls = len(n[k])
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        gflop = mops * 512.0 / calls / 1e9
        avg_us = dur[k][0] / dur[k][1] / 1e3 if k in dur else float("nan")
        tf = gflop * 1e9 / (avg_us * 1e-6) / 1e12
        rows.append((c["SQ_VALU_MFMA_BUSY_CYCLES"], k, calls, busy, gflop, avg_us, tf))
    for _, k, calls, busy, gflop, avg_us, tf in sorted(rows, reverse=True):
        print(f"| `{k[0]}` | {k[1]} | {calls} | {busy:.1%} | {gflop:.3f} | {avg_us:.1f} us | {tf:.1f} | {tf / PEAK:.1%} |")


if __name__ == "__main__":
    main()
