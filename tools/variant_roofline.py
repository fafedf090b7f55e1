#!/usr/bin/env python3
"""Roofline rows for the kernels that only the secondary workloads of bench.py run (two-memory-row models, the YOLO
head, long-memory decode, the soft-attention LSTM, beam search), from the kernel
statistics of one profiled run of the whole bench line:

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o p -- python3 bench.py --steps 3 --train-steps 3 --no-cpu-baseline
    python3 tools/variant_roofline.py OUT/p_kernel_stats.csv > profiles/<set>_variant_kernel_roofline.md

The average duration of a kernel name is taken over all its launches in that run; rows are given only for kernels
that one workload launches with one shape (stated), with the algorithmic work of that shape."""
import csv
import sys

PEAK_TF, PEAK_GBS = 157.3, 8000.0
E, H, V, B = 512, 4, 10000, 1024

# (substring of the kernel name, workload and shape, bound, algorithmic work per launch [FLOP or bytes], note)
ROWS = [
        ("attention_memory_kernel", "cfg4-as-worded decode: 1024 rows x 261 raw memory rows", "hbm", 1.0 * B * 261 * E * 4, "the memory once per row"),
    ("pair_gate_fwd_kernel", "cfg4 training: live rows of 512 x 20 (about 7.5 k), H=4", "hbm", 7488 * 2 * E * 4.0 + 512 * (2 * H + 1) * E * 4.0,
     "x1 read, a2 written per row; G, D per head and c1 once per sample (the 20 rows of a sample share them in cache)"),
    ("pair_gate_bwd_kernel", "cfg4 training: the same rows", "hbm", 7488 * 3 * E * 4.0 + 512 * 4 * H * E * 4.0,
     "d_a2, x1 read, d_x1 written per row; G, D read and their gradients written once per sample"),
    ("additive_attention_bwd_kernel", "soft-attention LSTM training: B=512, S=64, A=512, E=512, per time step", "hbm",
     512 * 64 * (2 * 512 + 2 * E) * 4.0, "pf, enc read, dpf, denc accumulated"),
    ("additive_attention_kernel", "soft-attention LSTM decode/training step: B=1024/512, S=64 (mixed)", "hbm", 768 * 64 * (512 + E) * 4.0,
     "pf and enc once (mean of the two batch sizes)"),
    ("beam_topk_kernel", "beam search W=3: 3072 hypotheses x 10000 logits", "hbm", 3072 * V * 4.0, "the logits row once"),
    ("beam_select_kernel", "beam search W=3: 1024 samples x 9 candidates", "hbm", 1024 * 9 * 8.0 + 3072 * 21 * 4.0, "candidates + token prefixes"),
    ("beam_cache_gather_kernel", "beam search W=3: KV caches re-gathered by parent", "hbm", 2.0 * 3072 * 10.5 * E * 4, "one cache, read + write, mean 10.5 positions"),
    ("splitk_reduce_kernel", "cfg5 head: 1024 x 512 outputs, partial slabs of the 86436-deep contraction", "hbm", 1024 * E * 4.0 * 17, "about 16 slabs read + 1 write"),
]


def main():
    stats = list(csv.DictReader(open(sys.argv[1])))
    print("| kernel | workload / shape | launches | avg µs | algorithmic work | achieved | of roof | note |")
    print("|---|---|---|---|---|---|---|---|")
    for key, shape, bound, work, note in ROWS:
        hit = [r for r in stats if key in r["Name"]]
        if not hit:
            print(f"| `{key}` | {shape} | 0 | - | - | - | - | not launched in this run |")
            continue
        calls = sum(int(r["Calls"]) for r in hit)
        us = sum(float(r["TotalDurationNs"]) for r in hit) / calls / 1e3
        if bound == "mfma":
            tf = work / us / 1e6
            print(f"| `{key}` | {shape} | {calls} | {us:.1f} | {work / 1e9:.2f} GFLOP | {tf:.1f} TFLOP/s | {100 * tf / PEAK_TF:.0f} % of the fp32 MFMA peak | {note} |")
        else:
            gbs = work / us / 1e3
            print(f"| `{key}` | {shape} | {calls} | {us:.1f} | {work / 1e6:.1f} MB | {gbs:.0f} GB/s | {100 * gbs / PEAK_GBS:.0f} % of 8 TB/s | {note} |")


if __name__ == "__main__":
    main()
