#!/usr/bin/env python3
"""Kernel study: per-layer kernel times of a decode step from a rocprofv3 kernel trace of bench.py.

    python3 tools/layer_trace.py <..._kernel_trace.csv> [layers]

Groups the library's kernels by (name, grid), prints the median duration of each group and how many
launches it has per layer, and sums them to a per-layer total with and without paged attention --
the step's run-to-run noise is mostly the attention launch (HBM), which hides changes of a few
microseconds per layer elsewhere.
"""
import collections
import csv
import re
import sys


def main():
    path = sys.argv[1]
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    rows = list(csv.DictReader(open(path)))
    groups = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        if not re.search(r"vla", n):
            continue
        m = re.search(r"(\w+_kernel)", n)
        key = (m.group(1) if m else n[:40], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        groups[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    steps = None
    for key, v in groups.items():
        if "pa_mfma" in key[0] or "pa_generic" in key[0]:
            steps = len(v) / layers
    total = other = 0.0
    for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        if steps is None or len(v) < layers * steps * 0.5:
            continue
        v.sort()
        per_layer = len(v) / (layers * steps)
        med = v[len(v) // 2] / 1e3
        mean = sum(v) / len(v) / 1e3
        print(f"{key[0][:34]:34s} grid {key[1]:>7s} x {key[2]:>3s} x {key[3]:>2s}  per layer {per_layer:4.1f}  "
              f"median {med:7.1f} us  mean {mean:7.1f}  min {v[0] / 1e3:7.1f}")
        total += mean * per_layer
        if "pa_" not in key[0]:
            other += mean * per_layer
    print(f"per layer: {total:.1f} us, without paged attention {other:.1f} us")


if __name__ == "__main__":
    main()
