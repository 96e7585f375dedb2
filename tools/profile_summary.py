"""Condense rocprofv3 output into the small summaries kept under profiles/.

  python tools/profile_summary.py stats  <..._kernel_stats.csv>        <out.csv> [top_n]
  python tools/profile_summary.py pmc    <..._counter_collection.csv>  <out.csv>

`stats` keeps the top kernels by total time with names cut to 120 characters.  `pmc` averages each
counter per kernel name over the dispatches in the file; FETCH_SIZE / WRITE_SIZE come out of
rocprofv3 in KiB, and on gfx950 FETCH_SIZE counts 64-byte requests as 32 bytes, so the bytes
columns apply x1024 and (for FETCH_SIZE) x2.
"""
import csv
import collections
import sys


def short(name: str) -> str:
    return name if len(name) <= 120 else name[:117] + "..."


def stats(src, dst, top_n=40):
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:top_n]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        r["Percentage"], r["MinNs"], r["MaxNs"]])


def pmc(src, dst):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(src)):
        k = (short(r["Kernel_Name"]), r["Counter_Name"])
        a = acc.setdefault(k, [0, 0.0, r["Grid_Size"], r["VGPR_Count"], r["LDS_Block_Size"]])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Dispatches", "AvgValue", "AvgBytes(corrected)", "Grid",
                    "VGPRs", "LDS"])
        for (name, ctr), (n, tot, grid, vg, lds) in acc.items():
            avg = tot / n
            b = ""
            if ctr.startswith("FETCH_SIZE"):
                b = f"{avg * 1024 * 2:.0f}"
            elif ctr.startswith("WRITE_SIZE"):
                b = f"{avg * 1024:.0f}"
            w.writerow([name, ctr, n, f"{avg:.3f}", b, grid, vg, lds])


if __name__ == "__main__":
    mode, src, dst = sys.argv[1:4]
    if mode == "stats":
        stats(src, dst, int(sys.argv[4]) if len(sys.argv) > 4 else 40)
    else:
        pmc(src, dst)
