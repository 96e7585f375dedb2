"""Condense rocprofv3 output into the small summaries kept under profiles/.

  python tools/profile_summary.py stats  <..._kernel_stats.csv>        <out.csv> [top_n]
  python tools/profile_summary.py pmc    <..._counter_collection.csv>  <out.csv>
  python tools/profile_summary.py record <..._kernel_stats.csv> <out.json> <source note> \
                                         [batch seq kv model quant]
  python tools/profile_summary.py attn_traffic <fetch pmc summary> <write pmc summary> <out.json> <note>
  python tools/profile_summary.py gemm_traffic <fetch pmc summary> <write pmc summary> <out.json> <note>

`record` writes the small JSON that bench.py reads for `roofline.rocprof_avg_us`: the average
duration of the paged-attention kernel and the summed averages of one layer's four int4 GEMM
launches (+ their split-K reductions) in a `rocprofv3 --kernel-trace --stats` run of bench.py.
`stats` keeps the top kernels by total time with names cut to 120 characters.  `pmc` averages each
counter per kernel name over the dispatches in the file; FETCH_SIZE / WRITE_SIZE come out of
rocprofv3 in KiB, and on gfx950 FETCH_SIZE counts 64-byte requests as 32 bytes, so the bytes
columns apply x1024 and (for FETCH_SIZE) x2.
"""
import csv
import re
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


def record(src, dst, note, workload):
    import json
    rows = list(csv.DictReader(open(src)))
    attn = [r for r in rows if "pa_mfma_kernel" in r["Name"]]
    attn.sort(key=lambda r: -float(r["TotalDurationNs"]))
    gemm = [r for r in rows if "q4_gemm" in r["Name"] or "q4_splitk_reduce" in r["Name"]]
    steps = None
    rec = {"comment": "average kernel durations in a rocprofv3 --kernel-trace --stats run of "
                      "bench.py on an MI355X; read by bench.py for roofline.rocprof_avg_us",
           "source": note, "workload": workload}
    if attn:
        a = attn[0]
        rec["paged_attention_kernel"] = short(a["Name"])
        rec["paged_attention_calls"] = int(a["Calls"])
        rec["paged_attention_us"] = round(float(a["AverageNs"]) / 1e3, 2)
    if gemm:
        # per layer: every GEMM kernel variant's total time / number of layer passes in the trace;
        # a layer pass launches four GEMMs, so passes = all GEMM launches / 4
        launches = sum(int(r["Calls"]) for r in gemm if "q4_gemm" in r["Name"])
        total_ns = sum(float(r["TotalDurationNs"]) for r in gemm)
        rec["q4_gemm_launches"] = launches
        rec["q4_gemm_layer_us"] = round(total_ns / (launches / 4) / 1e3, 2)
        rec["q4_gemm_kernels"] = {short(r["Name"]): {"calls": int(r["Calls"]),
                                                     "avg_us": round(float(r["AverageNs"]) / 1e3, 2)}
                                  for r in gemm}
    with open(dst, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")


def gemm_traffic(fetch_csv, write_csv, dst, note):
    """HBM bytes of one layer's four int4 GEMM launches at M = 256 from two PMC summaries (this
    tool's `pmc` output for FETCH_SIZE and for WRITE_SIZE of `tools/gemm_bench.py --m 256`): the
    unsplit kernel is gate_up; each split-K kernel serves as many of qkv, o and down as its dispatch
    count says (every shape is launched equally often)."""
    import json

    def per_kernel(path, counter):
        out = {}
        for r in csv.DictReader(open(path)):
            if r["Counter"].startswith(counter) and re.search(r"q4_gemm_(m32|dma2?|dma_kp)_kernel", r["Kernel"]):
                out[r["Kernel"]] = (float(r["AvgBytes(corrected)"]), int(r["Dispatches"]))
        return out

    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    rec = {"comment": "HBM bytes per layer of the int4 GEMMs (qkv + o + gate_up + down, M = 256) from "
                      "rocprofv3 PMC passes, FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x "
                      "1024; read by bench.py for roofline_gemm.traffic when the workload matches",
           "command": note,
           "workload": {"m": 256, "model": "llama3-8b", "quant": "int4", "rank_of": 1, "world": 1},
           "kernels": {}}
    total = 0.0
    # every layer shape is launched equally often: a kernel's launches per layer are its
    # dispatches over those of the unsplit kernel (gate_up, one launch per layer)
    unsplit = [n for name, (_, n) in f.items()
               if re.search(r"q4_gemm_\w+?_kernelILi\dELb(\d)E", name).group(1) == "0"]
    per_layer = min(unsplit) if unsplit else max(n for _, n in f.values())
    for name in f:
        fb, n = f[name]
        wb = w.get(name, (0.0, 0))[0]
        launches = max(1, round(n / per_layer))
        rec["kernels"][name] = {"fetch_bytes": round(fb), "write_bytes": round(wb), "dispatches": n,
                                "launches_per_layer": launches}
        total += launches * (fb + wb)
    rec["bytes_per_layer"] = round(total)
    with open(dst, "w") as fh:
        json.dump(rec, fh, indent=1)
        fh.write("\n")


def attn_traffic(fetch_csv, write_csv, dst, note, workload):
    """HBM bytes of one paged-attention launch from two PMC summaries (this tool's `pmc` output for
    FETCH_SIZE and for WRITE_SIZE of `tools/attn_bench.py` on the benchmark's shape).  The record
    names the kernel instantiation the counters were taken on; bench.py uses the record only when
    that name is the one its own launch runs."""
    import json

    def row(path, counter):
        best = None
        for r in csv.DictReader(open(path)):
            if r["Counter"].startswith(counter) and "pa_mfma_kernel" in r["Kernel"]:
                if best is None or int(r["Dispatches"]) > int(best["Dispatches"]):
                    best = r
        if best is None:
            raise SystemExit(f"no pa_mfma_kernel row with {counter} in {path}")
        return best

    f, w = row(fetch_csv, "FETCH_SIZE"), row(write_csv, "WRITE_SIZE")
    m = re.search(r"pa_mfma_kernel<[^>]*>", f["Kernel"])
    if m is None or m.group(0) not in w["Kernel"]:
        raise SystemExit("the two passes did not run the same kernel instantiation")
    rec = {"comment": "HBM bytes per paged_attention_v1 launch from rocprofv3 PMC passes (one counter "
                      "per pass), read by bench.py for roofline.traffic when workload and kernel match. "
                      "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 counts 64-byte "
                      "requests as 32).",
           "command": note, "workload": workload, "kernel": m.group(0),
           "dispatches": int(f["Dispatches"]),
           "fetch_size_kib_avg": float(f["AvgValue"]), "write_size_kib_avg": float(w["AvgValue"]),
           "fetch_bytes": int(f["AvgBytes(corrected)"]), "write_bytes": int(w["AvgBytes(corrected)"])}
    with open(dst, "w") as fh:
        json.dump(rec, fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    if sys.argv[1] == "attn_traffic":
        attn_traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5],
                     {"batch": 256, "seq": 4096, "kv": "fp8", "num_heads": 32, "num_kv_heads": 8,
                      "head_size": 128, "block_size": 16})
        sys.exit(0)
    if sys.argv[1] == "gemm_traffic":
        gemm_traffic(*sys.argv[2:6])
        sys.exit(0)
    mode, src, dst = sys.argv[1:4]
    if mode == "record":
        wl = sys.argv[5:10]
        workload = ({"batch": int(wl[0]), "seq": int(wl[1]), "kv": wl[2], "model": wl[3], "quant": wl[4]}
                    if len(wl) == 5 else {"batch": 256, "seq": 4096, "kv": "fp8", "model": "llama3-8b",
                                          "quant": "awq"})
        record(src, dst, sys.argv[4], workload)
    elif mode == "stats":
        stats(src, dst, int(sys.argv[4]) if len(sys.argv) > 4 else 40)
    else:
        pmc(src, dst)
