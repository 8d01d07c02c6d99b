#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel, the launches on the
FINEST level only (largest grid of that kernel), because --stats averages a
kernel over all 14 multigrid levels.

  python profiles/summarize.py gpurun_out/<dir>/<host>/<pid>_kernel_trace.csv > profiles/<name>.txt
"""
import collections
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    by = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        if "mgx::" not in name:
            continue
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        by[name].append((grid, dur, int(r["VGPR_Count"]), int(r["SGPR_Count"]), int(r["LDS_Block_Size"]),
                         int(r["Scratch_Size"])))
    print("%-100s %7s %7s %12s %12s %12s %12s %5s %5s %6s %7s" % ("kernel (finest-level launches only)", "calls", "all",
                                                                 "avg_us", "median_us", "min_us", "max_us", "vgpr",
                                                                 "sgpr", "lds", "scratch"))
    tot = sum(d for v in by.values() for (_, d, *_r) in v)
    for name, v in sorted(by.items(), key=lambda kv: -sum(d for (_, d, *_r) in kv[1])):
        gmax = max(g for (g, *_r) in v)
        fine = [d for (g, d, *_r) in v if g == gmax]
        _, _, vg, sg, lds, scr = v[0]
        short = name if len(name) <= 100 else name[:97] + "..."
        med = sorted(fine)[len(fine) // 2]
        print("%-100s %7d %7d %12.1f %12.1f %12.1f %12.1f %5d %5d %6d %7d" % (short, len(fine), len(v),
                                                                              sum(fine) / len(fine) / 1e3, med / 1e3,
                                                                              min(fine) / 1e3, max(fine) / 1e3, vg, sg,
                                                                              lds, scr))
    print("(the average includes the launches of mgx_plan_tune, which run every candidate launch shape; the median is "
          "the shape the timed region uses)")
    print("total mgx kernel time: %.3f ms over %d launches" % (tot / 1e6, sum(len(v) for v in by.values())))


if __name__ == "__main__":
    main(sys.argv[1])
