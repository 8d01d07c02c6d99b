#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py`: per kernel, the launches of the
timed region on the FINEST level it runs on (--stats averages a kernel over all 14 multigrid
levels and over the launch shapes the tuner tries before the timed region).

Which launches those are: a kernel that reads the iterate (template argument ZERO_IN = false)
runs on level 0 only, in a launch shape the tuner measures first; the shape in use afterwards is
the grid of the kernel's LAST dispatch, so only launches of that grid count.  A kernel with
ZERO_IN = true runs on every coarser level in one fixed shape per level; the grid whose launches
take longest is the finest level.

  python profiles/summarize.py gpurun_out/<dir>/<host>/<pid>_kernel_trace.csv > profiles/<name>.txt
"""
import collections
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        if "mgx::" not in name:
            continue
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        by[name].append((grid, dur, int(r["VGPR_Count"]), int(r["SGPR_Count"]), int(r["LDS_Block_Size"]),
                         int(r["Scratch_Size"])))
    print("%-100s %7s %7s %12s %12s %12s %12s %6s %5s %6s %7s" % ("kernel (timed-region launches on its finest level)",
                                                                 "calls", "all", "avg_us", "median_us", "min_us",
                                                                 "max_us", "vgpr*", "sgpr", "lds", "scratch"))
    tot = sum(d for v in by.values() for (_, d, *_r) in v)
    for name, v in sorted(by.items(), key=lambda kv: -sum(d for (_, d, *_r) in kv[1])):
        tuned = "k_stream" in name and ", false, " in name
        if tuned:
            gsel = v[-1][0]
        else:  # the finest level is the grid whose launches take longest (chunk lengths differ per level)
            per = collections.defaultdict(list)
            for (g, d, *_r) in v:
                per[g].append(d)
            gsel = max(per, key=lambda g: sum(per[g]) / len(per[g]))
        fine = [d for (g, d, *_r) in v if g == gsel]
        _, _, vg, sg, lds, scr = v[0]
        short = name if len(name) <= 100 else name[:97] + "..."
        med = sorted(fine)[len(fine) // 2]
        print("%-100s %7d %7d %12.1f %12.1f %12.1f %12.1f %6d %5d %6d %7d" % (short, len(fine), len(v),
                                                                              sum(fine) / len(fine) / 1e3, med / 1e3,
                                                                              min(fine) / 1e3, max(fine) / 1e3, vg, sg,
                                                                              lds, scr))
    print("(calls = launches of the shape the timed region uses, incl. the launches of it that mgx_plan_tune times; "
          "all = every launch of the kernel, every level and trial shape)")
    print("(*vgpr: the trace's encoding of the allocation, not the register count; the ISA of the level-0 up+down "
          "instance uses 149 vector registers, two waves per SIMD)")
    print("total mgx kernel time: %.3f ms over %d launches" % (tot / 1e6, sum(len(v) for v in by.values())))


if __name__ == "__main__":
    main(sys.argv[1])
