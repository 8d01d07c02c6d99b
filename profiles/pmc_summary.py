#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each in
its own run, never combined with a trace).

  python profiles/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv>

Per (kernel, grid): KiB per launch averaged over the launches.  A kernel whose launch shape the
tuner measures appears with several grids; the grid of its LAST dispatch is the shape the timed
region runs and is marked so.  gfx950 tallies a 128-byte
read request as 64 bytes in FETCH_SIZE (16-byte-per-lane streaming reads), so the HBM-side
traffic is 2*FETCH_SIZE + WRITE_SIZE; the runtime's own buffer copy of one 2.147 GB array in
the same kind of run reports FETCH 1.074 GB and WRITE 2.148 GB, which confirms the factor.
"""
import collections
import csv
import sys


LAST_GRID = {}  # kernel -> grid of its last dispatch: the launch shape in use once the tuner has decided


def load(path, counter):
    acc = collections.defaultdict(list)
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0) or 0))
    for r in rows:
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        if "mgx::" not in name:
            continue
        grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])
        acc[(name, grid)].append(float(r["Counter_Value"]))
        LAST_GRID[name] = grid
    return acc


def main(fetch_csv, write_csv):
    F = load(fetch_csv, "FETCH_SIZE")
    W = load(write_csv, "WRITE_SIZE")
    print("KiB per launch, averaged over the launches of each (kernel, grid); corrected = 2*FETCH + WRITE")
    for key in sorted(F, key=lambda k: -sum(F[k])):
        name, grid = key
        f = sum(F[key]) / len(F[key])
        w = sum(W[key]) / len(W[key]) if key in W else float("nan")
        short = name.replace("void mgx::(anonymous namespace)::", "").replace("(mgx::StreamArgs)", "")
        print("%-40s grid=%9d launches=%3d FETCH_KiB=%12.1f WRITE_KiB=%12.1f corrected=%.3f GB%s"
              % (short[:40], grid, len(F[key]), f, w, (2 * f + w) * 1024 / 1e9,
                 "  <- timed-region shape" if (LAST_GRID.get(name) == grid and "k_stream" in name and ", false, " in name)
                 else ""))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
