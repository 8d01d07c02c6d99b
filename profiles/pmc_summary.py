#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each in
its own run, never combined with a trace).

  python profiles/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv>

Per (kernel, grid): KiB per launch averaged over the launches.  gfx950 tallies a 128-byte
read request as 64 bytes in FETCH_SIZE (16-byte-per-lane streaming reads), so the HBM-side
traffic is 2*FETCH_SIZE + WRITE_SIZE; the runtime's own buffer copy of one 2.147 GB array in
the same kind of run reports FETCH 1.074 GB and WRITE 2.148 GB, which confirms the factor.
"""
import collections
import csv
import sys


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        if "mgx::" not in name:
            continue
        grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])
        acc[(name, grid)].append(float(r["Counter_Value"]))
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
        print("%-40s grid=%9d launches=%3d FETCH_KiB=%12.1f WRITE_KiB=%12.1f corrected=%.3f GB"
              % (short[:40], grid, len(F[key]), f, w, (2 * f + w) * 1024 / 1e9))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
