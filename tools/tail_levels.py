"""Duration of the LDS-resident tail launch as a function of its number of levels: plans of
129, 65, 33, 17 and 9 points per side run 50 cycles each (their tails start at 65, 33, ...).
Run under `rocprofv3 --kernel-trace --output-format csv` and read the trace with --read."""
import csv
import sys

N = 50
SIZES = (129, 65, 33, 17, 9)

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "k_tail" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for k, m in enumerate(SIZES):
        d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[k * N:(k + 1) * N])
        if d:
            print("tail from %3d^2: median %.1f us, min %.1f us (%d launches)" % ((m + 1) // 2, d[len(d) // 2] / 1e3, d[0] / 1e3, len(d)))
    sys.exit(0)

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_amd as mg
from multigrid_amd import rhs

for m in SIZES:
    plan = mg.Plan(m, 1.0 / (m - 1), 2)
    plan.set(rhs.random_torch(m), mg.colmajor(m, m))
    plan.cycles(N, 2, 2, 0.6)
    plan.close()
