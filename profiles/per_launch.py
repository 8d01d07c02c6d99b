import csv, collections, sys
by = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "mgx" not in n:
        continue
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_X"]))
    by[(n.replace("void mgx::(anonymous namespace)::", "").replace("(mgx::StreamArgs)", "")[:40], g)].append(
        int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0
for (n, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print("%-42s wg=%6d n=%4d avg=%9.1f us" % (n, g, len(v), sum(v) / len(v) / 1e3))
print("total %.3f ms" % (tot / 1e6))
