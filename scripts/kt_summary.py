"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size): calls, mean / median / min / max duration.
The same kernel template serves several configs of one bench run (k_call_v2<64>: C3 and C2) - the grid tells them apart.
usage: kt_summary.py DIR"""
import csv, glob, sys, collections, statistics
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
        wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
        acc[(name, grid, wg, r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel, grid (threads), workgroup, VGPRs, LDS bytes: calls, mean / median / min / max us")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%s grid %s wg %s vgpr %s lds %s: %d calls, mean %.1f median %.1f min %.1f max %.1f us" % (
        k[0], k[1], k[2], k[3], k[4], len(v), statistics.mean(v), statistics.median(v), min(v), max(v)))
