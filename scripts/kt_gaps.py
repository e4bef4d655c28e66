"""Timeline of the LAST from-alignments step in a rocprofv3 --kernel-trace CSV: every launch between the last two k_bp_rows_write
starts on one queue (the bench keeps two steps in flight on two streams: the other queue's launches are left out), with its start
offset, duration and the idle gap before it.  usage: kt_gaps.py DIR"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "")))
rows.sort()
marks = [r for r in rows if "k_bp_rows_write" in r[2]]          # (one per step: the sort before it may be one launch or two)
if marks:
    rows = [r for r in rows if r[3] == marks[-1][3]]
seg = [i for i, r in enumerate(rows) if "k_bp_rows_write" in r[2]]
if len(seg) < 2:
    sys.exit("fewer than two steps in the trace")
a, b = seg[-2], seg[-1]
t0, prev_end, busy, idle = rows[a][0], rows[a][0], 0.0, 0.0
print("start us | dur us | gap before us | kernel")
for s, e, n, _ in rows[a:b]:
    gap = (s - prev_end) / 1e3
    print("%9.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    busy += (e - s) / 1e3
    idle += max(0.0, gap)
    prev_end = max(prev_end, e)
print("step span %.1f us (to the next step's first kernel %.1f), kernels %.1f us, idle between kernels %.1f us" % (
    (prev_end - t0) / 1e3, (rows[b][0] - t0) / 1e3, busy, idle))
