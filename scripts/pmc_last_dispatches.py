"""The LAST n dispatches of one kernel in rocprofv3 --pmc CSV output, counter by counter (dev tool: scripts/r05_place_probe.py ends
with a walk into its fastest and one into its slowest allocation).  usage: pmc_last_dispatches.py KERNEL_SUBSTRING N DIR [DIR ...]"""
import csv, glob, sys, collections
name, n = sys.argv[1], int(sys.argv[2])
for root in sys.argv[3:]:
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ids = sorted(per)[-n:]
        for c in sorted(per[ids[0]]) if ids else []:
            print("%-44s %s" % (c, "  ".join("%14.6g" % per[i].get(c, float("nan")) for i in ids)))
