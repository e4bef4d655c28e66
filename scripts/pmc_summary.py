"""Summarise rocprofv3 --pmc CSV output: per kernel, mean counter value per dispatch (dev tool).
usage: pmc_summary.py DIR [DIR ...]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for root in sys.argv[1:]:
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s mean %.4g over %d dispatches (min %.4g, max %.4g)" % (c, sum(v) / len(v), len(v), min(v), max(v)))
