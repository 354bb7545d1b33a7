"""Average PMC counter values per dispatch of the kernels whose name starts with a prefix, from a rocprofv3 --pmc output directory.
usage: pmc_kernel.py DIR PREFIX"""
import csv, glob, sys
from collections import defaultdict
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0])))
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for r in rows:
    if r["Kernel_Name"].startswith(sys.argv[2]):
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]].add(r["Dispatch_Id"])
for k in acc:
    print(k[:50], "dispatches", len(n[k]), {c: round(v / len(n[k])) for c, v in acc[k].items()})
