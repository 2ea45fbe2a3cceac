"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import csv, glob, sys, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in d.items()})
