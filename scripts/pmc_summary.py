"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


if __name__ == "__main__":
    for k, d in load(sys.argv[1]).items():
        print(k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in d.items()})
