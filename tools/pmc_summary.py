#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel mean of every counter.
usage: pmc_summary.py DIR [--all]   (default: kernels whose name contains k_trunk; --all: every oth:: kernel)"""
import collections
import csv
import glob
import sys

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
every = "--all" in sys.argv
for d in dirs:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "k_trunk" in name or (every and "oth::" in name):
                short = name.split("(")[0].replace("void ", "")
                agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(agg):
            print(d, k, {c: "%.5g (n=%d)" % (sum(v) / len(v), len(v)) for c, v in sorted(agg[k].items())})
