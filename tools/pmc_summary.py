#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel.  usage: pmc_summary.py <dir with pass subdirs>"""
import csv
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in sorted(root.rglob("*counter_collection.csv")):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"== {k}")
    for c in sorted(tot[k]):
        print(f"   {c:28s} {tot[k][c]:18.0f}   dispatches {calls[k][c]}")
