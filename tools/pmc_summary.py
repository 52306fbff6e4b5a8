#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch, per kernel.
usage: pmc_summary.py <dir> [kernel-substring ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    want = sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void fsa::", "").replace("fsa::", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        if want and not any(w in k for w in want):
            continue
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"    {c:28s} {sum(v) / len(v):16.1f}  (n={len(v)})")


if __name__ == "__main__":
    main()
