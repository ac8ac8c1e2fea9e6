#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection / kernel_trace CSVs: per kernel name, mean counter value
per dispatch (counters) or mean duration (trace).  usage: pmc_summary.py DIR [name-substring[|name-substring...]]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    subs = (sys.argv[2] if len(sys.argv) > 2 else "").split("|")
    want = lambda k: any(s in k for s in subs)
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        with open(path) as f:
            for r in csv.DictReader(f):
                k = r.get("Kernel_Name", "")
                if want(k):
                    acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(os.path.relpath(path, d), "|", k)
            for c, v in sorted(cs.items()):
                print("    %-28s n=%-4d mean=%.6g" % (c, len(v), sum(v) / len(v)))
    for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        acc = defaultdict(list)
        with open(path) as f:
            for r in csv.DictReader(f):
                k = r.get("Kernel_Name", "")
                if want(k):
                    acc[k[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in acc.items():
            print(os.path.relpath(path, d), "|", k, "| n=%d mean=%.1f us min=%.1f us" % (len(v), sum(v) / len(v), min(v)))


if __name__ == "__main__":
    main()
