#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters from *_counter_collection.csv.
Usage: tools/pmc_summary.py counter_collection.csv [kernel-substring]"""
import collections
import csv
import re
import sys


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    first_ctr = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if sub and sub not in k:
                continue
            m = re.search(r"(\w+)(<[^>]*>)?\(", k.replace("(anonymous namespace)::", ""))
            k = (m.group(1) + (m.group(2) or "")) if m else k[:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
            # duration of the dispatch in this (profiled) pass, where the CSV carries timestamps: one row per counter, so
            # average over rows of the first counter only
            if "Start_Timestamp" in row and "End_Timestamp" in row:
                first = first_ctr.setdefault(k, row["Counter_Name"])
                if row["Counter_Name"] == first:
                    try:
                        acc[k]["_kernel_avg_us"] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3
                        cnt[k]["_kernel_avg_us"] += 1
                    except ValueError:
                        pass
    for k in acc:
        print(k)
        for c in sorted(acc[k]):
            print("   %-28s mean/dispatch %.6g   (n=%d)" % (c, acc[k][c] / cnt[k][c], cnt[k][c]))


if __name__ == "__main__":
    main()
