#!/usr/bin/env python3
"""Average the counter values rocprofv3 wrote (csv, one row per dispatch and counter) per kernel name.
usage: pmc_average.py <dir with p*/.../*counter_collection.csv> <kernel name substring>"""
import csv, glob, json, os, re, sys
src, flt = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if flt not in name:
                continue
            short = re.sub(r"\(.*", "", re.sub(r"^void ", "", name))
            a = acc.setdefault(short, {}).setdefault(row["Counter_Name"], [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
print(json.dumps({k: {c: {"launches": v[0], "mean": v[1] / v[0]} for c, v in d.items()} for k, d in acc.items()}, indent=1))
