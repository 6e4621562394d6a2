#!/usr/bin/env python3
"""Per-kernel means of the counters in a rocprofv3 --pmc output directory.   python tools/summarize_counters.py DIR [DIR ...]"""
import collections, csv, re, sys
from pathlib import Path
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in Path(d).rglob("*_counter_collection.csv"):
        for row in csv.DictReader(f.open()):
            m = re.search(r"(k_\w+(?:<[^>]*>)?)", row["Kernel_Name"])
            acc[m.group(1) if m else row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} {sum(v) / len(v):16.0f}   ({len(v)} launches)")
