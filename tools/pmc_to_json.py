#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name -> JSON {kernel: {COUNTER: {launches, sum}}}.
Kernel names are normalised the way bench.py's roofline names them (template arguments kept, signature dropped)."""
import csv
import glob
import json
import re
import sys


def norm(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"^(void )?([\w:]+(<[^(]*>)?)", name)
    return ((m.group(1) or "") + m.group(2)) if m else name


out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = norm(r["Kernel_Name"])
            c = out.setdefault(k, {}).setdefault(r["Counter_Name"], {"launches": 0, "sum": 0.0})
            c["launches"] += 1
            c["sum"] += float(r["Counter_Value"])
out = {k: v for k, v in out.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
json.dump(out, sys.stdout, indent=1)
