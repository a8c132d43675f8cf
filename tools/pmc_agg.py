#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name -> JSON
{kernel: {COUNTER: {launches, sum}, "_ns": {launches, sum}}} (dispatch durations when the CSV carries timestamps).
Kernel names are normalised the way bench.py's roofline names them (template arguments kept, signature dropped).
Usage: pmc_agg.py DIR [DIR ...] > out.json"""
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
        seen = set()
        for r in csv.DictReader(open(f)):
            k = norm(r["Kernel_Name"])
            c = out.setdefault(k, {}).setdefault(r["Counter_Name"], {"launches": 0, "sum": 0.0})
            c["launches"] += 1
            c["sum"] += float(r["Counter_Value"])
            did = (f, r.get("Dispatch_Id"))
            if did not in seen and r.get("Start_Timestamp") and r.get("End_Timestamp"):
                seen.add(did)
                t = out[k].setdefault("_ns", {"launches": 0, "sum": 0.0})
                t["launches"] += 1
                t["sum"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
json.dump(out, sys.stdout, indent=1)
