#!/usr/bin/env python3
"""Per-kernel matrix-pipe utilisation from the committed counter passes (tools/collect_profiles.sh).

  MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD)
    = rocprofiler-sdk's MfmaUtil for gfx950 (counter_defs.yaml: reduce(SQ_VALU_MFMA_BUSY_CYCLES,sum) /
    (reduce(GRBM_GUI_ACTIVE,max) * SIMD_NUM)).  SQ_VALU_MFMA_BUSY_CYCLES sums, over the chip's 256 CUs x 4 SIMDs, the
    cycles a SIMD's matrix pipe is busy (16 per v_mfma_f32_16x16x32_f16: MI355X_MICROARCH.md per-instruction constants);
    GRBM_GUI_ACTIVE has one instance per XCD and the CSV carries their SUM, so the per-XCD value (what `max` picks, the
    instances agree to < 1 %) is sum / 8.
  effective clock   = GRBM_GUI_ACTIVE per XCD / dispatch duration (the chip clocks to its power budget: DVFS give-back;
    counter passes serialise the dispatches, so the chip runs cooler and faster here than in the un-profiled bench)
  Dispatches shorter than 30 us: GRBM_GUI_ACTIVE's window (command-processor set-up + drain) is then a large part of the
    count, so gui / duration is no clock (it read 2.6 - 4 GHz on a 2.4 GHz part in round 3) and busy / gui under-states the
    kernel: the clock column prints n/a and the busy fraction is marked `>=` (a lower bound) for those rows.
  HBM MB / launch   = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE counts half of wide coalesced reads)
Usage: mfma_report.py mfma_counters.json kernel_stats.csv [hbm_counters.json]"""
import csv
import json
import re
import sys


def norm(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"^(void )?([\w:]+(<[^(]*>)?)", name)
    return ((m.group(1) or "") + m.group(2)) if m else name


mf = json.load(open(sys.argv[1]))
dur = {norm(r["Name"]): (float(r["AverageNs"]), int(r["Calls"]), float(r["Percentage"])) for r in csv.DictReader(open(sys.argv[2]))}
hbm = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else {}
print(f"{'kernel':58s} {'% time':>6s} {'avg us':>8s} {'MFMA busy':>9s} {'clock GHz':>9s} {'HBM MB/launch':>13s}")
rows = []
for k, v in mf.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in v or "GRBM_GUI_ACTIVE" not in v:
        continue
    n = v["GRBM_GUI_ACTIVE"]["launches"]
    gui = v["GRBM_GUI_ACTIVE"]["sum"] / n / 8.0          # per XCD
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / v["SQ_VALU_MFMA_BUSY_CYCLES"]["launches"]
    sqb = v.get("SQ_BUSY_CYCLES", {"sum": 0, "launches": 1})
    sqb = sqb["sum"] / max(1, sqb["launches"])
    ns = v["_ns"]["sum"] / v["_ns"]["launches"] if "_ns" in v else None
    avg_ns, calls, pct = dur.get(k, (None, 0, 0.0))
    hb = hbm.get(k)
    mb = None
    if hb and "FETCH_SIZE" in hb and "WRITE_SIZE" in hb:
        mb = (2 * hb["FETCH_SIZE"]["sum"] / hb["FETCH_SIZE"]["launches"] + hb["WRITE_SIZE"]["sum"] / hb["WRITE_SIZE"]["launches"]) * 1024 / 1e6
    rows.append((pct, k, avg_ns, busy / (1024.0 * gui) if gui else 0.0, sqb / gui if gui else 0.0, gui / ns if ns else None, mb))
SHORT_NS = 30e3
for pct, k, avg_ns, util, sq, clk, mb in sorted(rows, reverse=True)[:24]:
    short = (avg_ns or 0) < SHORT_NS or (clk is not None and clk > 2.45)      # (2.4 GHz is the part's ceiling)
    us = f"{(avg_ns or 0) / 1e3:8.1f}"
    ut = f"{'>=' if short else '  '}{util:7.3f}"
    ck = "      n/a" if short else (f"{clk:9.2f}" if clk else "        -")
    print(f"{k[:58]:58s} {pct:6.2f} {us} {ut} {ck} {(f'{mb:13.1f}' if mb is not None else '            -')}")
