#!/usr/bin/env python3
"""Kernel timeline from a rocprofv3 --kernel-trace CSV: start / end / duration / grid / queue / name,
relative to the first kernel of the selected window.

    python tools/timeline.py <..._kernel_trace.csv> [skip_fraction=0.6] [rows=60]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
nshow = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * skip):][:nshow]
t0 = int(rows[0]["Start_Timestamp"])
print("#  start_us    end_us   dur_us      grid queue kernel")
for r in rows:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    print(f"{a:9.1f} {b:9.1f} {b - a:8.1f} {grid:9d} {r.get('Queue_Id', '?'):>5} {r['Kernel_Name'][:60]}")
