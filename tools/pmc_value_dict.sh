#!/bin/bash
# HBM traffic per launch of the row-block mat-vec with and without the value-dictionary mirror: one rocprofv3 --pmc pass per counter
# (the guide's recipe: FETCH_SIZE x 2 on gfx950, WRITE_SIZE as is, KiB), + a --stats pass for the times.  Output: gpurun_out/pmc_value_dict.txt
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_value_dict.txt
mkdir -p $REPO/gpurun_out
: > $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pv_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pv_$c -- python3 $REPO/tools/pmc_value_dict.py > /tmp/pv_$c.log 2>&1
done
rm -rf /tmp/pv_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_stats -- python3 $REPO/tools/pmc_value_dict.py > /tmp/pv_stats.log 2>&1
python3 - >> $OUT <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pv_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "rbcsr" in k and row["Counter_Name"] == c:
                tot[k.split("(")[0][:70]][c].append(float(row["Counter_Value"]))
print("# per-launch means; launches in dispatch order: tfim-20 coded, tfim-20 plain, xxz-20 coded, xxz-20 plain (62 launches each)")
for k, d in tot.items():
    f, w = d["FETCH_SIZE"], d["WRITE_SIZE"]
    half = len(f) // 2
    for name, sl in (("tfim-20", slice(0, half)), ("xxz-20", slice(half, None))):
        ff, ww = f[sl], w[sl]
        if ff:
            fb, wb = 2 * 1024 * sum(ff) / len(ff), 1024 * sum(ww) / max(len(ww), 1)
            print(f"{k:72s} {name:8s} launches {len(ff):4d}  FETCH x2 {fb / 1e6:8.1f} MB  WRITE {wb / 1e6:7.1f} MB  total {(fb + wb) / 1e6:8.1f} MB = {(fb + wb) / 2**20:6.1f} B/row")
for f in glob.glob("/tmp/pv_stats/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rbcsr" in row["Name"]:
            print(f"stats: {row['Name'].split('(')[0][:70]:72s} calls {row['Calls']:>5s} avg {float(row['AverageNs']) / 1e3:8.2f} us (both chains)")
PY
cat $OUT
