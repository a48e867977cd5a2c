#!/usr/bin/env python3
"""profiles/rNN/single_gpu_sizes.json from a single-GPU bench line: the per-term times at 2^20 ... 2^24 rows that a
multi-GPU line's `scaling_prediction` is built from when the run itself cannot measure them (bench.py: STATIC_SIZES).

    python tools/make_single_gpu_sizes.py gpurun_out/r4/bench_line.json profiles/r04/single_gpu_sizes.json
"""
import json
import sys

line = json.load(open(sys.argv[1]))
sp = line["scaling_prediction"]
out = {"source": f"{sys.argv[1]}: python bench.py on one MI355X (value {line['value']:.1f} {line['unit']})",
       "value_1gpu": line["value"], "us_per_term_by_log2_rows": sp["us_per_term_by_log2_rows"],
       # what a rank of the row-partitioned step achieves (terms one by one): the one-term walk's times
       "us_per_term_one_term_walk_by_log2_rows": sp.get("us_per_term_one_term_walk_by_log2_rows")}
with open(sys.argv[2], "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")
print(out)
