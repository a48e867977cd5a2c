"""Turn the rocprofv3 output of tools/profile.sh (gpurun_out/prof_<tag>/) into the two files
kept under profiles/: the kernel-stats CSV as rocprofv3 wrote it and a JSON summary of the
PMC passes for the dominant kernel (per-launch means; FETCH_SIZE with the gfx950 x2
correction prescribed by MI355X_MICROARCH.md, WRITE_SIZE as is; units KiB -> bytes).

usage: python tools/summarize_pmc.py <tag> <out-prefix> [kernel-substring]"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path, kernel):
    out = {}
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel in row["Kernel_Name"]:
                    out.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return out


def main():
    tag, prefix = sys.argv[1], sys.argv[2]
    kernel = sys.argv[3] if len(sys.argv) > 3 else "hrb_spmv_kernel<qp::ChebyOp"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], prefix + "_kernel_stats.csv")
    vals = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_l2"):
        vals.update(counters(os.path.join(src, sub), kernel))
    summ = {"command": "rocprofv3 --pmc <counter> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --cpu-steps 0 "
                       "(one pass per counter group, tools/profile.sh)",
            "kernel": kernel, "counters": {}}
    for k, v in vals.items():
        summ["counters"][k] = {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    fetch = summ["counters"]["FETCH_SIZE"]["mean"] * 1024.0
    write = summ["counters"]["WRITE_SIZE"]["mean"] * 1024.0
    summ["FETCH_SIZE_bytes_raw"] = fetch
    summ["FETCH_SIZE_bytes_corrected_x2"] = 2 * fetch
    summ["WRITE_SIZE_bytes"] = write
    summ["hbm_traffic_bytes_per_launch"] = 2 * fetch + write
    if "TCC_HIT_sum" in summ["counters"]:
        h, m = summ["counters"]["TCC_HIT_sum"]["mean"], summ["counters"]["TCC_MISS_sum"]["mean"]
        summ["l2_hit_rate"] = h / (h + m)
    summ["note"] = ("gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams "
                    "(MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact. Units KiB.")
    with open(prefix + "_pmc_summary.json", "w") as fh:
        json.dump(summ, fh, indent=1)
    print(json.dumps({k: summ[k] for k in ("hbm_traffic_bytes_per_launch", "FETCH_SIZE_bytes_corrected_x2",
                                           "WRITE_SIZE_bytes")}, indent=1))


if __name__ == "__main__":
    main()
