#!/usr/bin/env python3
"""Per-launch means of rocprofv3 PMC counters for one kernel.

    python tools/pmc_summarize.py <dir with *_counter_collection.csv (searched recursively)> \
        [kernel substring = nmpc_solve_kernel] [instances per launch = 4096] [warm-up launches to drop = 0] \
        > profiles/<round>/pmc_summary.json

A *_kernel_trace.csv found beside them gives `kernel_ms` (mean / median / min over the same launches).

Each `--pmc` pass is its own run (rocprofv3 does not split counters over passes); point this at
the directory that holds all of them.  Values are summed over the dimensions rocprofv3 lists per
dispatch (XCC / SE / instance rows) and averaged over the launches of the kernel.
"""
import csv
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    kern = sys.argv[2] if len(sys.argv) > 2 else "nmpc_solve_kernel"
    per_launch = float(sys.argv[3]) if len(sys.argv) > 3 else 4096.0
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    out = {}
    for dp, _, files in os.walk(root):
        for f in files:
            if not f.endswith("counter_collection.csv"):
                continue
            vals = defaultdict(lambda: defaultdict(float))   # counter -> dispatch -> sum
            with open(os.path.join(dp, f), newline="") as fh:
                for row in csv.DictReader(fh):
                    if kern not in row["Kernel_Name"]:
                        continue
                    vals[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            for name, d in vals.items():
                keep = [d[k] for k in sorted(d, key=int)][skip:]        # dispatch order; warm-ups dropped
                m = sum(keep) / len(keep)
                out[name] = {"launches": len(keep), "mean_per_launch": m, "per_solve": m / per_launch}
    for dp, _, files in os.walk(root):
        for f in files:
            if not f.endswith("kernel_trace.csv"):
                continue
            rows = []
            with open(os.path.join(dp, f), newline="") as fh:
                for row in csv.DictReader(fh):
                    if kern in row["Kernel_Name"]:
                        rows.append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6))
            ms = [m for _, m in sorted(rows)][skip:]
            if ms:
                srt = sorted(ms)
                out["kernel_ms"] = {"launches": len(ms), "mean": sum(ms) / len(ms), "median": srt[len(srt) // 2],
                                    "min": srt[0], "max": srt[-1]}
    # (rocprofv3's per-dispatch metadata -- VGPR_Count, LDS_Block_Size ... -- is not reported: it disagrees with the
    # code object on this image; registers, scratch and LDS are in profiles/<round>/kernel_resources.txt)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
