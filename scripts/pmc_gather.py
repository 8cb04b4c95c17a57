#!/usr/bin/env python3
"""gpurun_out/r3/pmcg_<mode>_<set>/**/x_counter_collection.csv (scripts/pmc_gather.sh) -> a table of the lookup kernel's counters per
launch (2,097,152 points), linear vs patch visiting order; kernel time from the kernel trace of the same runs."""
import collections
import csv
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for mode in (0, 1):
    vals, times = collections.defaultdict(list), []
    for d in glob.glob(os.path.join(ROOT, "gpurun_out", "r3", f"pmcg_{mode}_*")):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = collections.defaultdict(float)          # (dispatch, counter) -> sum over instances / dimensions
            for r in csv.DictReader(open(f)):
                if "gather" in r["Kernel_Name"]:
                    per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            for (_, c), v in per.items():
                vals[c].append(v)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gather" in r["Kernel_Name"]:
                    times.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    out[mode] = ({c: sum(v) / len(v) for c, v in vals.items()}, sum(times) / max(len(times), 1))
names = sorted(set(out[0][0]) | set(out[1][0]))
print("| counter (sum over instances, per launch of 2,097,152 lookups) | gather_kernel (point by point, ray order) | gather_box_kernel (4x4x2 patches, distinct lines once via LDS) |")
print("|---|---|---|")
print(f"| kernel ms (under PMC, mean over all passes) | {out[0][1]:.4f} | {out[1][1]:.4f} |")
for n in names:
    print(f"| {n} | {out[0][0].get(n, float('nan')):.4g} | {out[1][0].get(n, float('nan')):.4g} |")
for mode in (0, 1):
    v = out[mode][0]
    if "TCP_PERF_SEL_TOTAL_READ" in v and "TCP_TCC_READ_REQ" in v:
        print(f"\nmode {mode}: vector-L1 read requests {v['TCP_PERF_SEL_TOTAL_READ']:.4g}, of which forwarded to L2 {v['TCP_TCC_READ_REQ']:.4g} "
              f"-> L1 hit rate {1 - v['TCP_TCC_READ_REQ'] / max(v['TCP_PERF_SEL_TOTAL_READ'], 1):.3f}")
    if "TCC_HIT" in v and "TCC_MISS" in v:
        print(f"mode {mode}: L2 hit rate {v['TCC_HIT'] / max(v['TCC_HIT'] + v['TCC_MISS'], 1):.3f}")
