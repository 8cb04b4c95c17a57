#!/usr/bin/env python3
"""Kernels of ONE render forward + backward step in launch order with their gaps (rocprofv3 --kernel-trace csv of scripts/time_backward.py):
python scripts/step_trace.py <dir>  -> the last step's kernels, idle gaps > 50 us flagged"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last step: walk back from the end to the previous merge_composite_kernel (forward's last kernel of the step before)
idx = [i for i, r in enumerate(rows) if "merge_composite_kernel" in r["Kernel_Name"]]
start = idx[-2] + 1 if len(idx) >= 2 else 0
# back up to the start of that forward: first kernel after the previous backward
rows = rows[start:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0; busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    busy += e - s
    flag = "  <-- idle %.0f us" % gap if gap > 50 else ""
    if (e - s) > 20000 or gap > 50:
        print("%9.3f %9.3f  %s%s" % ((s - t0) / 1e6, (e - s) / 1e6, r["Kernel_Name"][:90], flag))
    prev_end = max(prev_end, e)
print("span %.2f ms, busy %.2f ms, kernels %d" % ((prev_end - t0) / 1e6, busy / 1e6, len(rows)))
