#!/usr/bin/env python3
"""Last N kernels of a rocprofv3 --kernel-trace csv: start offset, duration (ms), name.  python scripts/trace_tail.py <dir> [n]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-(int(sys.argv[2]) if len(sys.argv) > 2 else 40):]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.3f %9.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, r["Kernel_Name"][:100]))
