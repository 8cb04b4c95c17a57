#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run:  python scripts/kernel_stats_top.py <dir> [n]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in list(csv.DictReader(open(f)))[:n]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s} total {float(r["TotalDurationNs"])/1e6:9.2f} ms avg {float(r["AverageNs"])/1e6:8.3f} ms  {r["Percentage"]}%')
