#!/usr/bin/env python3
"""gpurun_out/pmc_<precision>_<counter>/x_counter_collection.csv (scripts/pmc_traffic.sh) -> profiles/pmc_traffic.json:
per kernel the mean FETCH_SIZE / WRITE_SIZE per launch, the HBM bytes they stand for -- FETCH_SIZE (KB) is DOUBLED, as
MI355X_MICROARCH.md "HBM" prescribes for gfx950's 16-byte-per-lane reads, WRITE_SIZE (KB) is taken as it is -- and bytes per
point of the profile workload (batch 2, 128x128x64: 2,097,152 points per field launch, 4,194,304 per gather / composite
launch).  bench.py scales these to its own launch sizes for `roofline.traffic`."""
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POINTS = {"void cnerf::field_tile_kernel": 2 * 128 * 128 * 64, "void cnerf::h3::field_h3_kernel": 2 * 128 * 128 * 64,
          "cnerf::gather_kernel": 2 * 128 * 128 * 64, "cnerf::composite_kernel": 2 * 2 * 128 * 128 * 64,
          "cnerf::resample_kernel": 2 * 128 * 128 * 64, "cnerf::merge_composite_kernel": 2 * 2 * 128 * 128 * 64}


def collect(counter, precision):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{precision}_{counter}", "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    out = {"workload": "scripts/profile_workload.py 2 2 (batch 2, 128x128 rays x 64+64 samples, 64^3 volume, SHORTSIREN_FG hidden 256)",
           "method": "rocprofv3 --kernel-trace --pmc <one counter per pass>; FETCH_SIZE x2 (gfx950 wide reads), WRITE_SIZE x1; KB = 1024 B",
           "kernels": {}}
    for prec in ("fp32", "fp16x3", "unfused"):          # unfused: the gather / composite launches (CNERF_WORKLOAD=unfused)
        tabs = {c: collect(c, prec) for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum")}
        for name, fetch in tabs["FETCH_SIZE"].items():
            key = next((k for k in POINTS if name.startswith(k)), None)
            if key is None or name not in tabs["WRITE_SIZE"]:
                continue
            if prec == "unfused" and key not in ("cnerf::gather_kernel", "cnerf::composite_kernel"):
                continue                      # (that run's forward writes its sample points: not the benchmarked forward)
            if prec != "unfused" and "::field_" in key and ((prec == "fp32") != ("field_tile" in key)):
                continue
            if prec == "fp16x3" and "::field_" not in key:
                continue                      # the per-ray kernels do not depend on the precision: keep one copy
            write = tabs["WRITE_SIZE"][name]
            hbm = (2 * fetch + write) * 1024.0
            e = {"fetch_size_kb": fetch, "write_size_kb": write, "hbm_bytes_per_launch": hbm, "points_per_launch": POINTS[key],
                 "bytes_per_point": hbm / POINTS[key]}
            hit, miss = tabs["TCC_HIT_sum"].get(name), tabs["TCC_MISS_sum"].get(name)
            if hit is not None and miss is not None and hit + miss > 0:
                e["l2_hit_rate"] = hit / (hit + miss)
                e["l2_requests"] = hit + miss
            out["kernels"][name] = e
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out["kernels"].items():
        print(f"{k[:70]:70s} fetch {v['fetch_size_kb']:.0f} KB write {v['write_size_kb']:.0f} KB -> {v['hbm_bytes_per_launch']/1e6:.1f} MB "
              f"= {v['bytes_per_point']:.1f} B/point, L2 hit {v.get('l2_hit_rate', float('nan')):.4f}")


if __name__ == "__main__":
    main()
