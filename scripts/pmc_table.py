#!/usr/bin/env python3
"""gpurun_out/pmck_<workload>_<set>/ (scripts/pmc_kernels.sh) -> markdown table of the hot kernels' counters, per launch (mean).
Derived columns: clock = GRBM_GUI_ACTIVE / 8 / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES) (BUSY counts
cycles, WAVE_CYCLES quad-cycles); co-execution = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES; wait / issue-stall /
active shares of SQ_WAVE_CYCLES; HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (KB, gfx950 correction of the guide)."""
import collections, csv, glob, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"fwd32": ["void cnerf::field_tile_kernel"], "fwd16": ["void cnerf::h3::field_h3_kernel"],
           "pfilm": ["void cnerf::h3::pw::field_pw16_kernel<8, false", "void cnerf::h3::pw::field_pw16_kernel<8, true", "void cnerf::h3::pw::pw_deriv_kernel<8", "void cnerf::pwchain::chain_pre_kernel<8, false",
                     "void cnerf::pwchain::pw_gm_kernel<8, false", "void cnerf::weight_grad16_kernel<8, 8"],
           "bwd16": ["void cnerf::h3::field_h3_kernel<8, 2", "void cnerf::chain16_kernel<8, false", "void cnerf::weight_grad16_kernel<8, 8", "void cnerf::weight_grad16_kernel<1, 8", "cnerf::gather_kernel"]}
vals = collections.defaultdict(lambda: collections.defaultdict(list))      # (workload, kernel) -> counter -> values
durs = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmck_*_*"))):
    if not os.path.isdir(d):
        continue
    w = os.path.basename(d).split("_")[1]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for k in KERNELS.get(w, []):
                if r["Kernel_Name"].startswith(k):
                    vals[(w, k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                        durs[(w, k)].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
m = lambda x: sum(x) / len(x) if x else float("nan")
print("| workload | kernel | ms (under PMC) | clock GHz | MFMA insts | MFMA busy | VALU+MFMA co-exec / MFMA busy | VALU active | wait (barrier, waitcnt) | issue stall | HBM MB (2 x FETCH + WRITE) | L2 hit |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for (w, k), c in vals.items():
    g = {n: m(v) for n, v in c.items()}
    ms = m(durs[(w, k)])
    wc = g.get("SQ_WAVE_CYCLES", float("nan"))
    busy = g.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan"))
    hit, miss = g.get("TCC_HIT_sum", float("nan")), g.get("TCC_MISS_sum", float("nan"))
    print(f"| {w} | `{k.replace('void cnerf::', '')}...` | {ms:.2f} | {g.get('GRBM_GUI_ACTIVE', float('nan')) / 8 / (ms * 1e6):.2f} | {g.get('SQ_INSTS_MFMA', float('nan')):.3e} | "
          f"{busy / (4 * wc):.3f} | {g.get('SQ_VALU_MFMA_COEXEC_CYCLES', float('nan')) / busy:.3f} | {g.get('SQ_ACTIVE_INST_VALU', float('nan')) / wc:.3f} | "
          f"{g.get('SQ_WAIT_ANY', float('nan')) / wc:.3f} | {g.get('SQ_WAIT_INST_ANY', float('nan')) / wc:.3f} | "
          f"{(2 * g.get('FETCH_SIZE', float('nan')) + g.get('WRITE_SIZE', float('nan'))) * 1024 / 1e6:.1f} | {hit / (hit + miss):.4f} |")
