"""Per-kernel matrix-pipe counters from rocprofv3 PMC passes of the bench command (one counter per pass):

    rocprofv3 --pmc MfmaUtil     --kernel-trace --output-format csv -d <dirU> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc MfmaFlopsF16 --kernel-trace --output-format csv -d <dirF> -- python3 bench.py ...
    python tools/pmc_mfma.py <dirU> <dirF> <ms_per_step> > profiles/rN_pmc_mfma.json

MfmaUtil = 100 * sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM) per dispatch;
MfmaFlopsF16 = SQ_INSTS_VALU_MFMA_MOPS_F16 * 512 per dispatch.  Dispatches of the LAST train step only."""
import json
import sys

from pmc_traffic import label, last_step, load


def durations(d):
    """Dispatch_Id -> kernel duration (ns) from the pass's own kernel trace (--kernel-trace next to --pmc)."""
    import csv
    import glob
    import os
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not f:
        return {}
    out = {}
    for r in csv.DictReader(open(f[0])):
        try:
            out[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        except (KeyError, ValueError):
            pass
    return out


def main(dir_u, dir_f, ms_per_step):
    util, flops = load(dir_u, "MfmaUtil"), load(dir_f, "MfmaFlopsF16")
    dur = durations(dir_u)
    wsum = {}            # label -> [sum(util * duration), sum(duration)]
    for k in last_step(util):
        kn, v = util[k]
        lab = label(kn)
        if lab is None or not lab.startswith("conv_") or k not in dur:
            continue
        a = wsum.setdefault(lab, [0.0, 0.0])
        a[0] += v * dur[k]
        a[1] += dur[k]
    out = {"source": "rocprofv3 --pmc MfmaUtil (and, separately, --pmc MfmaFlopsF16) --kernel-trace --output-format csv "
                     "-- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (MI355X); dispatches of one "
                     "steady-state train step (between the last two mask_targets_kernel launches)"}
    agg = {}
    for key, d in ((0, util), (1, flops)):
        for k in last_step(d):
            kn, v = d[k]
            lab = label(kn)
            if lab is None or not lab.startswith("conv_"):
                continue
            a = agg.setdefault(lab, [0.0, 0, 0.0, 0])
            a[2 * key] += v
            a[2 * key + 1] += 1
    total = 0.0
    for lab, (u, nu, f, nf) in sorted(agg.items()):
        out[lab] = {"launches": nu or nf, "MfmaUtil_mean_percent": round(u / nu, 2) if nu else None,
                    # the same counter weighted by each launch's duration (the pass's own kernel trace): what
                    # fraction of the kernel's TIME the matrix pipes were busy -- the unweighted mean counts a
                    # 60-us HBM-bound 1x1 launch like a 5-ms 3x3 one
                    "MfmaUtil_time_weighted_percent": round(wsum[lab][0] / wsum[lab][1], 2)
                    if lab in wsum and wsum[lab][1] > 0 else None,
                    "MfmaFlopsF16_mean_per_launch": f / nf if nf else None, "MfmaFlopsF16_per_step": f}
        total += f
    out["whole_step"] = {"f16_mfma_flops": total, "at_ms_per_step": ms_per_step,
                         "f16_mfma_tflops": round(total / ms_per_step / 1e9, 1),
                         "fraction_of_2500_dense_16bit_peak": round(total / ms_per_step / 1e9 / 2500.0, 4)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]))
