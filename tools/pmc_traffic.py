"""Per-kernel HBM bytes per launch from two rocprofv3 PMC passes of the bench command:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dirA> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dirB> -- python3 bench.py ...
    python tools/pmc_traffic.py <dirA> <dirB> > profiles/rN_pmc_traffic.json

(MI355X_MICROARCH.md, HBM section: separate passes -- FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2; both in
KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B / lane) coalesced stream, so the raw
figure and the x2-corrected one are both given.)  Dispatches of the LAST optimizer step only."""
import collections
import csv
import glob
import json
import os
import re
import sys


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        raise SystemExit("no counter_collection.csv under " + d)
    rows = list(csv.DictReader(open(f[0])))
    out = collections.OrderedDict()
    for r in rows:
        if r.get("Counter_Name") != counter:
            continue
        key = int(r["Dispatch_Id"])
        out[key] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return out


def label(name):
    m = re.search(r"(conv_fwd128x256h_kernel|conv_fwd256h_kernel|conv_wgrad256h_kernel|conv_fwd256_kernel<\d>|conv_wgrad256_kernel<\d>|"
                  r"conv_fwd_kernel<\d|conv_wgrad_kernel<\d|grad_prep_kernel|act_split_kernel|wgrad_reduce_kernel|"
                  r"pyr_fwd_kernel|pyr_bwd_patch_kernel)", name)
    if not m:
        return None
    s = m.group(1)
    if s.startswith(("conv_fwd_kernel<", "conv_wgrad_kernel<")):
        s += ">"
    return s


def last_step(d):
    """dispatch ids between the last two mask_targets_kernel launches (one per train step)."""
    ids = [k for k, (n, _) in d.items() if "mask_targets_kernel" in n]
    if len(ids) < 2:
        return list(d)
    return [k for k in d if ids[-2] < k <= ids[-1]]


def main(dir_f, dir_w):
    fetch, write = load(dir_f, "FETCH_SIZE"), load(dir_w, "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace --output-format csv "
                     "-- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (MI355X)",
           "units": "KiB per launch (x1024 = bytes); dispatches between the last two mask_targets_kernel launches = "
                    "one steady-state train step; 128x128 kernels sum their tile-width instantiations"}
    agg = {}
    for name, d, key in (("FETCH_SIZE", fetch, 0), ("WRITE_SIZE", write, 1)):
        for k in last_step(d):
            kn, v = d[k]
            lab = label(kn)
            if lab is None:
                continue
            a = agg.setdefault(lab, [0.0, 0, 0.0, 0])
            a[2 * key] += v
            a[2 * key + 1] += 1
    for lab, (f, nf, w, nw) in sorted(agg.items()):
        if not nf or not nw:
            continue
        out[lab] = {"last_step": {
            "launches": nf, "FETCH_SIZE_KiB_per_launch": round(f / nf, 1), "WRITE_SIZE_KiB_per_launch": round(w / nw, 1),
            "read_bytes_per_launch_raw": int(f / nf * 1024),
            "read_bytes_per_launch_x2_gfx950_wide_load_correction": int(2 * f / nf * 1024),
            "write_bytes_per_launch": int(w / nw * 1024)}}
    out["note"] = ("the x2 correction of the guide applies to 1-KiB-per-wave streaming reads (the LDS-DMA pieces of the "
                   "256x256 kernels are that shape); true read traffic lies between the raw and the x2 figure")
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
