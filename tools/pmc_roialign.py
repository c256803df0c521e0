"""Measured HBM-side bytes of the RoIAlign / label-decode kernels next to their algorithmic figures
(tools/roialign_bench.py under two rocprofv3 PMC passes, FETCH_SIZE and WRITE_SIZE):

    python tools/pmc_roialign.py <dir FETCH_SIZE> <dir WRITE_SIZE> > profiles/rN_pmc_roialign.json

The algorithmic model charges 20 B per forward element (4 taps x 4 B + 4 B store) and 36 B per
backward element (4 B load + 4 x 8 B atomic read-modify-write); the four maps total 1.43 GB, far less than
what the model charges for 105 M elements, so most of that traffic is served by L2 / Infinity Cache --
this file shows how much reaches the fabric-side counters (KiB per launch; FETCH_SIZE x2 on gfx950 for wide
coalesced streams, MI355X_MICROARCH.md)."""
import json
import sys

from pmc_traffic import load


def main(dir_f, dir_w):
    fetch, write = load(dir_f, "FETCH_SIZE"), load(dir_w, "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace -- python3 "
                     "tools/roialign_bench.py: 16 images, 1600 rois, 256 channels, P2..P5 of a 1024^2 image",
           "units": "bytes per launch; mean over the launches of the run"}
    agg = {}
    for key, d in ((0, fetch), (1, write)):
        for k, (name, v) in d.items():
            for lab in ("pyr_fwd_kernel", "pyr_bwd_patch_kernel", "pyr_bwd_kernel", "fillBuffer", "label_decode_kernel"):
                if lab in name:
                    a = agg.setdefault(lab, [0.0, 0, 0.0, 0])
                    a[2 * key] += v * 1024
                    a[2 * key + 1] += 1
    for lab, (f, nf, w, nw) in sorted(agg.items()):
        out[lab] = {"launches": nf, "fetch_bytes_per_launch_raw": int(f / max(nf, 1)),
                    "fetch_bytes_per_launch_x2": int(2 * f / max(nf, 1)),
                    "write_bytes_per_launch": int(w / max(nw, 1))}
    K, C = 1600, 256
    out["algorithmic"] = {"pool16_fwd_bytes": K * 16 * 16 * C * 20, "pool16_bwd_bytes": K * 16 * 16 * C * 36,
                          "pool7_fwd_bytes": K * 7 * 7 * C * 20, "pool7_bwd_bytes": K * 7 * 7 * C * 36,
                          "four_maps_bytes": 16 * 256 * 4 * (256 ** 2 + 128 ** 2 + 64 ** 2 + 32 ** 2),
                          "note": "launches mix the pool-16 (mask head) and pool-7 (classifier) crops of the bench"}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
