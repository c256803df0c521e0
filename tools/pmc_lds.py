"""Per-kernel LDS bank-conflict share from two rocprofv3 PMC passes of the bench command (one counter per pass):

    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d <dirC> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strict
    rocprofv3 --pmc SQ_LDS_IDX_ACTIVE    --kernel-trace --output-format csv -d <dirA> -- python3 bench.py ...
    python tools/pmc_lds.py <dirC> <dirA> > profiles/rN_pmc_lds.json

SQ_LDS_BANK_CONFLICT = LDS-array cycles added by bank conflicts, SQ_LDS_IDX_ACTIVE = all LDS-array cycles of indexed
operations (MI355X_MICROARCH.md, LDS section).  Dispatches of the LAST train step only."""
import json
import sys

from pmc_traffic import last_step, load


def main(dir_c, dir_a):
    conf, act = load(dir_c, "SQ_LDS_BANK_CONFLICT"), load(dir_a, "SQ_LDS_IDX_ACTIVE")
    agg = {}
    for key, d in ((0, conf), (1, act)):
        for k in last_step(d):
            kn, v = d[k]
            name = kn.split("(")[0][:90]
            a = agg.setdefault(name, [0.0, 0, 0.0, 0])
            a[2 * key] += v
            a[2 * key + 1] += 1
    out = {"source": "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT (and, separately, --pmc SQ_LDS_IDX_ACTIVE) --kernel-trace -- python3 "
                     "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strict; dispatches of one steady-state train step"}
    rows = []
    for name, (c, nc, a, na) in agg.items():
        if not na or a <= 0:
            continue
        rows.append((a, name, {"launches": na, "lds_active_cycles_per_launch": round(a / na), "bank_conflict_cycles_per_launch":
                               round(c / max(nc, 1)), "conflict_share_of_active": round((c / max(nc, 1)) / (a / na), 3)}))
    for a, name, v in sorted(rows, reverse=True)[:24]:
        out[name] = v
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
