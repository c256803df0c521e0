"""Which layers carry the product path's distance from aten on the reference's five-step loop (VERDICT r5 #2b)?
Replays tests/golden/e2e_multistep_{0,1}.npz (tests/test_multistep_gpu.py's loop) with the strict 3 x bf16 format on
growing prefixes of the backbone, and with the gradient roles' head room varied, next to aten's fp32 convolutions:

    python tools/precision_subsets.py [--scene 0] [--repeats 2] [--json out.json]

prints per policy the worst-step errors (losses, clip norm, backbone slices, other slices) and their ratio to aten's."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# (name, strict pattern, gradient head room or None, {conv_hip switch: value})
POLICIES = [
    ("2 x fp16 everywhere (round-5 default)", "", None),
    ("gradient head room 2^0", "", 0),
    ("strict C1-C2", r"fpn\.C[12]\..*", None),
    ("strict C1-C3", r"fpn\.C[123]\..*", None),
    ("strict C1-C4", r"fpn\.C[1234]\..*", None),
    ("strict backbone C1-C5", r"fpn\.C[12345]\..*", None),
    ("strict everywhere (3 x bf16)", r".*", None),
    # what the steady-state pass adds over the bootstrap pass (step 0 runs none of these), one at a time
    ("2 x fp16, fp32 block outputs (PARTS_ONLY_TRAIN off)", "", None, {"PARTS_ONLY_TRAIN": False}),
    ("2 x fp16, stand-alone gradient preparation (CHAIN_GRAD_PREP off)", "", None, {"CHAIN_GRAD_PREP": False}),
    ("2 x fp16, no shortcut-gradient link", "", None, {"LINK_SHORTCUT_GRAD": False}),
    ("2 x fp16, no parts-only, no chains, no links, no pairs", "", None,
     {"PARTS_ONLY_TRAIN": False, "CHAIN_GRAD_PREP": False, "LINK_SHORTCUT_GRAD": False, "PAIR_STRIDED": False}),
    ("2 x fp16, scales frozen after step 0", "", None, {"_hold": True}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", type=int, default=0)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None, help="comma-separated policy indices")
    args = ap.parse_args()
    os.environ.setdefault("SLN_DEBUG_KNOBS", "1")
    from sln_amodal_amd import conv_hip, nn_ops
    from sln_amodal_amd.config import Config
    from tests._util import golden
    from tests.test_multistep_gpu import _replay_reference_loop
    nn_ops.BACKEND = "hip"
    g = golden("e2e_multistep_%d" % args.scene)
    worst = lambda rows, key: max((r[key][0] if isinstance(r[key], tuple) else r[key]) for r in rows)
    keys = ("dl", "dnorm", "deep", "rest")
    out = {"scene": args.scene, "runs": []}
    aten = []
    for r in range(args.repeats):
        rows, _ = _replay_reference_loop(g, "torch")
        aten.append({k: worst(rows, k) for k in keys})
        print("aten run %d: %s   per-step backbone %s" % (r, {k: "%.2e" % v for k, v in aten[-1].items()},
                                                          ["%.2e" % x["deep"][0] for x in rows]), flush=True)
    out["aten"] = aten
    base = {k: sum(a[k] for a in aten) / len(aten) for k in keys}
    sel = [int(i) for i in args.only.split(",")] if args.only else range(len(POLICIES))
    head0 = conv_hip.GRAD_HEADROOM_LOG2
    for i in sel:
        name, pattern, headroom = POLICIES[i][:3]
        switches = dict(POLICIES[i][3]) if len(POLICIES[i]) > 3 else {}
        Config.STRICT_LAYERS = pattern
        conv_hip.GRAD_HEADROOM_LOG2 = head0 if headroom is None else headroom
        hold = switches.pop("_hold", False)
        saved = {k: getattr(conv_hip, k) for k in switches}
        for k, v in switches.items():
            setattr(conv_hip, k, v)
        real_update = conv_hip.update_scales
        if hold:          # the scales of step 0's bootstrap stay: no delayed-scaling update afterwards
            calls = [0]

            def held(sync=True, _real=real_update, _calls=calls):
                _calls[0] += 1
                if _calls[0] <= 2:
                    return _real(sync)
                conv_hip.flush_wgrad_reduces(True)
                conv_hip.SCALE_EPOCH[0] += 1
                conv_hip._arena_reset()
            conv_hip.update_scales = held
        for r in range(args.repeats):
            sat0 = conv_hip.saturation_count()
            if hold:
                calls[0] = 0
            try:
                rows, opt = _replay_reference_loop(g, "hip")
            except Exception as e:       # a switch combination the path refuses: report, go on
                print("%-40s run %d: FAILED %s" % (name, r, str(e)[:200]), flush=True)
                continue
            w = {k: worst(rows, k) for k in keys}
            rec = {"policy": name, "pattern": pattern, "headroom": conv_hip.GRAD_HEADROOM_LOG2, "run": r, "worst": w,
                   "ratio_to_aten": {k: w[k] / max(base[k], 1e-30) for k in keys},
                   "per_step_backbone": [x["deep"][0] for x in rows], "per_step_rest": [x["rest"][0] for x in rows],
                   "saturated": conv_hip.saturation_count() - sat0, "skipped": opt.skipped_steps()}
            out["runs"].append(rec)
            print("%-40s run %d: worst %s  ratio to aten %s  backbone/step %s  sat %d" % (
                name, r, {k: "%.2e" % v for k, v in w.items()},
                {k: "%.2f" % v for k, v in rec["ratio_to_aten"].items()},
                ["%.2e" % v for v in rec["per_step_backbone"]], rec["saturated"]), flush=True)
        for k, v in saved.items():
            setattr(conv_hip, k, v)
        conv_hip.update_scales = real_update
    Config.STRICT_LAYERS = ""
    conv_hip.GRAD_HEADROOM_LOG2 = head0
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
