"""Bisect of the fixed-batch learning run (tests/test_zz_dynamics_gpu.py) over the A/B switches.

  python3 tools/bisect_learning.py            # parent: one child process per switch setting
  python3 tools/bisect_learning.py --child    # one run in this process' environment

Each child prints the six losses at steps 0/20/40/79, the clip norm, the skipped-step counter and
the fp16 saturation counter, so a regression names its loss and its step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SWITCHES = [
    {},
    {"SLN_CONV_PARTS": "3"},
    {"SLN_CHAIN_FPN_OUTPUTS": "0"},
    {"SLN_CHAIN_FPN_LATERAL": "0"},
    {"SLN_CHAIN_DECONV": "0"},
    {"SLN_STEM_POOL_HANDOFF": "0"},
    {"SLN_FUSE_CROP_GRADS": "0"},
    {"SLN_BATCH_WGRAD_REDUCE": "0"},
    {"SLN_SUMS_ARENA": "0"},
    {"SLN_PARTS_ONLY_TRAIN": "0"},
    {"SLN_LINK_SHORTCUT_GRAD": "0"},
    {"SLN_CHAIN_TWO_READERS": "0"},
    {"SLN_CHAIN_GRAD_PREP": "0"},
    {"SLN_CHAIN_BLOCK_OUTPUT": "0"},
    {"SLN_BACKEND": "torch"},
]


def child(steps=80, seed=3, report=(0, 20, 40, 79)):
    import torch
    from sln_amodal_amd import conv_hip, nn_ops, synthetic
    from sln_amodal_amd.config import Config
    from sln_amodal_amd.model import MaskRCNN
    if os.environ.get("SLN_BACKEND"):
        nn_ops.BACKEND = os.environ["SLN_BACKEND"]

    class C(Config):
        NAME = "t"
        IMAGE_MAX_DIM = 256
        ARCHITECTURE = "resnet50"

    torch.manual_seed(0)
    cfg = C()
    m = MaskRCNN(cfg, "/tmp/sln_logs").apply_amodal_heads().cuda()
    m.set_trainable(".*", exclusive_off=False)
    for p in m.GLM_modual.parameters():
        p.requires_grad = False
    batch = synthetic.make_batch(cfg, 2, 256, 256, seed=seed, anchors_f64=m.anchors_f64)
    synthetic.calibrate_batchnorm(m, batch["images"])
    synthetic.calibrate_glm(m, batch["images"])
    synthetic.warm_start_rpn(m, [batch], iters=40)
    opt = m.make_optimizer(float(os.environ.get("SLN_BISECT_LR", "0.01")))
    gen = torch.Generator(device="cuda").manual_seed(5)
    pr = {"pos": torch.rand(2, 1000, device="cuda", generator=gen),
          "neg": torch.rand(2, 1000, device="cuda", generator=gen)}
    rows = []
    for it in range(steps):
        loss, parts = m.train_step(batch, opt, priorities=pr)
        if it in report or it == steps - 1:
            row = {"step": it, "total": round(float(loss), 4)}
            row.update({k: round(float(v), 4) for k, v in parts.items()})
            row["norm"] = round(float(m.last_grad_norm), 3) if m.last_grad_norm is not None else None
            rows.append(row)
    skipped = opt.skipped_steps() if hasattr(opt, "skipped_steps") else None
    sat = conv_hip.saturation_count() if nn_ops.BACKEND != "torch" else None
    print(json.dumps({"rows": rows, "skipped": skipped, "saturated": sat}))


def main():
    if "--child" in sys.argv:
        child()
        return
    # arguments: "A=1,B=0" = one run with those variables set ("-" = defaults); none = the switch table
    only = [a for a in sys.argv[1:] if a != "--child"]
    sw = [dict(kv.split("=") for kv in a.split(",") if "=" in kv) for a in only] if only else SWITCHES
    for s in sw:
        env = dict(os.environ, **s)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env,
                           capture_output=True, text=True, timeout=900)
        line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
        print("=== %s rc=%d" % (s or "default", p.returncode), flush=True)
        try:
            r = json.loads(line)
            for row in r["rows"]:
                print("   ", row)
            print("    skipped=%s saturated=%s  d_total=%.3f" % (
                r["skipped"], r["saturated"], r["rows"][-1]["total"] - r["rows"][0]["total"]), flush=True)
        except Exception:
            print(p.stdout[-1500:], p.stderr[-3000:], flush=True)


if __name__ == "__main__":
    main()
