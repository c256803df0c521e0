"""Run-to-run spread of tests/test_zz_dynamics_gpu.py's lr-0.01 scenario: R repetitions of the same 80 steps from
the same weights on the product path (and a few on aten), every step's total and mask loss recorded.  Prints, per
run, the statistics a robust assertion could use (last value, mean / median of the last 10, minimum after step 40)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SLN_DEBUG_KNOBS", "1")
import numpy as np
import torch
from sln_amodal_amd import nn_ops
from tests.test_zz_dynamics_gpu import _prepared

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
RA = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nn_ops.BACKEND = "hip"
m, cfg, batch, pr = _prepared()
start = {k: v.detach().clone() for k, v in m.state_dict().items()}


def run(backend):
    m.load_state_dict(start)
    nn_ops.BACKEND = backend
    opt = m.make_optimizer(0.01)
    tot, lay, cls = [], [], []
    for it in range(80):
        loss, parts = m.train_step(batch, opt, priorities=pr)
        tot.append(float(loss)); lay.append(float(parts["layer"])); cls.append(float(parts["mrcnn_class"]))
    nn_ops.BACKEND = "hip"
    return np.array(tot), np.array(lay), np.array(cls)


out = []
for backend, n in (("hip", R), ("torch", RA)):
    for r in range(n):
        t, l, c = run(backend)
        row = dict(backend=backend, run=r, t0=t[0], t79=t[79], mean10=t[-10:].mean(), med10=float(np.median(t[-10:])),
                   min40=t[40:].min(), max_t=t.max(), l0=l[0], l79=l[79], lmean10=l[-10:].mean(), cls79=c[79],
                   clsmax60=c[60:].max())
        print({k: (round(float(v), 4) if not isinstance(v, (str, int)) else v) for k, v in row.items()}, flush=True)
        out.append(dict(backend=backend, run=r, total=t.round(4).tolist(), layer=l.round(4).tolist()))
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "dynamics_spread.json"), "w"))
