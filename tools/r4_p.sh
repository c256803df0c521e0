#!/bin/bash
cd $GRAFT_REPO_ROOT
export SLN_DEBUG_KNOBS=1
export SLN_CONV_TILE128H=2
for skew in 0 524288 1572864 3670016 7864320; do
  echo "== dbg $skew (sleep x $(( (skew >> 20) + 1 )) if on)"
  SLN_HBM_LAYERS_DBG=$skew python3 tools/hbm_layers.py "C4 conv3" "C4 conv1" 2>&1 | grep -E "^C|parts only|res16" | cut -c1-75
done
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from sln_amodal_amd import conv_hip
def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
N, C, H = 16, 256, 64
x = torch.randn(N, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(C, C, 3, 3, device="cuda") * 0.02
sc, sf = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
xp, xq = conv_hip.act_parts(x, 2)
slot = conv_hip._slot(w, ("y", H, H))
A = (xp, N, H, H, conv_hip.wsrc(w, 2), C, 3, 3, (1, 1), (1, 1), 1, 1, H, H)
for _ in range(2):
    conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, yslot=slot, xq=xq)
f = lambda: conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, want_y=False, yslot=slot, xq=xq)
fl = 2.0 * N * H * H * C * C * 9
for t128 in ("0", "2"):
    os.environ["SLN_CONV_TILE128H"] = t128
    for dbg in (0, 524288, 1572864, 3670016, 7864320, 15728640 | 524288):
        os.environ["SLN_CONV_DBG"] = str(dbg)
        t = timeit(f)
        print("3x3 256->256 @64 parts only  TILE128H=%s dbg=%d: %.3f ms %.0f TF" % (t128, dbg, t, fl / t / 1e9))
        if t128 == "0":
            break
PY
