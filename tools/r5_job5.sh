#!/bin/bash
# register-epilogue experiment: bit-exactness, then same-box A/B of the train step per kernel family
set -u
tag=${1:-r5_e}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_conv_gpu.py -q --maxfail=20 -k "register_epilogue or tile128x256 or tap_row or chained or bottleneck_stack or forward_matches or backward_matches" > $out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -3 $out/${tag}_tests.log; grep -E "^(FAILED|ERROR)" $out/${tag}_tests.log | head -20
export SLN_DEBUG_KNOBS=1
for d in 0 7 1 2 4 0 7; do
  SLN_CONV_DIRECT=$d python3 bench.py --steps 16 --warmup 3 --no-strict --no-cpu-baseline 2>/dev/null | tail -1 > $out/${tag}_ab_direct_$d.json
  python3 - <<PY
import json
d=json.load(open("$out/${tag}_ab_direct_$d.json")); r=d["roofline"]
o=r["other_kernels"]
print("DIRECT=$d", d["value"], d["ms_per_step"], "dom", r["frac"], r["avg_launch_us"], "128x256h", o.get("conv_fwd128x256h_kernel",{}).get("tflops"), "sat", d["config"]["conv_saturated_blocks"])
PY
done
