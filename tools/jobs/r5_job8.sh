#!/bin/bash
set -u
tag=${1:-r5_i}
out=$(pwd)/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py -q --maxfail=10 > $out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -2 $out/${tag}_tests.log; grep -E "^(FAILED|ERROR)" $out/${tag}_tests.log | head
export SLN_DEBUG_KNOBS=1
for d in 0 1 0 1 0 1; do
  SLN_CONV_EPI64=$d python3 bench.py --steps 16 --warmup 3 --no-strict --no-cpu-baseline 2>/dev/null | tail -1 > $out/${tag}_ab_epi64_$d.json
  python3 - <<PY
import json
d=json.load(open("$out/${tag}_ab_epi64_$d.json")); r=d["roofline"]; o=r["other_kernels"]
print("EPI64=$d", d["value"], d["ms_per_step"], "fwd128family", o.get("conv_fwd_kernel<2>",{}).get("tflops"), o.get("conv_fwd_kernel<2>",{}).get("share_of_step_time"))
PY
done
