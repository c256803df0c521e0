#!/bin/bash
# shape-level A/B of two ROW3 tile choices inside the train step (per-shape tables of bench.py)
set -u
tag=${1:-r5_u}
out=$(pwd)/gpurun_out
mkdir -p $out
export SLN_DEBUG_KNOBS=1
python3 -m pytest tests/test_conv_gpu.py -q -k "row3" > $out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -1 $out/${tag}_tests.log
for cfg in "base" "SLN_CONV_ROW3_NARROW=1" "SLN_WGRAD_ROW3_TM64=1" "base" "SLN_CONV_ROW3_NARROW=1" "SLN_WGRAD_ROW3_TM64=1"; do
  if [ "$cfg" = "base" ]; then e=""; else e="$cfg"; fi
  env $e SLN_PROFILE_SHAPES=1 python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-strict > $out/${tag}_tmp.json 2> $out/${tag}_tmp.txt
  echo "== $cfg: $(python3 -c "import json;d=json.load(open('$out/${tag}_tmp.json'));print(d['value'], d['ms_per_step'])")"
  grep "C128->128 k3\|C64->64 k3" $out/${tag}_tmp.txt | grep "N16" | head -6
done
