#!/bin/bash
# full GPU suite (-x, as the driver runs it) at the ROW3 head, then a same-box A/B of the train step under debug knobs
set -u
tag=${1:-r5_o}
out=$(pwd)/gpurun_out
mkdir -p $out
python3 -m pytest tests -x -q -m gpu > $out/${tag}_gpu_suite.log 2>&1
echo "suite rc=$?"; tail -3 $out/${tag}_gpu_suite.log; grep -E "^(FAILED|ERROR)" $out/${tag}_gpu_suite.log | head
export SLN_DEBUG_KNOBS=1
for i in 1 2 3; do
  for v in 0 1; do
    SLN_CONV_ROW3=$v python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-strict > $out/${tag}_ab_row3_${v}_$i.json 2>/dev/null
    python3 -c "import json;d=json.load(open('$out/${tag}_ab_row3_${v}_$i.json'));print('ROW3=$v run $i', d['value'], d['ms_per_step'], d['roofline']['other_kernels']['conv_fwd_kernel<2>']['tflops'], d['roofline']['other_kernels']['conv_fwd_kernel<2>']['share_of_step_time'])"
  done
done
