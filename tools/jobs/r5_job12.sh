#!/bin/bash
# ROW3 instances of conv_fwd_kernel: parity test, then same-box A/B of the train step (alternating), then the shape table
set -u
tag=${1:-r5_n}
out=$(pwd)/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_conv_gpu.py -q --maxfail=10 -k "row3" > $out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -3 $out/${tag}_tests.log; grep -n "^E " $out/${tag}_tests.log | head -12
for i in 1 2 3; do
  for v in 0 1; do
    SLN_CONV_ROW3=$v python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-strict > $out/${tag}_ab_row3_${v}_$i.json 2>/dev/null
    python3 -c "import json;d=json.load(open('$out/${tag}_ab_row3_${v}_$i.json'));print('ROW3=$v run $i', d['value'], d['ms_per_step'], d['roofline']['other_kernels']['conv_fwd_kernel<2>']['tflops'], d['roofline']['other_kernels']['conv_fwd_kernel<2>']['share_of_step_time'])"
  done
done
SLN_PROFILE_SHAPES=1 python3 bench.py --no-cpu-baseline --no-strict > /dev/null 2> $out/${tag}_shapes.txt
grep "conv_fwd_kernel<2>" $out/${tag}_shapes.txt | head -30
