#!/bin/bash
set -u
tag=${1:-r5_d}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_f16_gpu.py tests/test_resnext_gpu.py tests/test_loader_gpu.py "tests/test_zz_dynamics_gpu.py::test_config5_resnext101_msc_train_step_full_depth" "tests/test_e2e_gpu.py" -q --maxfail=30 > $out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -3 $out/${tag}_tests.log; grep -E "^(FAILED|ERROR)" $out/${tag}_tests.log | head -30
python3 bench.py --config resnext --parts 1 --steps 10 --warmup 3 > $out/${tag}_resnext_p1.json 2> $out/${tag}_resnext_p1.err
head -c 330 $out/${tag}_resnext_p1.json; echo; tail -3 $out/${tag}_resnext_p1.err
python3 bench.py --config resnext --parts 2 --steps 10 --warmup 3 > $out/${tag}_resnext_p2.json 2> $out/${tag}_resnext_p2.err
head -c 330 $out/${tag}_resnext_p2.json; echo; tail -3 $out/${tag}_resnext_p2.err
