#!/bin/bash
# round 6, fourth GPU call: why the 8-rank self-launch fails, the captured resnext step, the new precision tests
set -u
out=gpurun_out; mkdir -p $out
SLN_DIST_BACKEND=gloo SLN_DIST_TIMEOUT_S=300 timeout 900 python3 bench.py --gpus 8 --steps 2 --warmup 1 --settle 1 --batch 1 --dim 128 --arch resnet50 --no-cpu-baseline --no-strict > $out/r6_d_world8.json 2> $out/r6_d_world8.err; echo "world8 rc=$?"
grep -n "Error\|error" $out/r6_d_world8.err | grep -v "Connection closed" | head -20
head -c 600 $out/r6_d_world8.json; echo
python3 bench.py --config resnext --parts 1 --steps 10 --warmup 3 > $out/r6_d_resnext_p1.json 2> $out/r6_d_resnext_p1.err; echo "resnext eager rc=$?"; head -c 250 $out/r6_d_resnext_p1.json; echo
python3 bench.py --config resnext --parts 1 --steps 10 --warmup 3 --graph > $out/r6_d_resnext_p1_graph.json 2> $out/r6_d_resnext_p1_graph.err; echo "resnext graph rc=$?"; head -c 250 $out/r6_d_resnext_p1_graph.json; echo; tail -5 $out/r6_d_resnext_p1_graph.err
python3 -m pytest tests/test_precision_gpu.py -m gpu -x -q -s > $out/r6_d_precision_tests.log 2>&1; echo "precision tests rc=$?"; tail -5 $out/r6_d_precision_tests.log
