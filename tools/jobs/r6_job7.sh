#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
python3 -m pytest tests/test_tail_gpu.py tests/test_parallel_gpu.py -m gpu -x -q > $out/r6_g_tail_then_parallel.log 2>&1; echo "tail+parallel rc=$?"; tail -3 $out/r6_g_tail_then_parallel.log; grep -n "bench.py:" $out/r6_g_tail_then_parallel.log | head -5
python3 -m pytest tests/test_precision_gpu.py tests/test_parallel_gpu.py -m gpu -x -q > $out/r6_g_precision_then_parallel.log 2>&1; echo "precision+parallel rc=$?"; tail -3 $out/r6_g_precision_then_parallel.log; grep -n "bench.py:" $out/r6_g_precision_then_parallel.log | head -5
