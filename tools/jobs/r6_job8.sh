#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
python3 -m pytest tests/test_tail_gpu.py tests/test_parallel_gpu.py -m gpu -x -q > $out/r6_h_tail_then_parallel.log 2>&1; echo "tail+parallel rc=$?"; tail -2 $out/r6_h_tail_then_parallel.log
grep -v "rank[1-7]\]\|Gloo\|socket.cpp\|UserWarning\|detach()\|float(loss)\|amdgpu.ids" /tmp/sln_world8_stderr.txt | head -60
