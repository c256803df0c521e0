#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
echo "cgroup memory.max: $(cat /sys/fs/cgroup/memory.max 2>/dev/null)  current: $(cat /sys/fs/cgroup/memory.current 2>/dev/null)"; free -g | head -2; nproc
python3 -m pytest tests/test_parallel_gpu.py -m gpu -x -q > $out/r6_f_parallel_gpu_alone.log 2>&1; echo "alone rc=$?"; tail -3 $out/r6_f_parallel_gpu_alone.log
# the same test behind the suite's heavy files (the parent then holds what they left)
( while true; do echo "mem current $(cat /sys/fs/cgroup/memory.current 2>/dev/null) $(free -g | awk 'NR==2{print $3" used "$7" avail"}')"; sleep 10; done ) > $out/r6_f_mem_trace.txt 2>&1 &
MON=$!
python3 -m pytest tests/test_configs_gpu.py tests/test_loader_gpu.py tests/test_parallel_gpu.py -m gpu -x -q > $out/r6_f_parallel_gpu_behind.log 2>&1; echo "behind rc=$?"; tail -3 $out/r6_f_parallel_gpu_behind.log
kill $MON
grep -n "bench.py:" $out/r6_f_parallel_gpu_behind.log | head; tail -12 $out/r6_f_mem_trace.txt
dmesg 2>/dev/null | tail -5
