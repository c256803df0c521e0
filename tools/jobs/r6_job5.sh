#!/bin/bash
# why does bench.py --gpus 8 fail under pytest and pass alone? hypothesis: nine processes with a GPU context (pytest's + 8 ranks)
set -u
out=gpurun_out; mkdir -p $out
cat > /tmp/holder.py <<'PY'
import subprocess, sys, os, torch
torch.zeros(1, device="cuda")            # this process holds a GPU context, like the pytest process does
n = sys.argv[1]
env = dict(os.environ, SLN_DIST_BACKEND="gloo", SLN_DIST_TIMEOUT_S="300")
r = subprocess.run([sys.executable, "bench.py", "--gpus", n, "--steps", "2", "--warmup", "1", "--settle", "1", "--batch", "1",
                    "--dim", "128", "--arch", "resnet50", "--no-cpu-baseline", "--no-strict"], env=env, capture_output=True, text=True)
print("holder + %s ranks: rc %d" % (n, r.returncode))
print("\n".join(l for l in r.stderr.splitlines() if l.startswith("bench.py:"))[:600])
PY
timeout 900 python3 /tmp/holder.py 8
timeout 900 python3 /tmp/holder.py 7
rocminfo 2>/dev/null | grep -i -m3 "queue\|Max Waves" ; ls /sys/class/kfd/kfd/topology/nodes/ 2>/dev/null | head
