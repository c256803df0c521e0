"""Builds libsln_amodal_hip.so in-tree with hipcc for gfx950 (cross-compiles
without a GPU).  Exact-arithmetic kernels (NMS, crop_and_resize, label decode,
proposal decode) are compiled with -ffp-contract=off and correctly rounded
division so index-producing float math rounds like the reference CPU path."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libsln_amodal_hip.so")

EXACT = ["abi.hip", "nms.hip", "crop_and_resize.hip", "label_decode.hip", "proposal.hip",
         "pyramid_crop.hip", "topk.hip", "tail.hip", "optim.hip", "fpn_merge.hip", "maxpool.hip", "glm_tail.hip"]
FAST = [f for f in sorted(os.listdir(HERE)) if f.endswith(".hip") and f not in EXACT]

COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall",
          "-Wno-unused-function", "-fgpu-rdc" if False else "-fno-gpu-rdc"]
EXACT_FLAGS = ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math"]
# conv.hip: the epilogue's slab loop must unroll completely whatever the number of epilogue variants inlined
# into it (its index selects accumulator REGISTERS; a rolled loop would put the 128 accumulators in scratch)
FAST_FLAGS = ["-mllvm", "-pragma-unroll-threshold=131072", "-Werror=pass-failed"]


def _stale(out, deps):
    return (not os.path.exists(out)) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "..", "include", "sln_amodal.h"))
    objs, jobs = [], []
    for name in FAST + EXACT:            # (conv.hip first: it takes ~4 min of the build, the others seconds each)
        src = os.path.join(HERE, name)
        obj = os.path.join(HERE, name.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            flags = COMMON + (EXACT_FLAGS if name in EXACT else FAST_FLAGS)
            jobs.append([hipcc, "-c", "-x", "hip", src, "-o", obj] + flags)
    if jobs:
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:     # the translation units are independent
            list(pool.map(run, jobs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
