"""Build recipe for libhvpr_amd.so — the C-ABI HIP library (gfx950 only).

`python -m hvpr_amd.build` cross-compiles every csrc/*.hip with hipcc (no GPU needed) and links them
in-tree as hvpr_amd/libhvpr_amd.so, which travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libhvpr_amd.so")
OUT_CPU = os.path.join(HERE, "libhvpr_cpu.so")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# geometry that has to agree bit-for-bit with the CPU oracle: no FMA contraction (oracle/Makefile does the same)
PER_FILE = {"iou3d_nms.hip": ["-ffp-contract=off"], "voxelize.hip": ["-ffp-contract=off"], "pointnet2.hip": ["-ffp-contract=off"],
            "preprocess.hip": ["-ffp-contract=off"], "assign_loss.hip": ["-ffp-contract=off"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, timing=False):
    """timing=True: a second library, libhvpr_amd_timing.so, with -DHVPR_EXP_TIMING (in-kernel cycle counters printed by a
    few kernels; kernel experiments only — load it with HVPR_AMD_LIB=...)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "hvpr_amd.h"))
    objdir = os.path.join(HERE, "csrc", "_obj_timing" if timing else "_obj")
    out = os.path.join(HERE, "libhvpr_amd_timing.so") if timing else OUT
    extra = (["-DHVPR_EXP_TIMING"] + os.environ.get("HVPR_EXP_DEFINES", "").split()) if timing else []
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + COMMON + extra + PER_FILE.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(out, objs):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs)
    build_cpu(force, verbose)
    return out


def build_cpu(force=False, verbose=False):
    """libhvpr_cpu.so: the host-side natives of the data pipeline (g++, no GPU code)."""
    src = os.path.join(HERE, "csrc_cpu", "gt_sampling.cpp")
    if force or _stale(OUT_CPU, [src, os.path.join(os.path.dirname(HERE), "include", "hvpr_cpu.h")]):
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-ffp-contract=off", "-o", OUT_CPU, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT_CPU


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, timing="--timing" in sys.argv))
