"""Builds csrc/libm3t_hip.so for gfx950 with hipcc (in-tree, no JIT cache)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libm3t_hip.so")
# m3t_hip_api.hip is the one translation unit; it includes the other .hip files and m3t_device.h
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc", ".map")))
HEADERS = [os.path.join(HERE, "..", "include", "m3t_hip.h"), os.path.join(HERE, "..", "include", "m3t_types.h")]
# -ffp-contract=off: the kernels follow the reference's f32 expression trees op by op
# -fvisibility=hidden: only what include/m3t_hip.h declares (default visibility pushed there) is a dynamic symbol
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function", "-fvisibility=hidden", "-fvisibility-inlines-hidden",
         "-Wl,--version-script=" + os.path.join(CSRC, "libm3t_hip.map")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB, os.path.join(CSRC, "m3t_hip_api.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
