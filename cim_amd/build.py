"""Builds libcim_hip.so (all HIP kernels + the C ABI of include/cim_hip.h) for gfx950, in-tree.

    python -m cim_amd.build            # rebuild if sources are newer than the library

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to the GPU box.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcim_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "cim_hip.h")]
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, out=None):
    global LIB
    if out is not None:
        LIB = out
        force = True
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function"] + os.environ.get("CIM_HIPCC_FLAGS", "").split() + \
          ["-o", LIB + ".tmp"] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    outs = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--out=")]
    build(force="--force" in sys.argv, verbose=True, out=outs[0] if outs else None)
    print(LIB)
