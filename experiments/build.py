"""Builds experiments/libcim_exp.so (gfx950): the superseded engines (csrc/gemm_engines.hip) + all Winograd algorithms
(csrc/winograd_all.hip) + the product's error-reporting unit.  python -m experiments.build"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libcim_exp.so")
CSRC = os.path.join(HERE, "csrc")
PROD_CSRC = os.path.join(REPO, "cim_amd", "csrc")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(PROD_CSRC, "common.cpp")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    deps = sources() + glob.glob(os.path.join(PROD_CSRC, "*.h")) + [os.path.join(HERE, "include", "cim_exp.h"),
                                                                     os.path.join(REPO, "include", "cim_hip.h")]
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I", PROD_CSRC,
               "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for c, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, c)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
