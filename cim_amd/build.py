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


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "cim_hip.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    deps = sources() + _headers()
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, out=None):
    """One object per source file under csrc/_obj/ (rebuilt only when the source or a header is newer), then one
    link: touching one kernel file costs one compile.  CIM_HIPCC_FLAGS / --out builds (ablations) bypass the cache."""
    global LIB
    extra = os.environ.get("CIM_HIPCC_FLAGS", "").split()
    cache = not extra and out is None
    if out is not None:
        LIB = out
        force = True
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + extra
    objdir = os.path.join(CSRC, "_obj" if cache else "_obj_%d" % os.getpid())
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if cache and not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            continue
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
        if len(procs) >= int(os.environ.get("CIM_BUILD_JOBS", "4")):
            c, p = procs.pop(0)
            if p.wait() != 0:
                raise subprocess.CalledProcessError(p.returncode, c)
    for c, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, c)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    if not cache:
        for o in objs:
            os.remove(o)
        os.rmdir(objdir)
    return LIB


if __name__ == "__main__":
    outs = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--out=")]
    build(force="--force" in sys.argv, verbose=True, out=outs[0] if outs else None)
    print(LIB)
