"""Build libscenesplat_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libscenesplat_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "scenesplat_hip.h")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [HEADER]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    if os.environ.get("SS_EXTRA_HIPCC_FLAGS"):
        force = True      # a variant build: the up-to-date check only looks at file times (an unchanged tree would silently keep the old code)
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
               "-Wno-unused-value", "-Wno-unused-result",
               # keep MFMA accumulators in VGPRs: the softmax / epilogue VALU code consumes them directly,
               # the default AGPR form costs a v_accvgpr_read/write per element (measured: 144 of ~560
               # instructions per attention tile)
               "-mllvm", "-amdgpu-mfma-vgpr-form"] + os.environ.get("SS_EXTRA_HIPCC_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
