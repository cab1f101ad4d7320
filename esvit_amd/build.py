"""Build libesvit_hip.so (the C-ABI HIP library) in-tree for gfx950.

    python -m esvit_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the CPU-only container as well as on the
MI355X box.  Objects are cached by mtime under esvit_amd/csrc/build/; the shared library is
written to esvit_amd/lib/libesvit_hip.so (git-ignored, travels with the gpurun snapshot).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libesvit_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",  # include/esvit_hip.h is the export list
         "-Wno-unused-result", "-I", INCLUDE, "-I", CSRC]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(INCLUDE, "esvit_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force, hdr_mtime):
    obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(path)
            and os.path.getmtime(obj) >= hdr_mtime):
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", path, "-o", obj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stdout))
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdr = _deps_mtime()
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, hdr), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    stale = os.path.exists(LIB) and any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)  # (an object compiled by hand / by tools/)
    if rebuilt or stale or not os.path.exists(LIB) or force:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout)
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
