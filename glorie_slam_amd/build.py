"""Build recipe for libglorie_hip.so (gfx950 only, in-tree).

    python glorie_slam_amd/build.py            # incremental
    python glorie_slam_amd/build.py --force

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the
gpurun snapshot.  Objects are cached under glorie_slam_amd/lib/obj keyed by source mtime.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libglorie_hip.so")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", "-I", INCLUDE]
# kernel experiments (-DEXP_...): extra flags for every source, or - GLORIE_EXTRA_HIPFLAGS_ONLY=corr_dm.hip - for one file
EXTRA = os.environ.get("GLORIE_EXTRA_HIPFLAGS", "").split()
EXTRA_ONLY = os.environ.get("GLORIE_EXTRA_HIPFLAGS_ONLY", "")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _flags(src):
    return FLAGS + (EXTRA if (not EXTRA_ONLY or os.path.basename(src) == EXTRA_ONLY) else [])


def _digest(paths):
    h = hashlib.sha1()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(_flags(paths[0])).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hiph")]
    headers.append(os.path.join(INCLUDE, "glorie_hip.h"))
    srcs = _sources()
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, s[:-4] + ".o")
        stamp = obj + ".sha1"
        dig = _digest([src] + headers)
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.exists(stamp)
                and open(stamp).read() == dig):
            continue
        jobs.append((src, obj, stamp, dig))

    def compile_one(job):
        src, obj, stamp, dig = job
        cmd = [HIPCC] + _flags(src) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write(dig)
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for warn in ex.map(compile_one, jobs):
                if warn and verbose:
                    print(warn, file=sys.stderr)
    if jobs or not os.path.exists(LIB_PATH):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
