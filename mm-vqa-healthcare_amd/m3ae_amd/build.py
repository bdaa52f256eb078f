"""Build libm3ae_hip.so (hipcc, --offload-arch=gfx950) in-tree.  Cross-compiles without a GPU.

    python -m m3ae_amd.build            # rebuild what changed
    python -m m3ae_amd.build --force

Diagnostic / timing-only variants (M3AE_EXTRA_HIPCC_FLAGS=...) never overwrite the product library: they are built into
lib_diag/ (own objects, own .so) and are only loaded when the caller exports M3AE_DIAGNOSTIC_LIB=1 (m3ae_amd/_lib.py).
The flag string is part of the staleness key of either directory (lib*/.build_flags).
"""
import concurrent.futures as cf
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(PKG))
CSRC = os.path.join(os.path.dirname(PKG), "csrc")
INC = os.path.join(ROOT, "include")
EXTRA = os.environ.get("M3AE_EXTRA_HIPCC_FLAGS", "").split()
OUT_DIR = os.path.join(PKG, "lib_diag" if EXTRA else "lib")
LIB = os.path.join(OUT_DIR, "libm3ae_hip.so")
STAMP = os.path.join(OUT_DIR, ".build_flags")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value",
         f"-I{INC}", f"-I{CSRC}"] + EXTRA


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    key = " ".join(FLAGS)
    old = open(STAMP).read() if os.path.exists(STAMP) else None
    if old != key:
        # objects built with other flags are stale whatever their mtime; a pre-stamp tree (old is None) may hold ANY build
        force = True
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INC, "m3ae_hip.h")]
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OUT_DIR, s[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC, *FLAGS, "-c", src, "-o", obj])

    def run(cmd):
        p = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, p.returncode, p.stdout + p.stderr

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for cmd, rc, out in ex.map(run, jobs):
                if verbose:
                    print("[m3ae build]", os.path.basename(cmd[-3]), "rc =", rc, flush=True)
                if rc != 0:
                    raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + out)
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        cmd, rc, out = run(cmd)
        if rc != 0:
            raise RuntimeError("link failed:\n" + out)
        if verbose:
            print("[m3ae build] linked", LIB, flush=True)
    with open(STAMP, "w") as f:
        f.write(key)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
