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


# Kernels whose epilogue operands are loaded from inline asm and waited for by the kernel's own s_waitcnt (gemm_nt_common.h:
# gload16_asm): between such a load and its wait the destination registers are in flight, and a register spill there would save
# garbage.  hipcc cannot know; this build checks instead: every instantiation of these kernels except the catch-all epilogue class
# (EPI_ANY = 5, which keeps compiler-visible loads) must come out without scratch memory.
ASM_LOAD_SOURCES = {"gemm_nt_pp2.hip": r"gemm_nt_pp2_kernel", "gemm_mfma.hip": r"gemm_nt_(bf16|pp|pp_persistent)_kernel"}
REMARK = "-Rpass-analysis=kernel-resource-usage"


def check_no_scratch(src_name, hipcc_output):
    """Parse hipcc's kernel-resource-usage remarks of one source; raise if a kernel with asm loads uses scratch."""
    import re
    pat = re.compile(ASM_LOAD_SOURCES[src_name])
    name, seen, bad = None, 0, []
    for line in hipcc_output.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            continue
        m = re.search(r"remark:\s+ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and pat.search(name):
            seen += 1
            if int(m.group(1)) and not re.search(r"Li5E(L[bi][0-9]E)?E+v", name):   # ...<..., EPI_ANY>: visible loads, may spill
                bad.append((name, int(m.group(1))))
    if not seen:
        raise RuntimeError(f"{src_name}: no kernel-resource-usage remarks for {ASM_LOAD_SOURCES[src_name]} (hipcc output format changed?)")
    if bad:
        raise RuntimeError(f"{src_name}: kernels with inline-asm epilogue loads use scratch memory (a spill between an asm load and "
                           f"its wait would save an in-flight register): {bad}")
    return seen


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
            jobs.append([HIPCC, *FLAGS, *([REMARK] if s in ASM_LOAD_SOURCES else []), "-c", src, "-o", obj])

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
                if os.path.basename(cmd[-3]) in ASM_LOAD_SOURCES and not EXTRA:
                    try:
                        n = check_no_scratch(os.path.basename(cmd[-3]), out)
                    except RuntimeError:
                        os.remove(cmd[-1])   # never link (or keep as up to date) an object that failed the check
                        raise
                    if verbose:
                        print(f"[m3ae build] {os.path.basename(cmd[-3])}: {n} kernels with asm epilogue loads, none uses scratch", flush=True)
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
