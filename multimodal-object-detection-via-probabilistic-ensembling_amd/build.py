"""Build libproben_hip.so (gfx950) in-tree with hipcc.  No torch headers involved:
the library is a plain C-ABI (include/proben_hip.h).

    python -m proben_amd.build        (or __graft_entry__.build())

Objects are cached under csrc/_build by source mtime; hipcc cross-compiles
without a GPU.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(PKG, "libproben_hip.so")
ARCH = "gfx950"

# Per-source extra flags.  The detection post-ops and ProbEn take discrete decisions on
# IoU / score thresholds, so they must round like the reference's separate mul/add/div.
# conv_wd9.hip: its asm statements own the accumulation registers a[0:255] - the compiler must not spill VGPRs into them - and its
# epilogue's scalar fp32 adds must stay scalar (tests/test_build_audit.py audits the emitted code).
WD9_FLAGS = ["-fno-slp-vectorize", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]
EXTRA = {"default": ["-ffp-contract=off"], "conv_igemm.hip": [], "conv_igemm2.hip": [], "conv_wd.hip": [], "bneck64.hip": [], "stem.hip": [],
         "conv_wd9.hip": WD9_FLAGS}


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libproben_hip.so")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    return max(os.path.getmtime(h) for h in hdrs)


def build(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    hdr_m = _deps_mtime()
    objs, rebuilt = [], False
    for src in sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        objs.append(op)
        if not force and os.path.exists(op) and os.path.getmtime(op) > max(os.path.getmtime(sp), hdr_m):
            continue
        cmd = [cc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", sp, "-o", op,
               "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wall", "-Wno-unused-function"]
        cmd += EXTRA.get(src, EXTRA["default"])
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        rebuilt = True
    if rebuilt or force or not os.path.exists(LIB):
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
