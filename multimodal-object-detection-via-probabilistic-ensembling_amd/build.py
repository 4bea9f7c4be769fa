"""Build libproben_hip.so (gfx950) in-tree with hipcc.  No torch headers involved:
the library is a plain C-ABI (include/proben_hip.h).

    python -m proben_amd.build        (or __graft_entry__.build())

Objects are cached under csrc/_build by source mtime; hipcc cross-compiles
without a GPU.

    python -m proben_amd.build --lab  builds libproben_hip_lab.so with -DPE_LAB (objects under
csrc/_build_lab): the product library plus the kernels' MEASUREMENT switches (ablation
branches that give wrong results, e.g. csrc/conv1x1_ring.hip RingArgs::abl).  Nothing in the
product, the tests or bench.py loads it; scripts/archive/r05_ring_abl.py does, by pointing
proben_amd._lib.LIB_PATH at it before the first call.
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


def build(verbose=True, force=False, lab=False):
    obj_dir, lib_path = (OBJ + "_lab", LIB.replace(".so", "_lab.so")) if lab else (OBJ, LIB)
    os.makedirs(obj_dir, exist_ok=True)
    cc = hipcc()
    hdr_m = _deps_mtime()
    objs, rebuilt = [], False
    for src in sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(obj_dir, src.rsplit(".", 1)[0] + ".o")
        objs.append(op)
        if not force and os.path.exists(op) and os.path.getmtime(op) > max(os.path.getmtime(sp), hdr_m):
            continue
        cmd = [cc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", sp, "-o", op,
               "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wall", "-Wno-unused-function"]
        cmd += EXTRA.get(src, EXTRA["default"]) + (["-DPE_LAB"] if lab else [])
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        rebuilt = True
    if rebuilt or force or not os.path.exists(lib_path):
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib_path] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib_path


if __name__ == "__main__":
    build(force="--force" in sys.argv, lab="--lab" in sys.argv)
