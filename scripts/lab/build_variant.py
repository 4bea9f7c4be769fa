"""Lab only: build a VARIANT of the library for a same-box A/B - the product objects of csrc/_build, except that the sources that mention a
`PE_EXP_` macro are recompiled with the given -D definitions - as  <package>/libproben_hip_exp_<tag>.so  (git-ignored like every .so).
Nothing in the product, the tests or bench.py loads such a library; scripts/lab/ab_kernels.py and scripts/lab/bench_with_lib.py do.

    python scripts/lab/build_variant.py b64nt -DPE_EXP_B64_OUT_AUX=2
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import proben_amd  # noqa: E402,F401
from proben_amd import build  # noqa: E402


def main():
    tag, defs = sys.argv[1], sys.argv[2:]
    build.build(verbose=False)                      # product objects up to date
    obj_dir = build.OBJ + "_exp_" + tag
    os.makedirs(obj_dir, exist_ok=True)
    hdr_exp = [h for h in os.listdir(build.CSRC) if h.endswith(".h") and "PE_EXP_" in open(os.path.join(build.CSRC, h)).read()]
    objs = []
    for src in build.sources():
        text = open(os.path.join(build.CSRC, src)).read()
        touched = "PE_EXP_" in text or any('"%s"' % h in text for h in hdr_exp)
        base = src.rsplit(".", 1)[0] + ".o"
        if not touched:
            objs.append(os.path.join(build.OBJ, base))
            continue
        op = os.path.join(obj_dir, base)
        objs.append(op)
        cmd = [build.hipcc(), "--offload-arch=" + build.ARCH, "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", os.path.join(build.CSRC, src), "-o", op,
               "-I", os.path.join(build.ROOT, "include"), "-I", build.CSRC, "-Wall", "-Wno-unused-function"]
        cmd += build.EXTRA.get(src, build.EXTRA["default"]) + defs
        print("[variant %s] %s %s" % (tag, src, " ".join(defs)), flush=True)
        subprocess.check_call(cmd)
    lib = build.LIB.replace(".so", "_exp_%s.so" % tag)
    subprocess.check_call([build.hipcc(), "--offload-arch=" + build.ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
