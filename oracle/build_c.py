"""Build the oracle's C restatement (test infrastructure): oracle/_build/liboracle.so (gcc)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "liboracle.so")


def build():
    srcs = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".c"))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) > os.path.getmtime(s) for s in srcs):
        return OUT
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", OUT] + srcs + ["-lm"])
    return OUT


if __name__ == "__main__":
    print(build())
