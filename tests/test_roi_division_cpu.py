"""csrc/roi_align.hip divides an accumulator by the sample count with one multiply and two FMAs; this compiles tests/csrc/div_check.c
(gcc, hardware FMA, no contraction) and runs the exhaustive comparison with IEEE division the kernel's comment cites."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_three_instruction_division_is_exact_for_every_sample_count(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "div_check")
    flags = ["-O2", "-ffp-contract=off", "-fno-fast-math"]
    probe = subprocess.run([gcc, "-mfma", "-E", "-x", "c", os.devnull], capture_output=True)
    if probe.returncode == 0:
        flags.append("-mfma")          # fmaf as one instruction (without it glibc's software fmaf gives the same bits, slower)
    subprocess.check_call([gcc, *flags, os.path.join(ROOT, "tests", "csrc", "div_check.c"), "-o", exe, "-lm"])
    counts, bad = map(int, subprocess.check_output([exe], timeout=600).split())
    assert counts == 176 and bad == 0, (counts, bad)
