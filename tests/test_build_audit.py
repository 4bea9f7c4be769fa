"""Audit of the code hipcc emits for csrc/conv_wd9.hip (CPU-only: hipcc cross-compiles).  The kernel's inline-asm MFMAs address the
accumulation registers a[0:255] literally; that is only sound if the compiler (1) never spills (a spilled VGPR may be parked in an
AGPR, or the scratch traffic lands in the counted vmcnt stream), and (2) emits no accumulator-file instruction of its own."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_wd9_kernels_do_not_spill_and_leave_the_accumulation_registers_alone(tmp_path):
    import proben_amd  # noqa: F401
    from proben_amd import build
    src = os.path.join(build.CSRC, "conv_wd9.hip")
    out = tmp_path / "conv_wd9.s"
    cmd = [build.hipcc(), "--offload-arch=" + build.ARCH, "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only", src, "-o", str(out),
           "-I", os.path.join(ROOT, "include"), "-I", build.CSRC] + build.EXTRA["conv_wd9.hip"]
    subprocess.check_call(cmd)
    text = out.read_text()
    kernels = re.findall(r"\.amdhsa_kernel (\S*wd9\S*)", text)
    assert kernels, "no wd9 kernel in the translation unit"
    # (1) no scratch, no spills: per-kernel metadata
    for name, body in re.findall(r"\.amdhsa_kernel (\S*wd9\S*)(.*?)\.end_amdhsa_kernel", text, flags=re.S):
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body), name
        m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        assert m and int(m.group(1)) <= 512, name
    for count in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text):
        assert int(count) == 0
    assert "scratch_" not in text
    # (2) outside ;;#ASMSTART .. ;;#ASMEND the compiler issues no MFMA, and touches no accumulation register that the asm statements
    #     own: all of a[0:255] in the conv_wd9.h kernels
    inside, kernel = False, ""
    for line in text.splitlines():
        m = re.match(r"^(_Z\S+):", line)
        if m:
            kernel = m.group(1)
        if "#ASMSTART" in line:
            inside = True
        elif "#ASMEND" in line:
            inside = False
        elif not inside and not line.lstrip().startswith(";"):
            assert not re.search(r"\bv_mfma_\w+\b", line), "compiler-generated MFMA: " + line.strip()
            if "accvgpr" in line or re.search(r"\ba\[?\d+", line):
                regs = [int(r) for r in re.findall(r"\ba(\d+)\b", line)] + [int(b) for _, b in re.findall(r"\ba\[(\d+):(\d+)\]", line)]
                assert regs and max(regs) < 0, f"{kernel}: compiler code touches an accumulation register of the asm statements: {line.strip()}"


def test_two_wave_weights_direct_kernels_fit_two_per_simd_without_scratch(tmp_path):
    """csrc/conv_wd.h's kernels run two waves per SIMD: 256 registers each and no scratch (the fused tail sits AT 256 since its
    line-store epilogue; an edit that tips it over spills into scratch inside the chunk loop - it did during round 4)."""
    import proben_amd  # noqa: F401
    from proben_amd import build
    src = os.path.join(build.CSRC, "conv_wd.hip")
    out = tmp_path / "conv_wd.s"
    cmd = [build.hipcc(), "--offload-arch=" + build.ARCH, "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only", src, "-o", str(out),
           "-I", os.path.join(ROOT, "include"), "-I", build.CSRC] + build.EXTRA.get("conv_wd.hip", build.EXTRA["default"])
    subprocess.check_call(cmd)
    text = out.read_text()
    seen = 0
    for name, body in re.findall(r"\.amdhsa_kernel (\S*conv3x3_wd_kernel\S*)(.*?)\.end_amdhsa_kernel", text, flags=re.S):
        seen += 1
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body), name + ": scratch in use"
        m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        assert m and int(m.group(1)) <= 256, name
    assert seen >= 3, "pure, fused-head and fused-tail instantiations expected"
