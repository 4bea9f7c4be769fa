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
    """csrc/conv_wd.h's kernels run two waves per SIMD: 256 registers each and no scratch (the fused tail sits at 248 - 256 before round 6's single set of conv3 fragments -; an edit that tips it over spills into scratch inside the chunk loop - it did during round 4)."""
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


def test_ring_kernel_fits_three_waves_per_simd_without_scratch_and_waits_with_counts(tmp_path):
    """csrc/conv1x1_ring.hip runs ten waves per CU (eight consumers + two loaders = three on two of the SIMDs): 168 registers and no
    scratch in every instantiation (the residual form sits AT 168; a second compute path or a resident bias tipped it into scratch during
    round 5 - scratch traffic would sit in the loaders' counted vmcnt stream).  Its synchronisation rests on two things the source cannot
    show: the loaders wait with COUNTED `s_waitcnt vmcnt(32)` in front of every barrier (a compiler-inserted `vmcnt(0)` there would
    drain the prefetch), and the consumers' K-step has no vector-memory instruction between its barrier and its MFMAs."""
    import proben_amd  # noqa: F401
    from proben_amd import build
    src = os.path.join(build.CSRC, "conv1x1_ring.hip")
    out = tmp_path / "conv1x1_ring.s"
    cmd = [build.hipcc(), "--offload-arch=" + build.ARCH, "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only", src, "-o", str(out),
           "-I", os.path.join(ROOT, "include"), "-I", build.CSRC] + build.EXTRA.get("conv1x1_ring.hip", build.EXTRA["default"])
    subprocess.check_call(cmd)
    text = out.read_text()
    bodies = re.findall(r"\.amdhsa_kernel (\S*conv1x1_ring_kernel\S*)(.*?)\.end_amdhsa_kernel", text, flags=re.S)
    assert len(bodies) == 4, "RELU x RES instantiations expected"
    # no scratch TRAFFIC: not one scratch_ / private-segment instruction in the translation unit and no VGPR spill.  (Since the round-6
    # removal of the ablation branches hipcc reserves a 36-byte private segment for the two residual-free instantiations that no
    # instruction addresses - the SGPR spills all go to VGPR lanes; a reservation is not traffic, so the bound is on the instructions.)
    assert "scratch_" not in text and not re.search(r"buffer_(load|store)_dword\w* v\d+, off, s\[\d+:\d+\], (0|s\d+)( offset:\d+)?$", text, flags=re.M)
    for count in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text):
        assert int(count) == 0
    for name, body in bodies:
        m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
        assert m and int(m.group(1)) <= 64, name + ": a private segment beyond the scavenger's slot"
        m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        assert m and int(m.group(1)) <= 168, (name, m and m.group(1))
    # per kernel: every s_barrier of the loader loops is preceded (within a few instructions) by the counted wait, never by vmcnt(0)
    for name, _ in bodies:
        code = text[text.index(name + ":"):]
        code = code[:code.index(".end_amdhsa_kernel")] if ".end_amdhsa_kernel" in code else code
        lines = [l.strip() for l in code.splitlines() if l.strip() and not l.strip().startswith(";")]
        counted = [i for i, l in enumerate(lines) if l == "s_waitcnt vmcnt(32)"]
        assert len(counted) >= 2, name + ": the two loader loops wait with vmcnt(32)"
        for i in counted:
            window = lines[i + 1:i + 8]
            assert any(l == "s_barrier" for l in window), (name, window)
            assert not any(l.startswith("s_waitcnt vmcnt(0)") for l in window[:window.index("s_barrier")]), (name, window)
        # the consumers' K-step: from its first MFMA to its last (16 MFMAs with the second half's fragment reads between them) there is no
        # vector-memory instruction and no vmcnt wait - the matrix waves only read LDS and multiply
        mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
        assert len(mf) == 16, (name, len(mf))
        block = lines[mf[0]:mf[-1] + 1]
        assert not any(l.startswith(("buffer_", "global_", "flat_", "scratch_")) or "vmcnt" in l for l in block), name
        assert sum(l.startswith("ds_read_b128") for l in block) == 8, name       # the pinned double-buffered schedule: sets 2 and 3 are read under MFMAs


def test_product_library_has_no_measurement_switches(tmp_path):
    """the ablation branches of the ring kernel (RingArgs::abl: "results are WRONG") and the hook that turned them on
    process-wide exist in the LAB build only (`python -m proben_amd.build --lab`, -DPE_LAB).  The default object must neither export
    the hook nor carry the field: every use of it in the source sits behind the RG_ABL() macro (the literal `false` without PE_LAB),
    the product build is the smaller program (-DPE_LAB is what adds the branches), and the shipped library exports no such symbol."""
    import proben_amd  # noqa: F401
    from proben_amd import build
    src = os.path.join(build.CSRC, "conv1x1_ring.hip")
    text = open(src).read()
    assert "a.abl" not in re.sub(r"#ifdef PE_LAB.*?#endif", "", text, flags=re.S)
    base = [build.hipcc(), "--offload-arch=" + build.ARCH, "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only", src,
            "-I", os.path.join(ROOT, "include"), "-I", build.CSRC] + build.EXTRA.get("conv1x1_ring.hip", build.EXTRA["default"])
    prod, lab = tmp_path / "prod.s", tmp_path / "lab.s"
    subprocess.check_call(base + ["-o", str(prod)])
    subprocess.check_call(base + ["-DPE_LAB", "-o", str(lab)])
    count = lambda t: sum(1 for ln in t.splitlines() if ln.startswith("\t") and not ln.lstrip().startswith((";", ".")))
    assert count(lab.read_text()) > count(prod.read_text())      # -DPE_LAB is what brings the branches in; the product build has fewer instructions
    if os.path.exists(build.LIB):
        syms = subprocess.run(["nm", "-D", "--defined-only", build.LIB], capture_output=True, text=True).stdout
        assert "pe_test_set_ring_ablation" not in syms and "pe_conv_wd_set_concurrent_streams" not in syms
    # round 6's diagnosis switches (PE_EXP_*: compile-time, variant libraries of scripts/lab/build_variant.py only) default to 0 in every source,
    # and nothing in the product build defines one
    for f in os.listdir(build.CSRC):
        if f.endswith((".h", ".hip", ".cpp")):
            for ln in open(os.path.join(build.CSRC, f)).read().splitlines():
                m = re.match(r"\s*#define\s+(PE_EXP_\w+)\s+(\S+)", ln)
                assert m is None or m.group(2) == "0", (f, ln)
    assert "PE_EXP_" not in open(os.path.join(build.PKG, "build.py")).read()
    assert not any("PE_EXP_" in flag for flags in build.EXTRA.values() for flag in flags)
