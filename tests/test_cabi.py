"""The C-ABI library builds, loads, and exports every symbol include/*.h declares.
No compute calls (CPU-only container)."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__ as g
    g.build()
    import proben_amd
    return proben_amd._lib.LIB_PATH


def declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        syms |= set(re.findall(r"\b(pe_[a-z0-9_]+)\s*\(", txt))
    return sorted(syms)


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "pe_proben_fuse_batch" in syms and "pe_last_error" in syms


def test_library_exports_every_declared_symbol(built_lib):
    L = ctypes.CDLL(built_lib)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_python_binding_covers_header(built_lib):
    import proben_amd
    bound = set(proben_amd._lib.SIGNATURES) | {"pe_last_error"}
    assert set(declared_symbols()) <= bound, set(declared_symbols()) - bound
    assert proben_amd._lib.lib().pe_version() >= 1


def test_product_path_refuses_cpu_tensors(built_lib):
    import torch
    import proben_amd
    import proben_amd.fusion as F
    z = torch.zeros((0, 4), dtype=torch.float64)
    with pytest.raises(proben_amd._lib.HipLibraryError):
        F.fuse_batch(z, z[:, 0], z[:, :3], z[:, 0], z[:, 0].int(), torch.zeros(1, dtype=torch.int32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "multimodal-object-detection-via-probabilistic-ensembling_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_argument_checks_answer_before_any_device_work(built_lib):
    """Error behaviour of the C-ABI without a GPU: argument checks come first and explain themselves through pe_last_error()
    (null pointers, misaligned flat buffers of the fused SGD step, a ROI order workspace that is missing, bad stream counts)."""
    import proben_amd
    L = proben_amd._lib.lib()
    err = lambda: L.pe_last_error().decode()
    assert L.pe_sgd_momentum_f32(None, None, None, None, 0, 0.1, 0.9, 0.0, 1.0, 1, None) == 0          # empty range: nothing to do
    assert L.pe_sgd_momentum_f32(None, None, None, None, 8, 0.1, 0.9, 0.0, 1.0, 1, None) != 0 and "null pointer" in err()
    assert L.pe_sgd_momentum_f32(4096 + 8, 4096, 4096, None, 8, 0.1, 0.9, 0.0, 1.0, 1, None) != 0 and "16-byte aligned" in err()
    assert L.pe_sgd_momentum_f32(4096, 4096, 4096, 4096 + 4, 8, 0.1, 0.9, 0.0, 1.0, 1, None) != 0 and "16-byte aligned" in err()
    feats = (ctypes.c_void_p * 4)(4096, 4096, 4096, 4096)
    hw = (ctypes.c_int32 * 8)(200, 256, 100, 128, 50, 64, 25, 32)
    sc = (ctypes.c_float * 4)(0.25, 0.125, 0.0625, 0.03125)
    st = L.pe_roi_align_nhwc_sorted(feats, hw, sc, 4, 2, 256, 0, 4096, 1000, None, 7, 7, 0, 1, 4096, None, None, None)
    assert st != 0 and "order workspace" in err()
