"""Round 5: where the 1x1 ring kernel's time goes - ablation builds (wrong results, timing only) on res4 conv1 and two more shapes.
The switches live in the LAB library only (round 6): build it first, this script points the binding at it.
Usage (GPU box): python -m proben_amd.build --lab  (from the repo root, with proben_amd.py on the path), then python scripts/archive/r05_ring_abl.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import proben_amd  # noqa: E402,F401
from proben_amd import _lib, layers as L  # noqa: E402
_lib.LIB_PATH = _lib.LIB_PATH.replace(".so", "_lab.so")      # the -DPE_LAB build: the product library has no ablation branch
assert os.path.exists(_lib.LIB_PATH), "build the lab library first: python -m proben_amd.build --lab"
from r05_ring import timeit  # noqa: E402

ABL = [(0, "full"), (1, "no pixel DMA"), (2, "no weight DMA"), (3, "no DMA at all"), (4, "no ds_read / MFMA"), (8, "no stores"),
       (12, "no MFMA, no stores (streaming only)"), (14, "pixel stream only"), (16, "pixels chunk-major (contiguous 16 KiB slabs)"),
       (16 + 12, "chunk-major, streaming only"), (16 + 14, "chunk-major pixel stream only"), (7, "barriers only"), (0, "full")]


def main():
    hooks = _lib.test_hooks()
    hooks.pe_test_set_conv_policy(9 + 64, 1)
    for (N, H, W, Cin, Cout) in [(32, 50, 64, 1024, 256), (16, 50, 64, 1024, 256), (32, 100, 128, 512, 256)]:
        x = torch.randn(N, H, W, Cin, device="cuda").half().relu()
        w = (torch.randn(Cout, 1, 1, Cin, device="cuda") / Cin ** 0.5).half()
        b = torch.randn(Cout, device="cuda")
        out = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.float16)
        M = N * H * W
        nbytes = (M * Cin + Cout * Cin + M * Cout) * 2
        print(f"--- N{N} {H}x{W} {Cin}->{Cout}: {nbytes / 1e6:.0f} MB algorithmic, {2.0 * M * Cin * Cout / 1e9:.1f} GFLOP")
        for wgs in (256, 512):
            hooks.pe_test_set_ring_wgs(wgs)
            for bits, name in ABL:
                hooks.pe_test_set_ring_ablation(bits)
                ms = timeit(lambda: L.conv2d_nhwc(x, w, b, kernel=1, relu=True, out=out))
                print(f"wgs {wgs} abl {bits:2d} {name:45s} {ms:.4f} ms  {nbytes / ms / 1e6:5.0f} GB/s-equivalent", flush=True)
    hooks.pe_test_set_ring_ablation(0)
    hooks.pe_test_set_ring_wgs(256)
    hooks.pe_test_set_conv_policy(9, 1)


if __name__ == "__main__":
    main()
