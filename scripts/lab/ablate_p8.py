"""Ablation of the phase-split 256x256 kernel (policy bit 7): full / no LDS-DMA / no MFMA / no fragment reads."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proben_amd  # noqa: E402
from proben_amd import _lib, layers as L  # noqa: E402
from ablate_conv import timeit  # noqa: E402

lib = _lib.lib()
for (N, H, W, Cin, Cout, k) in [(32000, 1, 1, 12544, 1024, 1), (8192, 1, 1, 8192, 8192, 1)]:
    x = torch.randn(N, H, W, Cin, device="cuda").half()
    w = (torch.randn(Cout, k, k, Cin, device="cuda") / (Cin * k * k) ** 0.5).half()
    b = torch.randn(Cout, device="cuda")
    out = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.float16)
    fl = 2.0 * N * H * W * Cout * k * k * Cin
    row = []
    lib.pe_set_conv_tile256(128)
    for abl in (0, 16, 32, 48, 1, 17, 33):
        lib.pe_set_conv_ablation(abl)
        ms = timeit(lambda: L.conv2d_nhwc(x, w, b, kernel=k, relu=True, out=out))
        row.append(f"abl{abl}: {ms:.4f}ms {fl / ms / 1e9:6.0f}TF")
    lib.pe_set_conv_ablation(0)
    lib.pe_set_conv_tile256(41)
    print(f"N{N} {H}x{W} {Cin}->{Cout} k{k} | " + " | ".join(row), flush=True)
