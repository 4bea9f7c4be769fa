"""A/B of the conv kernel families on representative layer shapes (post-ReLU inputs).  Usage (GPU box): python scripts/ablate_conv.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proben_amd  # noqa: E402
from proben_amd import _lib, layers as L  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    hooks = _lib.test_hooks()   # csrc/test_hooks.h: kernel-selection policy (not part of the product ABI)
    shapes = [(32, 200, 256, 256, 256, 3), (32, 50, 64, 256, 256, 3), (32, 100, 128, 256, 256, 3), (32, 50, 64, 1024, 256, 1),
              (32, 50, 64, 256, 1024, 1), (32000, 1, 1, 12544, 1024, 1)]
    for (N, H, W, Cin, Cout, k) in shapes:
        x = torch.randn(N, H, W, Cin, device="cuda").half().relu()
        w = (torch.randn(Cout, k, k, Cin, device="cuda") / (Cin * k * k) ** 0.5).half()
        b = torch.randn(Cout, device="cuda")
        out = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.float16)
        fl = 2.0 * N * H * W * Cout * k * k * Cin
        row = []
        # (reuse3x3, tile bits); the register-staged first-generation kernel ("i1" in the r01 / r02 tables) left the library in r03
        for reuse, tile in [(0, 1), (1, 0), (1, 1), (1, 3), (1, 9), (1, 25)]:
            hooks.pe_test_set_conv_policy(tile, reuse)
            ms = timeit(lambda: L.conv2d_nhwc(x, w, b, kernel=k, relu=True, out=out))
            row.append(f"{'kw-reuse' if reuse else 'per-tap'} t{tile}: {ms:.4f}ms {fl / ms / 1e9:6.0f}TF")
        hooks.pe_test_set_conv_policy(9, 1)
        if k == 3 and L.conv_wd_supported(3, 1, H, W, Cin, Cout):
            pk = L.conv_wd_pack(w)
            ms = timeit(lambda: L.conv3x3_wd(x, pk, b, Cout, relu=True, out=out))
            row.append(f"weights-direct: {ms:.4f}ms {fl / ms / 1e9:6.0f}TF")
        print(f"N{N} {H}x{W} {Cin}->{Cout} k{k} | " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
