"""Round 5: the persistent loader / consumer 1x1 kernel (csrc/conv1x1_ring.hip) against conv_igemm2 - bit comparison and timing per shape.
Usage (GPU box): python scripts/r05_ring.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proben_amd  # noqa: E402,F401
from proben_amd import _lib, layers as L  # noqa: E402


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    hooks = _lib.test_hooks()
    torch.manual_seed(0)
    print("# bit comparison: ring (policy 9 + 64) vs conv_igemm2 (policy 9)")
    for (N, H, W, Cin, Cout, relu, bias) in [(32, 50, 64, 1024, 256, 1, 1), (3, 50, 64, 1024, 256, 1, 1), (1, 13, 16, 2048, 256, 0, 1),
                                             (2, 25, 32, 2048, 512, 1, 0), (1000, 1, 1, 1024, 1024, 1, 1), (1, 7, 9, 512, 256, 1, 1),
                                             (16, 50, 64, 1024, 256, 1, 1), (5, 1, 1, 1024, 256, 0, 1), (32000, 1, 1, 1024, 1024, 1, 1),
                                             (4000, 1, 1, 12544, 1024, 1, 1), (32, 25, 32, 2048, 512, 1, 1)]:
        x = torch.randn(N, H, W, Cin, device="cuda").half().relu()
        w = (torch.randn(Cout, 1, 1, Cin, device="cuda") / Cin ** 0.5).half()
        b = torch.randn(Cout, device="cuda") if bias else None
        hooks.pe_test_set_conv_policy(9, 1)
        ref = L.conv2d_nhwc(x, w, b, kernel=1, relu=bool(relu))
        hooks.pe_test_set_conv_policy(9 + 64, 1)
        out = torch.full_like(ref, float("nan"))
        L.conv2d_nhwc(x, w, b, kernel=1, relu=bool(relu), out=out)
        torch.cuda.synchronize()
        same = torch.equal(ref, out)
        md = (ref.float() - out.float()).abs().max().item()
        print(f"N{N} {H}x{W} {Cin}->{Cout} relu{relu} bias{bias}: identical={same} max|d|={md:.3g} nan={int(torch.isnan(out).sum())}", flush=True)
    print("# timing (ms per launch; GB/s = algorithmic bytes)")
    for shape in [(32, 50, 64, 1024, 256), (16, 50, 64, 1024, 256), (32, 25, 32, 2048, 512), (32, 25, 32, 2048, 256),
                  (32000, 1, 1, 1024, 1024), (32, 100, 128, 512, 256), (32000, 1, 1, 12544, 1024),
                  (32, 200, 256, 256, 512, 2), (32, 100, 128, 512, 1024, 2), (32, 50, 64, 1024, 2048, 2)]:
        N, H, W, Cin, Cout = shape[:5]
        stride = shape[5] if len(shape) > 5 else 1
        x = torch.randn(N, H, W, Cin, device="cuda").half().relu()
        w = (torch.randn(Cout, 1, 1, Cin, device="cuda") / Cin ** 0.5).half()
        b = torch.randn(Cout, device="cuda")
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        out = torch.empty(N, Ho, Wo, Cout, device="cuda", dtype=torch.float16)
        M = N * Ho * Wo
        nbytes = (M * Cin + Cout * Cin + M * Cout) * 2
        row = []
        for name, pol, wgs in [("r04", 9, 256), ("ring256", 73, 256), ("ring224", 73, 224), ("r04", 9, 256), ("ring256", 73, 256)]:
            hooks.pe_test_set_conv_policy(pol, 1)
            hooks.pe_test_set_ring_wgs(wgs)
            ms = timeit(lambda: L.conv2d_nhwc(x, w, b, kernel=1, stride=stride, relu=True, out=out))
            row.append(f"{name}: {ms:.4f} ms {nbytes / ms / 1e6:5.0f} GB/s")
        hooks.pe_test_set_ring_wgs(256)
        print(f"N{N} {H}x{W} {Cin}->{Cout} s{stride} | " + " | ".join(row), flush=True)
    hooks.pe_test_set_conv_policy(9, 1)


if __name__ == "__main__":
    main()
