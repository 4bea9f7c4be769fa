"""Fixed vs per-K-step cost of the 1x1 kernel: M = 102400 (res4 at batch 32), N in {256, 1024}, K swept."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proben_amd  # noqa
from proben_amd import layers as L
from ablate_conv import timeit
M = 32 * 50 * 64
for N in (1024, 256):
    for K in (64, 128, 256, 512, 1024, 2048):
        x = torch.randn(32, 50, 64, K, device="cuda").half()
        w = (torch.randn(N, 1, 1, K, device="cuda") / K ** 0.5).half()
        b = torch.randn(N, device="cuda")
        out = torch.empty(32, 50, 64, N, device="cuda", dtype=torch.float16)
        ms = timeit(lambda: L.conv2d_nhwc(x, w, b, kernel=1, relu=True, out=out))
        by = (M * K + M * N + N * K) * 2
        print(f"N{N} K{K}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.0f} TF  {by/ms/1e9:.2f} TB/s", flush=True)
