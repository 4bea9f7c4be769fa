"""Does the 256 MB Infinity Cache help the res4 bottleneck chain when the batch is split?  Times 6 consecutive res4
blocks (conv1 1x1 1024->256, conv2 3x3 256->256, conv3 1x1 256->1024 + residual + ReLU) at several batch sizes and
prints ms per image.  Usage (GPU box): python scripts/mall_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proben_amd  # noqa: E402,F401
from proben_amd import layers as L  # noqa: E402


def main():
    g = torch.Generator(device="cpu").manual_seed(0)
    w1 = (torch.randn(256, 1, 1, 1024, generator=g) / 32).cuda().half()
    w2 = (torch.randn(256, 3, 3, 256, generator=g) / 48).cuda().half()
    w3 = (torch.randn(1024, 1, 1, 256, generator=g) / 16 * 0.3).cuda().half()
    b1, b2, b3 = torch.zeros(256).cuda(), torch.zeros(256).cuda(), torch.zeros(1024).cuda()
    p2 = L.conv_wd_pack(w2)
    for N in (4, 8, 16, 32, 64):
        x = torch.randn(N, 50, 64, 1024, generator=g).cuda().half().relu()
        bufs = [torch.empty_like(x) for _ in range(2)]
        t1 = torch.empty(N, 50, 64, 256, device="cuda", dtype=torch.float16)
        t2 = torch.empty_like(t1)

        def chain():
            cur = x
            for i in range(6):
                L.conv2d_nhwc(cur, w1, b1, kernel=1, relu=True, out=t1)
                L.conv3x3_wd(t1, p2, b2, 256, relu=True, out=t2)
                out = bufs[i & 1]
                L.conv2d_nhwc(t2, w3, b3, kernel=1, relu=True, residual=cur, residual_mode=1, out=out)
                cur = out
        for _ in range(3):
            chain()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 10
        for _ in range(reps):
            chain()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"N={N:3d}: {ms:8.3f} ms per 6 blocks, {ms / N * 1e3:8.1f} us per image, block output {N * 50 * 64 * 1024 * 2 / 1e6:6.1f} MB", flush=True)


if __name__ == "__main__":
    main()
