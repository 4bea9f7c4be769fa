"""Round 5: the ring kernel on the RESIDUAL 1x1 layers (policy bit 8) against conv_igemm2: bits and per-launch time.
Usage (GPU box): python scripts/r05_ring_res.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proben_amd  # noqa: E402,F401
from proben_amd import _lib, layers as L  # noqa: E402
from r05_ring import timeit  # noqa: E402

hooks = _lib.test_hooks()
# N, H, W, Cin, Cout, residual mode
for (N, H, W, Cin, Cout, res) in [(32, 200, 256, 256, 256, 2), (32, 100, 128, 512, 256, 2), (32, 50, 64, 1024, 256, 2), (32, 25, 32, 512, 2048, 1),
                                  (3, 51, 65, 256, 256, 2), (2, 13, 16, 512, 512, 1), (32, 100, 128, 128, 512, 1)]:
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(N, H, W, Cin, generator=g).cuda().half().relu()
    w = (torch.randn(Cout, 1, 1, Cin, generator=g) / Cin ** 0.5).cuda().half()
    b = torch.randn(Cout, generator=g).cuda()
    r = torch.randn(N, H, W, Cout, generator=g).cuda().half() if res == 1 else torch.randn(N, (H + 1) // 2, (W + 1) // 2, Cout, generator=g).cuda().half()
    out = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.float16)
    M = N * H * W
    nbytes = (M * Cin + Cout * Cin + M * Cout + r.numel()) * 2
    row, outs = [], []
    for pol in (73, 329, 73, 329):
        hooks.pe_test_set_conv_policy(pol, 1)
        out.fill_(float("nan"))
        L.conv2d_nhwc(x, w, b, kernel=1, relu=True, residual=r, residual_mode=res, out=out)
        outs.append(out.clone())
        ms = timeit(lambda: L.conv2d_nhwc(x, w, b, kernel=1, relu=True, residual=r, residual_mode=res, out=out))
        row.append(f"{'ring' if pol & 256 else 'igemm2'}: {ms:.4f} ms {nbytes / ms / 1e6:5.0f} GB/s")
    print(f"N{N} {H}x{W} {Cin}->{Cout} res{res} | " + " | ".join(row) + f" | identical={torch.equal(outs[0], outs[1])} nan={int(torch.isnan(outs[1]).sum())}", flush=True)
hooks.pe_test_set_conv_policy(L.DEFAULT_CONV_POLICY, 1)
