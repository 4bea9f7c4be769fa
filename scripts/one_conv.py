"""Launches a handful of representative conv shapes a few times each (PMC passes: rocprofv3 --pmc ... -- python scripts/one_conv.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proben_amd  # noqa
from proben_amd import layers as L
for (N, H, W, Cin, Cout, k, res) in [(32, 200, 256, 256, 256, 3, 0), (32, 50, 64, 256, 256, 3, 0), (32, 50, 64, 256, 1024, 1, 1), (32, 50, 64, 1024, 256, 1, 0),
                                     (32, 200, 256, 64, 256, 1, 1)]:
    x = torch.randn(N, H, W, Cin, device="cuda").half().relu()
    w = (torch.randn(Cout, k, k, Cin, device="cuda") / (Cin * k * k) ** 0.5).half()
    b = torch.randn(Cout, device="cuda")
    r = torch.randn(N, H, W, Cout, device="cuda").half() if res else None
    out = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.float16)
    pk = L.conv_wd_pack(w) if k == 3 and L.conv_wd_supported(3, 1, H, W, Cin, Cout) else None
    for _ in range(4):
        if pk is not None:
            L.conv3x3_wd(x, pk, b, Cout, relu=True, out=out)
        else:
            L.conv2d_nhwc(x, w, b, kernel=k, relu=True, residual=r, residual_mode=res, out=out)
    torch.cuda.synchronize()
