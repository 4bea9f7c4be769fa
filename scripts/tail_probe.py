"""Where does the fused bottleneck tail spend its time?  res4 shape, batch 32: the fused launch with / without the shortcut,
against its two halves as separate launches (weights-direct 3x3, LDS-DMA 1x1)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proben_amd  # noqa
from proben_amd import layers as L


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


N, H, W, C, CT = 32, 50, 64, 256, 1024
torch.manual_seed(0)
x = torch.randn(N, H, W, C, device="cuda").half().relu()
w2 = (torch.randn(C, 3, 3, C, device="cuda") / (C * 9) ** 0.5).half()
b2 = torch.randn(C, device="cuda") * 0.1
w3 = (torch.randn(CT, 1, 1, C, device="cuda") / C ** 0.5).half()
b3 = torch.randn(CT, device="cuda") * 0.1
res = torch.randn(N, H, W, CT, device="cuda").half().relu()
out = torch.empty(N, H, W, CT, device="cuda", dtype=torch.float16)
t = torch.empty(N, H, W, C, device="cuda", dtype=torch.float16)
pk2 = L.conv_wd_pack(w2)
pk3 = L.conv_wd_pack_tail(w3.reshape(CT, C))
flops3, flops1 = 2.0 * N * H * W * C * C * 9, 2.0 * N * H * W * C * CT
rows = [
    ("fused tail, shortcut", lambda: L.bottleneck_tail_wd(x, pk2, b2, pk3, b3, res, CT, out=out), flops3 + flops1),
    ("fused tail, no shortcut", lambda: L.bottleneck_tail_wd(x, pk2, b2, pk3, b3, None, CT, out=out), flops3 + flops1),
    ("3x3 weights-direct alone", lambda: L.conv3x3_wd(x, pk2, b2, C, relu=True, out=t), flops3),
    ("1x1 LDS-DMA + shortcut", lambda: L.conv2d_nhwc(t, w3, b3, kernel=1, relu=True, residual=res, residual_mode=1, out=out), flops1),
    ("1x1 LDS-DMA, no shortcut", lambda: L.conv2d_nhwc(t, w3, b3, kernel=1, relu=True, out=out), flops1),
]
for name, fn, fl in rows:
    ms = timed(fn)
    print(f"{name:28s} {ms:.4f} ms  {fl / ms / 1e9:7.0f} TFLOP/s")
