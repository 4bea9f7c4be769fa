"""Practical HBM rates on this MI355X for the traffic mix of the res4 1x1 convs, next to the conv kernel itself."""
import torch
import proben_amd  # noqa: F401
from proben_amd import layers as L


def t(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


M = 32 * 50 * 64
a = torch.randn(M, 1024, device="cuda").half()
b = torch.empty_like(a)
y = torch.randn(M, 256, device="cuda").half()
s = t(lambda: b.copy_(a)); print(f"copy 210->210 MB: {s*1e3:.4f} ms {2*a.numel()*2/s/1e12:.2f} TB/s")
s = t(lambda: torch.add(a, b, out=b)); print(f"add 2x210->210 MB: {s*1e3:.4f} ms {3*a.numel()*2/s/1e12:.2f} TB/s")
s = t(lambda: a.float().sum() if False else torch.sum(a, dtype=torch.float32)); print(f"read 210 MB: {s*1e3:.4f} ms {a.numel()*2/s/1e12:.2f} TB/s")
big = torch.randn(32 * 200 * 256, 256, device="cuda").half(); big2 = torch.empty_like(big)
s = t(lambda: big2.copy_(big)); print(f"copy 839->839 MB: {s*1e3:.4f} ms {2*big.numel()*2/s/1e12:.2f} TB/s")
s = t(lambda: torch.add(big, big2, out=big2)); print(f"add 2x839->839 MB: {s*1e3:.4f} ms {3*big.numel()*2/s/1e12:.2f} TB/s")
del big, big2
w3 = (torch.randn(1024, 1, 1, 256, device="cuda") / 16).half(); b3 = torch.randn(1024, device="cuda")
w1 = (torch.randn(256, 1, 1, 1024, device="cuda") / 32).half(); b1 = torch.randn(256, device="cuda")
x4 = y.view(32, 50, 64, 256); r4 = a.view(32, 50, 64, 1024); o4 = b.view(32, 50, 64, 1024); o1 = torch.empty_like(x4)
for name, f, by, fl in [
    ("conv3 256->1024 +res", lambda: L.conv2d_nhwc(x4, w3, b3, kernel=1, relu=True, residual=r4, residual_mode=1, out=o4), (M*256 + 2*M*1024)*2, 2.0*M*256*1024),
    ("conv3 256->1024 no res", lambda: L.conv2d_nhwc(x4, w3, b3, kernel=1, relu=True, out=o4), (M*256 + M*1024)*2, 2.0*M*256*1024),
    ("conv1 1024->256", lambda: L.conv2d_nhwc(r4, w1, b1, kernel=1, relu=True, out=o1), (M*1024 + M*256)*2, 2.0*M*256*1024)]:
    s = t(f); print(f"{name}: {s*1e3:.4f} ms {by/s/1e12:.2f} TB/s {fl/s/1e12:.0f} TF")
