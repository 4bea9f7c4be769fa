import torch, proben_amd
from proben_amd import layers as L, weights as WT
x = torch.randn(32, 800, 1024, 4).cuda().half(); x[..., 3] = 0
w = torch.randn(64, 3, 7, 7) / 12; b = torch.randn(64).cuda()
wf, wu = WT.pack_stem_fused(w).cuda(), WT._pack_stem(w).cuda()
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("fused ms", t(lambda: L.stem_conv_pool(x, wf, b)))
print("unfused ms", t(lambda: L.maxpool3x3s2_nhwc(L.conv2d_nhwc(x, wu, b, kernel=7, stride=2, relu=True))))
