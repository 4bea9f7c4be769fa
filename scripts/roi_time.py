"""Times pe_roi_align_nhwc on the bench shape (32 images x 1000 proposals, p2..p5 of 800x1024, 256 channels)."""
import torch
import proben_amd  # noqa: F401
from proben_amd import layers as L
torch.manual_seed(0)
N, P = 32, 1000
feats = [torch.randn(N, 800 // s, 1024 // s, 256, device="cuda").half() for s in (4, 8, 16, 32)]
cx, cy = torch.rand(N, P, device="cuda") * 1000, torch.rand(N, P, device="cuda") * 800
sz = torch.exp(torch.rand(N, P, 2, device="cuda") * 3.5 + 2.5)   # 12 .. 400 px
boxes = torch.stack([(cx - sz[..., 0] / 2).clamp(0, 1000), (cy - sz[..., 1] / 2).clamp(0, 800),
                     (cx + sz[..., 0] / 2).clamp(0, 1000), (cy + sz[..., 1] / 2).clamp(0, 800)], -1).contiguous()
cnt = torch.full((N,), P, dtype=torch.int32, device="cuda")
f = lambda: L.roi_align_nhwc(feats, boxes, scales=[1 / 4, 1 / 8, 1 / 16, 1 / 32], pooled=(7, 7), sampling_ratio=0, aligned=True,
                             counts=cnt, per_image=P, num_rois=N * P)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    out = f()
e1.record(); torch.cuda.synchronize()
print("roi_align ms", e0.elapsed_time(e1) / 10, "checksum", float(out.float().abs().mean()))
