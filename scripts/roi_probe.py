"""ROIAlign timing on FPN-shaped inputs (batch 32, 1000 boxes per image, box sizes log-uniform 16 .. 500 px)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proben_amd  # noqa
from proben_amd import layers as L

N, P, C = 32, 1000, 256
torch.manual_seed(0)
feats = [torch.randn(N, 800 // s, 1024 // s, C, device="cuda").half() for s in (4, 8, 16, 32)]
g = torch.Generator(device="cuda").manual_seed(1)
sz = torch.exp(torch.rand(N, P, 2, device="cuda", generator=g) * (6.2 - 2.77) + 2.77)
ctr = torch.rand(N, P, 2, device="cuda", generator=g) * torch.tensor([1000.0, 800.0], device="cuda")
boxes = torch.cat([(ctr - sz / 2).clamp(min=0), torch.minimum(ctr + sz / 2, torch.tensor([1000.0, 800.0], device="cuda"))], dim=2).contiguous()
cnt = torch.full((N,), P, dtype=torch.int32, device="cuda")


def run():
    return L.roi_align_nhwc(feats, boxes, scales=[1 / 4, 1 / 8, 1 / 16, 1 / 32], pooled=(7, 7), sampling_ratio=0, aligned=True, counts=cnt, per_image=P,
                            num_rois=N * P)


out = run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"roi_align N={N} P={P}: {ms:.4f} ms, output {out.numel() * 2 / 1e6:.0f} MB, checksum {out.float().sum().item():.6e}")
