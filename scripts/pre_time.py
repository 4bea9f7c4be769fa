"""Times the two pre-processing kernels on the bench shape (32 uint8 frames 512x640 -> 800x1000, padded 800x1024)."""
import torch
import proben_amd  # noqa: F401
from proben_amd import layers as L
src = torch.randint(0, 256, (32, 512, 640, 3), dtype=torch.uint8, device="cuda")
dst = torch.empty((32, 800, 1024, 4), dtype=torch.float16, device="cuda")
kw = dict(ch0=0, nch=3, flip_rgb=False, dst_hw=(800, 1000), mean=[103.53, 116.28, 123.675], std=[1.0, 1.0, 1.0])
def t(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("pillow-exact ms", t(lambda: L.preprocess_pack_pil_u8(src, dst, **kw)))
print("float bilinear ms", t(lambda: L.preprocess_pack_batch(src, dst, src_kind=0, **kw)))
