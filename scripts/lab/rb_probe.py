"""Lab only: res3's 3x3 (N32 100x128, 128 -> 128) on the kw-reuse kernel's two tile heights (policy bit 0), the generic per-tap kernel and - for scale - the same flops as one 256 -> 256 launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import proben_amd  # noqa
from proben_amd import _lib, layers as L
if len(sys.argv) > 1 and sys.argv[1] != "product":
    _lib.LIB_PATH = _lib.LIB_PATH.replace(".so", "_%s.so" % sys.argv[1])
def timed(fn, reps=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N, H, W, C = 32, 100, 128, 128
x = torch.randn(N, H, W, C, device="cuda").half().relu()
w = (torch.randn(C, 3, 3, C, device="cuda") / (C * 9) ** 0.5).half(); b = torch.randn(C, device="cuda") * 0.1
out = torch.empty(N, H, W, C, device="cuda", dtype=torch.float16)
import hashlib
for name, pol, reuse in (("rb<256,128>", 329, 1), ("rb<128,128>", 328, 1), ("generic per-tap", 329, 0)):
    _lib.test_hooks().pe_test_set_conv_policy(pol, reuse)
    t = min(timed(lambda: L.conv2d_nhwc(x, w, b, kernel=3, relu=True, out=out)) for _ in range(3))
    torch.cuda.synchronize()
    print("%-8s %-18s %.1f us  %.0f TFLOP/s  %s" % (sys.argv[1] if len(sys.argv) > 1 else "product", name, t, 2 * N * H * W * C * C * 9 / t / 1e6, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:10]))
