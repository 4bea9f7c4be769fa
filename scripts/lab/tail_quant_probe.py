"""Lab only (round 6): the fused res4 tail at tile counts around the 512 workgroup slots (128-pixel tiles, 25 per image):
20 images = 500 tiles (one round, 98 % full), 32 = 800 (1.56 rounds), 40 = 1000 (1.95 rounds), 41 = 1025 (2.002)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import proben_amd  # noqa
from proben_amd import _lib, layers as L
if len(sys.argv) > 1 and sys.argv[1] != "product":
    _lib.LIB_PATH = _lib.LIB_PATH.replace(".so", "_%s.so" % sys.argv[1])
H, W, C, CT = 50, 64, 256, 1024
rnd = lambda *s: torch.randn(*s, device="cuda")
w2 = (rnd(C, 3, 3, C) / (C * 9) ** 0.5).half(); b2 = rnd(C) * 0.1
w3 = (rnd(CT, C) / C ** 0.5).half(); b3 = rnd(CT) * 0.1
pk2, pk3 = L.conv_wd_pack(w2), L.conv_wd_pack_tail(w3)
REPS = int(os.environ.get('PE_REPS', '40'))
def timed(fn, reps=REPS):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for N in ([int(v) for v in sys.argv[2:]] or (10, 20, 21, 32, 40, 41, 61)):
    x = rnd(N, H, W, C).half().relu(); res = rnd(N, H, W, CT).half().relu()
    out = torch.empty(N, H, W, CT, device="cuda", dtype=torch.float16)
    t = min(timed(lambda: L.bottleneck_tail_wd(x, pk2, b2, pk3, b3, res, CT, out=out)) for _ in range(3 if REPS >= 40 else 1))
    print("%s N=%2d tiles=%4d rounds=%.3f  %.1f us  = %.1f us per 512 tiles, %.0f TFLOP/s" % (sys.argv[1] if len(sys.argv) > 1 else "product", N, N * 25, N * 25 / 512, t, t / (N * 25 / 512), N * 3200 * 2 * (2304 * 256 + 256 * 1024) / t / 1e6))
