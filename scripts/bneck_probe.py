"""res2 at batch 32 (200 x 256 pixels): the fused 64-wide bottleneck chain against the launches it replaces."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proben_amd  # noqa
from proben_amd import layers as L


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


N, H, W = 32, 200, 256
if len(sys.argv) > 1:
    N, H, W = (int(v) for v in sys.argv[1:4])
torch.manual_seed(0)
M = N * H * W
rnd = lambda *s: torch.randn(*s, device="cuda")
t1 = rnd(N, H, W, 64).half().relu()
x = rnd(N, H, W, 256).half().relu()
s = rnd(N, H, W, 64).half().relu()
w2 = (rnd(64, 3, 3, 64) / 24).half(); b2 = rnd(64) * 0.1
w3 = (rnd(256, 1, 1, 64) / 8).half(); b3 = rnd(256) * 0.1
wsc = (rnd(256, 1, 1, 64) / 8).half(); bsc = rnd(256) * 0.1
w1n = (rnd(64, 1, 1, 256) / 16).half(); b1n = rnd(64) * 0.1
out = torch.empty(N, H, W, 256, device="cuda", dtype=torch.float16)
t2 = torch.empty(N, H, W, 64, device="cuda", dtype=torch.float16)
t1n = torch.empty(N, H, W, 64, device="cuda", dtype=torch.float16)
sc_out = torch.empty(N, H, W, 256, device="cuda", dtype=torch.float16)
for sc, nxt in ((False, True), (False, False), (True, True)):
    pk = L.bneck64_pack(w2, w3.reshape(256, 64), wsc.reshape(256, 64) if sc else None, w1n.reshape(64, 256) if nxt else None)
    fused = timed(lambda: L.bneck64(t1, s if sc else x, pk, b2, b3, bsc if sc else None, b1n if nxt else None, out=out, t1_next=t1n if nxt else None))
    parts = {"3x3": timed(lambda: L.conv2d_nhwc(t1, w2, b2, kernel=3, relu=True, out=t2))}
    if sc:
        parts["shortcut 1x1"] = timed(lambda: L.conv2d_nhwc(s, wsc, bsc, kernel=1, out=sc_out))
    parts["conv3 + shortcut"] = timed(lambda: L.conv2d_nhwc(t2, w3, b3, kernel=1, relu=True, residual=sc_out if sc else x, residual_mode=1, out=out))
    if nxt:
        parts["next conv1"] = timed(lambda: L.conv2d_nhwc(out, w1n, b1n, kernel=1, relu=True, out=t1n))
    mb = M * 2 * (64 + (64 if sc else 256) + 256 + (64 if nxt else 0)) / 1e6
    print(f"shortcut conv {int(sc)}, next conv1 {int(nxt)}: fused {fused:.4f} ms ({mb:.0f} MB -> {mb / fused / 1e3:.2f} TB/s) | unfused {sum(parts.values()):.4f} ms: "
          + ", ".join(f"{k} {v:.4f}" for k, v in parts.items()))
