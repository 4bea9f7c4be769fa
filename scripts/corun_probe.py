"""Do an MFMA-bound and an HBM-bound kernel gain from running at the same time on two streams?  (a) FPN-level 3x3
(weights-direct) x 4, (b) a res4-like chain of 1x1 launches sized to take about as long, alone and together."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proben_amd  # noqa
from proben_amd import layers as L

torch.manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda")
# (a) p2 3x3, 256 -> 256
xa = rnd(32, 200, 256, 256).half().relu()
wa = L.conv_wd_pack((rnd(256, 3, 3, 256) / 48).half()); ba = rnd(256) * 0.1
oa = torch.empty(32, 200, 256, 256, device="cuda", dtype=torch.float16)
# (b) res4 1x1 pair: 1024 -> 256 and 256 -> 1024 + residual
xb = rnd(32, 50, 64, 1024).half().relu()
w1 = (rnd(256, 1, 1, 1024) / 32).half(); b1 = rnd(256) * 0.1
w3 = (rnd(1024, 1, 1, 256) / 16).half(); b3 = rnd(1024) * 0.1
t = torch.empty(32, 50, 64, 256, device="cuda", dtype=torch.float16)
ob = torch.empty(32, 50, 64, 1024, device="cuda", dtype=torch.float16)


def work_a(n=4):
    for _ in range(n):
        L.conv3x3_wd(xa, wa, ba, 256, relu=True, out=oa)


def work_b(n=30):
    for _ in range(n):
        L.conv2d_nhwc(xb, w1, b1, kernel=1, relu=True, out=t)
        L.conv2d_nhwc(t, w3, b3, kernel=1, relu=True, residual=xb, residual_mode=1, out=ob)


def wall(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(sa):
        work_a()
    with torch.cuda.stream(sb):
        work_b()


ta, tb, tab = wall(work_a), wall(work_b), wall(both)
print(f"3x3 alone {ta:.2f} ms, 1x1 chain alone {tb:.2f} ms, together {tab:.2f} ms (sum {ta + tb:.2f}, max {max(ta, tb):.2f})")
