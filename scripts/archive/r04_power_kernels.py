"""Board power and shader clock per hot kernel (1 x MI355X, power cap 1 400 W): each launch replayed back-to-back for ~2.5 s while a
thread polls `rocm-smi --showpower --showclocks --json`.  Answers whether the MFMA kernels sit on the power cap.
    python scripts/r04_power_kernels.py > gpurun_out/r04_power_kernels.txt"""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proben_amd  # noqa: E402,F401
from proben_amd import _lib, layers as L  # noqa: E402


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.rows = False, []

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
                d = json.loads(out[out.index("{"):])
                c = next(v for v in d.values() if isinstance(v, dict))
                num = lambda s: float(re.search(r"[-+]?\d+(\.\d+)?", str(s)).group(0))
                self.rows.append((num(next(v for k, v in c.items() if "Power" in k)), num(c["sclk clock speed:"])))
            except Exception:
                pass


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else float("nan")


def measure(name, fn, flops=0.0, nbytes=0.0, seconds=2.5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / 20
    reps = max(20, int(seconds * 1e3 / per))
    s = Sampler()
    s.start()
    time.sleep(0.3)
    s.rows.clear()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    s.stop = True
    ms = e0.elapsed_time(e1) / reps
    rows = s.rows[1:-1] if len(s.rows) > 4 else s.rows
    print(f"{name:58s} {ms:8.4f} ms  {flops / ms / 1e9:7.0f} TFLOP/s  {nbytes / ms / 1e6:6.0f} GB/s  {med([r[0] for r in rows]):6.0f} W  "
          f"{med([r[1] for r in rows]):5.0f} MHz  ({len(rows)} samples)", flush=True)


def main():
    hooks = _lib.test_hooks()
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g)
    print("idle:", end=" ")
    s = Sampler(); s.start(); time.sleep(2.0); s.stop = True
    print(f"{med([r[0] for r in s.rows]):.0f} W, {med([r[1] for r in s.rows]):.0f} MHz")
    # pure 3x3, FPN p2 output conv
    w = (rnd(256, 3, 3, 256) / 48).half().cuda()
    pk = L.conv_wd_pack(w)
    b = rnd(256).cuda()
    fl = 2.0 * 32 * 200 * 256 * 256 * 2304
    for label, x in (("activations = relu(N(0,1))", rnd(32, 200, 256, 256).relu().half().cuda()), ("activations = 0", torch.zeros(32, 200, 256, 256).half().cuda())):
        for mode, kn in ((1, "wd9 (1 wave / SIMD, AGPR accumulators)"), (0, "two-wave conv_wd.h")):
            hooks.pe_test_set_wd9_mode(mode)
            out = torch.empty(32, 200, 256, 256, dtype=torch.float16, device="cuda")
            measure(f"3x3 256->256 N32 200x256, {kn}, {label}"[:58], lambda: L.conv3x3_wd(x, pk, b, 256, relu=True, out=out), fl, x.numel() * 4.0)
    del x, out
    # res4 fused tail
    x = rnd(32, 50, 64, 256).relu().half().cuda()
    r = rnd(32, 50, 64, 1024).relu().half().cuda()
    p3 = L.conv_wd_pack_tail((rnd(1024, 256) / 16).half().cuda())
    b3 = rnd(1024).cuda()
    out = torch.empty(32, 50, 64, 1024, dtype=torch.float16, device="cuda")
    fl = 2.0 * 32 * 50 * 64 * (256 * 2304 + 1024 * 256)
    for mode, kn in ((1, "two-wave"), (5, "wd9 tail")):
        hooks.pe_test_set_wd9_mode(mode)
        measure(f"res4 tail N32 50x64, {kn}", lambda: L.bottleneck_tail_wd(x, pk, b, p3, b3, r, 1024, out=out), fl, 473.6e6)
    hooks.pe_test_set_wd9_mode(1)
    # 1x1 convs
    w1 = (rnd(256, 1, 1, 1024) / 32).half().cuda()
    o1 = torch.empty(32, 50, 64, 256, dtype=torch.float16, device="cuda")
    measure("1x1 1024->256 N32 50x64 (res4 conv1)", lambda: L.conv2d_nhwc(r, w1, b, kernel=1, relu=True, out=o1), 2.0 * 102400 * 1024 * 256, 262e6)
    x3 = rnd(32, 100, 128, 128).relu().half().cuda()
    r3 = rnd(32, 100, 128, 512).relu().half().cuda()
    w3 = (rnd(512, 1, 1, 128) / 11).half().cuda()
    bb = rnd(512).cuda()
    o3 = torch.empty(32, 100, 128, 512, dtype=torch.float16, device="cuda")
    measure("1x1 128->512 + shortcut N32 100x128 (res3 conv3)", lambda: L.conv2d_nhwc(x3, w3, bb, kernel=1, relu=True, residual=r3, residual_mode=1, out=o3),
            2.0 * 409600 * 128 * 512, 943e6)
    # fc1
    xa = rnd(32000, 12544).relu().half().cuda()
    wf = (rnd(1024, 12544) / 112).half().cuda()
    bf = rnd(1024).cuda()
    measure("fc1 12544->1024, 32000 rows", lambda: L.linear_f16(xa, wf, bf, relu=True), 2.0 * 32000 * 12544 * 1024, 894e6)
    del xa
    # ROIAlign
    feats = [rnd(32, 200 >> l, 256 >> l, 256).half().cuda() for l in range(4)]
    ctr = torch.rand(32, 1000, 2, generator=g) * torch.tensor([1000.0, 780.0])
    wh = torch.exp(torch.rand(32, 1000, 2, generator=g) * 3.5 + 2.5)
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=2).cuda()
    cnt = torch.full((32,), 1000, dtype=torch.int32).cuda()
    po = torch.empty(32000, 7, 7, 256, dtype=torch.float16, device="cuda")
    measure("ROIAlign 32 x 1000 proposals, 4 levels", lambda: L.roi_align_nhwc(feats, boxes, scales=[1 / 4, 1 / 8, 1 / 16, 1 / 32], pooled=(7, 7), counts=cnt, per_image=1000, out=po),
            0.0, 803e6)


if __name__ == "__main__":
    main()
