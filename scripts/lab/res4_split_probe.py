"""Lab only (round 6): does the res4 stage tile better as MORE, SMALLER chains?  A batch-32 tail is 800 tiles of 128 pixels on 512 workgroup
slots = 1.56 rounds (the second round is 56 % full) and the two detectors' tails, co-running, take exactly twice one tail's time
(profiles/r06_kernel_stats_two_streams.csv: 385 us against 192 us serial): the holes are not filled.  Here: the stride-1 blocks of res4
(22 x [ring conv1 1024 -> 256, fused tail]) for TWO detectors as 2 chains of 32 images (what the pipeline does), as 4 chains of 16 images and
as 8 chains of 8, each chain on its own stream.  Same kernels, same bits per image (the batch size never enters a result)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import proben_amd  # noqa: E402,F401
from proben_amd import layers as L  # noqa: E402

NB = 22
H, W, C, CT = 50, 64, 256, 1024


def make_chain(n_img, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    w1 = (rnd(C, 1, 1, CT) / CT ** 0.5).half(); b1 = rnd(C) * 0.1
    w2 = (rnd(C, 3, 3, C) / (C * 9) ** 0.5).half(); b2 = rnd(C) * 0.1
    w3 = (rnd(CT, C) / C ** 0.5).half(); b3 = rnd(CT) * 0.1
    return dict(n=n_img, w1=w1, b1=b1, pk2=L.conv_wd_pack(w2), b2=b2, pk3=L.conv_wd_pack_tail(w3), b3=b3,
                x=[rnd(n_img, H, W, CT).half().relu() for _ in range(2)], t=torch.empty(n_img, H, W, C, device="cuda", dtype=torch.float16))


def run_chain(c):
    x = c["x"]
    for b in range(NB):
        L.conv2d_nhwc(x[b & 1], c["w1"], c["b1"], kernel=1, relu=True, out=c["t"])
        L.bottleneck_tail_wd(c["t"], c["pk2"], c["b2"], c["pk3"], c["b3"], x[b & 1], CT, out=x[(b + 1) & 1])


def measure(split, reps=6, stagger=0):
    chains = [make_chain(32 // split, 10 * d + s) for d in range(2) for s in range(split)]
    streams = [torch.cuda.Stream() for _ in chains]
    def step():
        main = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(main)
        for c, st in zip(chains, streams):
            with torch.cuda.stream(st):
                run_chain(c)
        for st in streams:
            main.wait_stream(st)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if __name__ == "__main__":
    for rnd_i in range(2):
        for split in (1, 2, 4):
            print("round %d: 2 detectors x %d chain(s) of %2d images: %.3f ms per res4 stage pair (22 blocks each)" % (rnd_i, split, 32 // split, measure(split)), flush=True)
