"""Round 6, VERDICT r05 item 1: what would a NEXT epilogue in the fused res4 tail cost - the next block's conv1 (1024 -> 256) multiplied
into the tail's conv3 phase, so that the 44 ring-kernel launches per step that re-read the 210 MB block output disappear?
Registers and LDS do not close at two waves per SIMD, so this PRICES the idea before anything is built: the LAB library's
`pe_lab_bottleneck_tail_next_pricing` (conv_wd.h, ABL & 8; results wrong) runs every conv3 chunk's K-loop twice - the second pass is the
MFMA / weight-record / fragment-read diet of the NEXT product (same 33.5 M MAC per tile, same 128 KiB of L2 records per wave) - and
stores 64 KiB more whole lines per tile (the 256-channel conv1 output), as if the 128 extra accumulators and the 64 KiB exchange tile
were free.  Against it: the shipped tail and the ring kernel's conv1 launch it would replace.
    python -m proben_amd.build --lab && python scripts/lab/r06_tail_next_pricing.py      (GPU box)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import proben_amd  # noqa: E402,F401
from proben_amd import _lib, layers as L  # noqa: E402

_lib.LIB_PATH = _lib.LIB_PATH.replace(".so", "_lab.so")
assert os.path.exists(_lib.LIB_PATH), "build the lab library first: python -m proben_amd.build --lab"


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    lib = _lib.lib()
    fn = lib.pe_lab_bottleneck_tail_next_pricing
    fn.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_int32] * 5 + [ctypes.c_void_p]
    fn.restype = ctypes.c_int32
    N, H, W, C, CT = 32, 50, 64, 256, 1024
    torch.manual_seed(0)
    x = torch.randn(N, H, W, C, device="cuda").half().relu()
    w2 = (torch.randn(C, 3, 3, C, device="cuda") / (C * 9) ** 0.5).half()
    b2 = torch.randn(C, device="cuda") * 0.1
    w3 = (torch.randn(CT, 1, 1, C, device="cuda") / C ** 0.5).half()
    b3 = torch.randn(CT, device="cuda") * 0.1
    w1 = (torch.randn(C, 1, 1, CT, device="cuda") / CT ** 0.5).half()
    b1 = torch.randn(C, device="cuda") * 0.1
    res = torch.randn(N, H, W, CT, device="cuda").half().relu()
    out = torch.empty(N, H, W, CT, device="cuda", dtype=torch.float16)
    t1n = torch.empty(N, H, W, C, device="cuda", dtype=torch.float16)
    pk2 = L.conv_wd_pack(w2)
    pk3 = L.conv_wd_pack_tail(w3.reshape(CT, C))
    tail = lambda: L.bottleneck_tail_wd(x, pk2, b2, pk3, b3, res, CT, out=out)
    conv1 = lambda: L.conv2d_nhwc(out, w1, b1, kernel=1, relu=True, out=t1n)
    priced = lambda: _lib.check(fn(_lib.ptr(x), _lib.ptr(pk2), _lib.ptr(b2), _lib.ptr(pk3), _lib.ptr(b3), _lib.ptr(res), _lib.ptr(out), _lib.ptr(t1n),
                                   N, H, W, C, CT, _lib.stream()), "pricing")
    both = lambda: (tail(), conv1())
    for rep in range(3):
        a, b, c, d = timed(tail), timed(conv1), timed(both), timed(priced)
        print(f"rep {rep}: shipped tail {a * 1e3:6.1f} us   ring conv1 1024->256 {b * 1e3:5.1f} us   tail then conv1 {c * 1e3:6.1f} us   "
              f"tail + NEXT's work on garbage (free registers / LDS) {d * 1e3:6.1f} us   -> a fused block would save at most {(c - d) * 1e3:5.1f} us of {c * 1e3:5.1f}", flush=True)


if __name__ == "__main__":
    main()
