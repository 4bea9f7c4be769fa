"""Do the two detector streams co-run better on DISJOINT compute units?  (r03 probe: an MFMA-bound and an HBM-bound wave set
sharing a CU slow each other to 0.87 x of running them one after the other - VMEM issue stalls - while the chip as a whole has
the power / bandwidth for both.)  hipExtStreamCreateWithCUMask gives every detector stream its own CU set; bench.py's step
loop is timed with: the default streams, even / odd CUs, halves of every XCD's CU list, 3:1 splits.
    PYTHONPATH=. python scripts/cumask_ab.py [steps]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import proben_amd  # noqa: E402,F401

hip = ctypes.CDLL("libamdhip64.so")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


cfg = bench.CONFIGS[2]
dev = torch.device("cuda", 0)
models, sds = bench.build_models(cfg, cfg["depth"], dev)
frames = bench.make_frames(cfg, cfg["batch"], 0, dev)


def run(name, streams, stagger=3):
    step = bench.make_step(models, frames, cfg, 1, stagger=stagger)
    if streams is not None:
        step.pipe.streams = streams
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    print(f"{name:60s} {dt * 1e3:7.3f} ms/step  {cfg['batch'] / dt:7.1f} pairs/s", flush=True)


NW = 8   # 256 CUs = 8 x 32-bit words
run("default streams (all CUs shared)", None)
run("even CUs | odd CUs", [masked_stream([0x55555555] * NW), masked_stream([0xAAAAAAAA] * NW)])
run("low half | high half of every 32-CU word", [masked_stream([0x0000FFFF] * NW), masked_stream([0xFFFF0000] * NW)])
run("words 0-3 | words 4-7", [masked_stream([0xFFFFFFFF] * 4 + [0] * 4), masked_stream([0] * 4 + [0xFFFFFFFF] * 4)])
run("3/4 | 1/4 (bits)", [masked_stream([0x77777777] * NW), masked_stream([0x88888888] * NW)])
run("all CUs on both (masked API, full masks)", [masked_stream([0xFFFFFFFF] * NW), masked_stream([0xFFFFFFFF] * NW)])
run("default streams again", None)
