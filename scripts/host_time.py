"""Host enqueue time per step vs device time per step (is the pipeline host-bound?).  python scripts/host_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = bench.parse([])
cfg = bench.CONFIGS[2]
dev = torch.device("cuda", 0)
models, sds = bench.build_models(cfg, 101, dev)
frames = bench.make_frames(cfg, 32, 0, dev)
for stagger in (3, 0):
    step = bench.make_step(models, frames, cfg, 1, stagger=stagger)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(20):
        a = time.perf_counter()
        step()
        host.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / 20
    print(f"stagger {stagger}: host enqueue {sum(host) / len(host) * 1e3:.2f} ms/step (max {max(host) * 1e3:.2f}), wall {total * 1e3:.2f} ms/step", flush=True)
# ---- one detector forward from an idle GPU: pure host enqueue time vs device time ----
m = models[0]
for _ in range(2):
    m.forward_batch(frames[0], out_sizes=[(512, 640)] * 32, resize_to=(800, 1000))
torch.cuda.synchronize()
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a = time.perf_counter()
    e0.record()
    m.forward_batch(frames[0], out_sizes=[(512, 640)] * 32, resize_to=(800, 1000))
    e1.record()
    h = time.perf_counter() - a
    torch.cuda.synchronize()
    print(f"one detector forward from idle: host enqueue {h * 1e3:.2f} ms, device {e0.elapsed_time(e1):.2f} ms", flush=True)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    m.forward_batch(frames[0], out_sizes=[(512, 640)] * 32, resize_to=(800, 1000))
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
