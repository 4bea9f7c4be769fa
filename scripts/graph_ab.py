"""does capturing a detector forward in a hipGraph buy anything?  One R101-FPN detector, batch 32, full
size: eager launches (what the pipeline does) against a replayed graph of the same ~280 launches, same stream, same buffers.
    python scripts/graph_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proben_amd  # noqa: E402,F401
from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN  # noqa: E402
from proben_amd.synthetic import synthetic_images, synthetic_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = GeneralizedRCNN(DetectorConfig(), synthetic_state_dict(101, 3, 3, seed=1))
frames = torch.from_numpy(synthetic_images(B, seed=10)).cuda()
sizes = [(512, 640)] * B


def fwd():
    return model.forward_batch(frames, out_sizes=sizes, resize_to=(800, 1000))


def wall(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, host * 1e3


for _ in range(3):
    ref = fwd()
eager_ms, eager_host = wall(fwd, 20)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    fwd()                      # warm-up on the capture stream (allocator pools, LDS attributes, size tables)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        out = fwd()
torch.cuda.current_stream().wait_stream(side)
g.replay()
torch.cuda.synchronize()
same = all(torch.equal(out[k][: 1], ref[k][: 1]) for k in ("boxes", "scores", "counts"))
graph_ms, graph_host = wall(g.replay, 20)
print(f"one R101-FPN detector forward, batch {B}: eager {eager_ms:.3f} ms per forward (host enqueue {eager_host:.2f} ms), "
      f"hipGraph replay {graph_ms:.3f} ms (host {graph_host:.3f} ms); results identical: {same}")
