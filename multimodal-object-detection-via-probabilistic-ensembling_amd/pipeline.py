"""Frame-pair pipeline: N detectors over the same batch of aligned frames + ProbEn, device to device.

This is the reference's two-stage, file-based flow (demo_FLIR_save_predictions.py per detector -> JSON ->
demo_probEn.py) as ONE stream of launches: every detector's forward is enqueued on its own HIP stream (their
small late-stage kernels - res5, p5/p6, heads, NMS - do not fill 256 CUs alone and overlap with the other
detector's work), the main stream waits for all of them and runs the packing + ProbEn kernels.
(Measured r01, batch 32 pairs: 2 streams 680-700 pairs/s vs 622 serial; splitting each detector's batch over more
streams is slower - 651 at 2 sub-batches, 612 at 4 - smaller launches tile worse.)"""
import torch

from . import fusion as F
from . import layers as L


class FramePairPipeline:
    """concurrent: one HIP stream per detector.
    staggered (throughput mode, two detectors): detector 1's stream starts a batch only when detector 0 has finished
    its res`stagger_stage` stage, and nothing makes detector 0 wait for the fusion of the previous batch - in steady
    state the two identical networks run half a forward pass apart, so one's MFMA-bound p2/p3 3x3 convolutions
    share the chip with the other's HBM/latency-bound res4 1x1 chain instead of with their own kind.  The call
    returns with work still in flight on the side streams: use the results after `wait()` (or a device
    synchronisation); inputs must not be modified before that either."""

    def __init__(self, models, score_fusion="probEn", box_fusion="v-avg", max_class=2, concurrent=True,
                 staggered=False, stagger_stage=4, fuse=True):
        self.models = list(models)
        self.fuse = fuse and len(self.models) > 1   # a single detector has nothing to fuse (configs[1])
        self.method = (score_fusion, box_fusion)
        self.max_class = max_class
        self.concurrent = concurrent and len(self.models) > 1
        self.staggered = staggered and self.concurrent and len(self.models) == 2
        self.stagger_stage = stagger_stage
        self.streams = [torch.cuda.Stream() for _ in self.models] if self.concurrent else None

    def wait(self, results=None):
        """Make the current stream wait for everything the pipeline has enqueued (staggered mode).  Pass the
        (dets, fused) a call returned to also tell the caching allocator that those tensors - allocated on the side
        streams - are now used on the current stream."""
        if self.streams:
            main = torch.cuda.current_stream()
            for st in self.streams:
                main.wait_stream(st)
            if results is not None:
                dets, fused = results
                for d in list(dets) + ([fused] if fused is not None else []):
                    for v in d.values():
                        if isinstance(v, torch.Tensor) and v.is_cuda:
                            v.record_stream(main)

    @staticmethod
    def _record(frames, stream):
        """Inputs are allocated on the caller's stream and read on a side stream: without record_stream the caching
        allocator may hand their memory to the next batch while the side stream is still reading it."""
        for f in (frames if isinstance(frames, (list, tuple)) else [frames]):
            if isinstance(f, torch.Tensor) and f.is_cuda:
                f.record_stream(stream)

    @torch.no_grad()
    def _call_staggered(self, frames_per_detector, out_sizes, resize_to):
        sa, sb = self.streams
        ma, mb = self.models
        main = torch.cuda.current_stream()
        ev_in = torch.cuda.Event()
        ev_in.record(main)              # inputs are ready; NOT a wait for earlier pipeline work (that lives on sa / sb)
        ev_mid, ev_a = torch.cuda.Event(), torch.cuda.Event()
        sa.wait_event(ev_in)
        self._record(frames_per_detector[0], sa)
        self._record(frames_per_detector[1], sb)
        with torch.cuda.stream(sa):
            ma.stage_hook = lambda stage: ev_mid.record(sa) if stage == self.stagger_stage else None
            try:
                det_a = ma.forward_batch(frames_per_detector[0], out_sizes=out_sizes, resize_to=resize_to)
            finally:
                ma.stage_hook = None
            ev_a.record(sa)
        sb.wait_event(ev_in)
        sb.wait_event(ev_mid)
        with torch.cuda.stream(sb):
            det_b = mb.forward_batch(frames_per_detector[1], out_sizes=out_sizes, resize_to=resize_to)
            sb.wait_event(ev_a)
            for v in det_a.values():    # detector 0's results were allocated on sa and are read on sb
                if isinstance(v, torch.Tensor):
                    v.record_stream(sb)
            fused = F.fuse_detections([det_a, det_b], self.method[0], self.method[1], max_class=self.max_class)
        return [det_a, det_b], fused

    @torch.no_grad()
    def __call__(self, frames_per_detector, out_sizes, resize_to):
        """frames_per_detector[d]: the batch for detector d ([B,H,W,C] uint8/float tensor or list of tensors).
        Returns (list of per-detector result dicts, fused dict of fusion.fuse_detections)."""
        return self._run(frames_per_detector, out_sizes, resize_to)

    def _run(self, frames_per_detector, out_sizes, resize_to):
        if self.staggered:
            return self._call_staggered(frames_per_detector, out_sizes, resize_to)
        if not self.concurrent:
            dets = [m.forward_batch(fr, out_sizes=out_sizes, resize_to=resize_to)
                    for m, fr in zip(self.models, frames_per_detector)]
        else:
            main = torch.cuda.current_stream()
            dets = []
            for m, fr, st in zip(self.models, frames_per_detector, self.streams):
                st.wait_stream(main)
                self._record(fr, st)
                with torch.cuda.stream(st):
                    dets.append(m.forward_batch(fr, out_sizes=out_sizes, resize_to=resize_to))
            for st in self.streams:
                main.wait_stream(st)
            for d in dets:               # allocated on a side stream, consumed on the main stream from here on
                for v in d.values():
                    if isinstance(v, torch.Tensor):
                        v.record_stream(main)
        fused = F.fuse_detections(dets, self.method[0], self.method[1], max_class=self.max_class) if self.fuse else None
        return dets, fused


class HostFeeder:
    """Pinned-memory, double-buffered uploader: batches live in page-locked host memory, every `next()`
    enqueues the H2D copies of one batch on a copy stream into one of two device buffer sets and makes the current
    stream wait for them; the set being overwritten was consumed two calls ago (an event recorded at the following
    `next()` proves it), so the upload of batch i+1 overlaps the detectors working on batch i.
    `host_batches`: list (detectors) of pinned uint8 [B,H,W,C] tensors (one fixed batch, re-uploaded every step) or a
    callable returning such a list per step."""

    def __init__(self, host_batches, device):
        self.source = host_batches if callable(host_batches) else (lambda: host_batches)
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.bufs = None           # two device buffer sets, allocated from the first batch's shapes
        self.free = [[], []]   # events after which buffer set k may be overwritten
        self.k = 0
        self._last = None

    def next(self):
        """Upload the next batch; returns the list of device tensors (valid on the current stream)."""
        cur = torch.cuda.current_stream(self.device)
        k = self.k
        self.k ^= 1
        for ev in self.free[k]:          # set k was handed out two calls ago: wait until its consumers are done
            self.copy_stream.wait_event(ev)
        host = self.source()
        if self.bufs is None:
            self.bufs = [[torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in host] for _ in range(2)]
        with torch.cuda.stream(self.copy_stream):
            for dst, src in zip(self.bufs[k], host):
                assert src.is_pinned(), "HostFeeder needs page-locked host tensors (tensor.pin_memory())"
                dst.copy_(src, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        cur.wait_event(done)
        self._last = k
        return self.bufs[k]

    def mark_consumed(self, *streams):
        """Call right after enqueueing the work that reads the last batch: records, on every stream that reads it (the
        pipeline's side streams, or the current stream), the point after which the buffer set may be overwritten.  No
        stream is made to wait for another here."""
        sts = list(streams) or [torch.cuda.current_stream(self.device)]
        evs = []
        for st in sts:
            ev = torch.cuda.Event()
            ev.record(st)
            evs.append(ev)
        self.free[self._last] = evs
