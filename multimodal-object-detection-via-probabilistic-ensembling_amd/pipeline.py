"""Frame-pair pipeline: N detectors over the same batch of aligned frames + ProbEn, device to device.

This is the reference's two-stage, file-based flow (demo_FLIR_save_predictions.py per detector -> JSON ->
demo_probEn.py) as ONE stream of launches: every detector's forward is enqueued on its own HIP stream (their
small late-stage kernels - res5, p5/p6, heads, NMS - do not fill 256 CUs alone and overlap with the other
detector's work), the main stream waits for all of them and runs the packing + ProbEn kernels.
(Measured r01, batch 32 pairs: 2 streams 680-700 pairs/s vs 622 serial; splitting each detector's batch over more
streams is slower - 651 at 2 sub-batches, 612 at 4 - smaller launches tile worse.)"""
import torch

from . import fusion as F


class FramePairPipeline:
    def __init__(self, models, score_fusion="probEn", box_fusion="v-avg", max_class=2, concurrent=True):
        self.models = list(models)
        self.method = (score_fusion, box_fusion)
        self.max_class = max_class
        self.concurrent = concurrent and len(self.models) > 1
        self.streams = [torch.cuda.Stream() for _ in self.models] if self.concurrent else None

    @torch.no_grad()
    def __call__(self, frames_per_detector, out_sizes, resize_to):
        """frames_per_detector[d]: the batch for detector d ([B,H,W,C] uint8/float tensor or list of tensors).
        Returns (list of per-detector result dicts, fused dict of fusion.fuse_detections)."""
        if not self.concurrent:
            dets = [m.forward_batch(fr, out_sizes=out_sizes, resize_to=resize_to)
                    for m, fr in zip(self.models, frames_per_detector)]
        else:
            main = torch.cuda.current_stream()
            dets = []
            for m, fr, st in zip(self.models, frames_per_detector, self.streams):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    dets.append(m.forward_batch(fr, out_sizes=out_sizes, resize_to=resize_to))
            for st in self.streams:
                main.wait_stream(st)
        fused = F.fuse_detections(dets, self.method[0], self.method[1], max_class=self.max_class)
        return dets, fused
