"""Golden vectors for the Pillow bilinear resize the reference applies to 3-channel images
(detectron2/data/transforms/transform.py:92-97).  Run in the build container (needs Pillow):
    python tests/golden/gen_pil_resize.py
Writes tests/golden/pil_resize.npz: seeded uint8 inputs and Pillow's outputs for up- and down-scaling."""
import os

import numpy as np
from PIL import Image
import PIL

CASES = [(48, 60, 75, 94), (37, 53, 80, 91), (64, 64, 31, 47), (40, 80, 40, 133), (90, 30, 135, 30), (128, 160, 200, 250)]


def main():
    rng = np.random.default_rng(2024)
    out = {"pillow_version": np.array(PIL.__version__)}
    for i, (h, w, nh, nw) in enumerate(CASES):
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        if i == 0:
            img[:, : w // 2] = 255     # saturated region: exercises the clip
        out[f"in{i}"] = img
        out[f"size{i}"] = np.array([nh, nw])
        out[f"out{i}"] = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pil_resize.npz"), **out)


if __name__ == "__main__":
    main()
