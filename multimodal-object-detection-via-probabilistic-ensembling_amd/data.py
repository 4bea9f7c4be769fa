"""The slice of the reference's data layer the inference path touches:
dataset / metadata catalogs and COCO-json registration (detectron2/data/catalog.py,
data/datasets/register_coco.py:14, coco.py:29-195), the contiguous inference shards
(data/samplers/distributed_sampler.py:172-199), the batch-1 test loader (data/build.py:342-386) and the
ResizeShortestEdge size rule (data/transforms/transform_gen.py:192-213)."""
import json
import os
import types

import numpy as np
import torch

from . import comm


class _Metadata(types.SimpleNamespace):
    def get(self, key, default=None):
        return getattr(self, key, default)


class _MetadataCatalog:
    _store = {}

    @classmethod
    def get(cls, name):
        if name not in cls._store:
            cls._store[name] = _Metadata(name=name)
        return cls._store[name]


class _DatasetCatalog:
    _fns = {}

    @classmethod
    def register(cls, name, fn):
        cls._fns[name] = fn

    @classmethod
    def get(cls, name):
        return cls._fns[name]()

    @classmethod
    def list(cls):
        return list(cls._fns)


MetadataCatalog = _MetadataCatalog
DatasetCatalog = _DatasetCatalog


def load_coco_json(json_file, image_root, dataset_name=None):
    """COCO json -> list of dataset dicts {file_name, height, width, image_id, annotations[...]}; fills
    thing_classes / thing_dataset_id_to_contiguous_id of the dataset's metadata (sorted category ids)."""
    with open(json_file) as f:
        d = json.load(f)
    cats = sorted(d.get("categories", []), key=lambda c: c["id"])
    id_map = {c["id"]: i for i, c in enumerate(cats)}
    if dataset_name is not None:
        meta = MetadataCatalog.get(dataset_name)
        meta.thing_classes = [c["name"] for c in cats]
        meta.thing_dataset_id_to_contiguous_id = id_map
    anns = {}
    for a in d.get("annotations", []):
        anns.setdefault(a["image_id"], []).append(a)
    out = []
    for im in d["images"]:
        rec = {"file_name": os.path.join(image_root, im["file_name"]), "height": im["height"], "width": im["width"],
               "image_id": im["id"], "annotations": []}
        for a in anns.get(im["id"], []):
            if a.get("ignore", 0) != 0:
                continue
            rec["annotations"].append({"bbox": a["bbox"], "bbox_mode": 1, "iscrowd": a.get("iscrowd", 0),
                                       "category_id": id_map.get(a["category_id"], a["category_id"])})
        out.append(rec)
    return out


def register_coco_instances(name, metadata, json_file, image_root):
    DatasetCatalog.register(name, lambda: load_coco_json(json_file, image_root, name))
    meta = MetadataCatalog.get(name)
    meta.json_file, meta.image_root, meta.evaluator_type = json_file, image_root, "coco"
    for k, v in metadata.items():
        setattr(meta, k, v)


def resize_shortest_edge_shape(h, w, short_edge_length=800, max_size=1333):
    """ResizeShortestEdge.get_transform (transform_gen.py:192-213): short side -> 800 capped so the long side <= 1333,
    round half up; short_edge_length 0 = NoOpTransform (the image keeps its size)."""
    size = short_edge_length
    if size == 0:
        return int(h), int(w)
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


PIL_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_tables(in_size, out_size):
    """Fixed-point tap tables of Pillow's bilinear resampler for one axis (libImaging/Resample.c precompute_coeffs +
    normalize_coeffs_8bpc; the reference resizes 3-channel images with Image.resize(..., BILINEAR),
    data/transforms/transform.py:92-97).  Returns int32 [out_size, 2 + ksize]: first tap, tap count, 22-bit weights.
    float64 arithmetic in Pillow's order, so the integers are Pillow's integers (tests/test_oracle_resize.py)."""
    scale = filterscale = float(in_size) / float(out_size)
    filterscale = max(filterscale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5), 0.0).astype(np.int64)
    xmax = np.minimum(np.trunc(center + support + 0.5), float(in_size)).astype(np.int64)
    n = xmax - xmin
    w = np.zeros((out_size, ksize), dtype=np.float64)
    ww = np.zeros(out_size, dtype=np.float64)
    for x in range(ksize):
        v = np.abs((x + xmin - center + 0.5) * ss)
        wx = np.where((v < 1.0) & (x < n), 1.0 - v, 0.0)
        w[:, x] = wx
        ww = ww + wx              # same left-to-right sum as the C loop
    nz = ww != 0.0
    w[nz] = w[nz] / ww[nz, None]
    tab = np.empty((out_size, 2 + ksize), dtype=np.int32)
    tab[:, 0], tab[:, 1] = xmin, n
    tab[:, 2:] = np.trunc(0.5 + w * float(1 << PIL_PRECISION_BITS)).astype(np.int32)
    return tab


def cv2_linear_resize_u8(img, new_h, new_w):
    """cv2.resize(img, (new_w, new_h)) with the default INTER_LINEAR on a uint8 [H, W, C] image: the resize the
    reference applies to the RGB frame before stacking it with the thermal one
    (demo/FLIR/demo_FLIR_save_predictions.py:109 - the `cv2.INTER_CUBIC` argument lands in the `dst` slot, so the
    interpolation is the default bilinear).  2 x 2 taps, NO antialiasing - unlike Pillow's BILINEAR, whose
    support grows with the down-scale factor.  OpenCV is a third-party dependency that is absent offline (4.6.0 in
    probEn.yml): this restates its published 8-bit rule - half-pixel source coordinates, 11-bit fixed-point
    coefficients (cvRound(w * 2048), saturated to int16), horizontal pass in int32, vertical pass
    ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2 - PARITY UNPINNED (no OpenCV to check against)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w = img.shape[:2]
    if (h, w) == (new_h, new_w):
        return img.copy()

    def taps(n_in, n_out):
        scale = n_in / n_out
        f = (np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5
        i0 = np.floor(f).astype(np.int64)
        a = (f - i0).astype(np.float32)
        lo = i0 < 0
        a[lo], i0[lo] = 0.0, 0
        hi = i0 >= n_in - 1
        a[hi], i0[hi] = 0.0, n_in - 1
        i1 = np.minimum(i0 + 1, n_in - 1)
        c1 = np.clip(np.rint(a.astype(np.float64) * 2048.0), -32768, 32767).astype(np.int32)   # cvRound: half to even
        c0 = np.clip(np.rint((1.0 - a).astype(np.float64) * 2048.0), -32768, 32767).astype(np.int32)
        return i0, i1, c0, c1
    x0, x1, ax0, ax1 = taps(w, new_w)
    y0, y1, by0, by1 = taps(h, new_h)
    src = img.astype(np.int32)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]          # [h, new_w, C], scaled by 2^11
    r0, r1 = rows[y0] >> 4, rows[y1] >> 4
    out = (((by0[:, None, None] * r0) >> 16) + ((by1[:, None, None] * r1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def cv2_linear_resize_f(img, new_h, new_w):
    """cv2.resize(img, (new_w, new_h)) default INTER_LINEAR on a floating-point [H, W, C] image (the reference's branch
    for 4- / 6-channel inputs, transform.py:82-91): half-pixel coordinates computed in double and cast to float, float
    weights, horizontal then vertical pass in the image's dtype.  PARITY UNPINNED (OpenCV absent); the GPU preprocess
    kernel (pe_preprocess_pack, src_kind 1) and oracle/resize.py follow the same rule."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    dt = img.dtype if img.dtype in (np.float32, np.float64) else np.float64

    def taps(n_in, n_out):
        scale = n_in / n_out
        f = (np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5
        i0 = np.floor(f).astype(np.int64)
        a = (f - i0).astype(np.float32)
        lo = i0 < 0
        a[lo], i0[lo] = 0.0, 0
        hi = i0 >= n_in - 1
        a[hi], i0[hi] = 0.0, n_in - 1
        return i0, np.minimum(i0 + 1, n_in - 1), (np.float32(1.0) - a), a
    x0, x1, ax0, ax1 = taps(w, new_w)
    y0, y1, by0, by1 = taps(h, new_h)
    src = img.astype(dt)
    rows = src[:, x0] * ax0[None, :, None].astype(dt) + src[:, x1] * ax1[None, :, None].astype(dt)
    return rows[y0] * by0[:, None, None].astype(dt) + rows[y1] * by1[:, None, None].astype(dt)


class InferenceSampler:
    """Contiguous per-rank index blocks of ceil(N / W)."""

    def __init__(self, size, rank=None, world=None):
        self._range = comm.shard_range(size, rank, world)

    def __iter__(self):
        yield from self._range

    def __len__(self):
        return len(self._range)


def read_image(file_name, format="BGR"):
    """PIL decode -> HWC uint8 in the requested channel order (detection_utils.py:37-96, 3-channel case)."""
    from PIL import Image
    with open(file_name, "rb") as f:
        img = np.asarray(Image.open(f).convert("RGB"))
    return img[:, :, ::-1].copy() if format == "BGR" else img


def build_detection_test_loader(dataset_dicts, mapper, rank=None, world=None):
    """Batch-1, rank-sharded, in-order loader of mapped dataset dicts (list of lists, like the reference's
    trivial_batch_collator)."""
    idx = list(InferenceSampler(len(dataset_dicts), rank, world))
    return [[mapper(dataset_dicts[i])] for i in idx]
