"""`Boxes`, `Instances`, `ImageList`: the containers of the reference's Python API
(detectron2/structures/boxes.py:125-296, instances.py:9-187, image_list.py:8-102), restated.
Same constructor arguments, attribute names and error behaviour, so code written against the
reference (`inst.pred_boxes.tensor`, `inst.scores`, `Instances.cat`, `inst.to('cpu')` ...) runs unchanged."""
import itertools
from typing import Any, Dict, List, Tuple, Union

import numpy as np
import torch


class Boxes:
    """N boxes as a float32 tensor [N,4] in absolute (x1, y1, x2, y2)."""

    def __init__(self, tensor):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device("cpu")
        tensor = torch.as_tensor(np.asarray(tensor) if isinstance(tensor, (list, tuple)) else tensor,
                                 dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, 4)).to(dtype=torch.float32, device=device)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, device):
        return Boxes(self.tensor.to(device))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size):
        assert torch.isfinite(self.tensor).all(), "Box tensor contains infinite or NaN!"
        h, w = box_size
        self.tensor[:, 0].clamp_(min=0, max=w)
        self.tensor[:, 1].clamp_(min=0, max=h)
        self.tensor[:, 2].clamp_(min=0, max=w)
        self.tensor[:, 3].clamp_(min=0, max=h)

    def nonempty(self, threshold=0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, "Indexing on Boxes with {} failed to return a matrix!".format(item)
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"

    def inside_box(self, box_size, boundary_threshold=0):
        height, width = box_size
        t = self.tensor
        return ((t[..., 0] >= -boundary_threshold) & (t[..., 1] >= -boundary_threshold)
                & (t[..., 2] < width + boundary_threshold) & (t[..., 3] < height + boundary_threshold))

    def get_centers(self):
        return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

    def scale(self, scale_x, scale_y):
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    @staticmethod
    def cat(boxes_list):
        assert isinstance(boxes_list, (list, tuple))
        assert len(boxes_list) > 0
        assert all(isinstance(b, Boxes) for b in boxes_list)
        return Boxes(torch.cat([b.tensor for b in boxes_list], dim=0))

    @property
    def device(self):
        return self.tensor.device

    def __iter__(self):
        yield from self.tensor


class Instances:
    """Per-image bag of equally long fields (`pred_boxes`, `scores`, `pred_classes`, `class_logits`,
    `prob_score`, `vars`, ...) plus the (height, width) they refer to."""

    def __init__(self, image_size, **kwargs):
        self._image_size = image_size
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return self._fields[name]

    def set(self, name, value):
        data_len = len(value)
        if len(self._fields):
            assert len(self) == data_len, "Adding a field of length {} to a Instances of length {}".format(data_len, len(self))
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def to(self, device):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(device)
            ret.set(k, v)
        return ret

    def __getitem__(self, item):
        if type(item) == int:
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        raise NotImplementedError("Empty Instances does not support __len__!")

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    @staticmethod
    def cat(instance_lists):
        assert all(isinstance(i, Instances) for i in instance_lists)
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        image_size = instance_lists[0].image_size
        for i in instance_lists[1:]:
            assert i.image_size == image_size
        ret = Instances(image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = list(itertools.chain(*values))
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError("Unsupported type {} for concatenation".format(type(v0)))
            ret.set(k, values)
        return ret

    def __str__(self):
        s = self.__class__.__name__ + "("
        s += "num_instances={}, ".format(len(self))
        s += "image_height={}, ".format(self._image_size[0])
        s += "image_width={}, ".format(self._image_size[1])
        s += "fields=[{}])".format(", ".join(self._fields.keys()))
        return s

    __repr__ = __str__


class ImageList:
    """A padded batch tensor + the unpadded (h, w) of every image."""

    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, idx):
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    def to(self, *args, **kwargs):
        return ImageList(self.tensor.to(*args, **kwargs), self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        assert len(tensors) > 0 and isinstance(tensors, (tuple, list))
        sizes = [tuple(t.shape[-2:]) for t in tensors]
        H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if size_divisibility > 0:
            d = size_divisibility
            H, W = (H + d - 1) // d * d, (W + d - 1) // d * d
        batch = tensors[0].new_full((len(tensors),) + tuple(tensors[0].shape[:-2]) + (H, W), pad_value)
        for img, pad in zip(tensors, batch):
            pad[..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batch.contiguous(), sizes)
