"""`Boxes`, `Instances`, `ImageList`: the containers of the reference's Python API.

Written from the CONTRACT of detectron2/structures/boxes.py:125-296, instances.py:9-187 and image_list.py:8-102
(constructor arguments, attribute / method names, return types, error types and messages that callers see) so that
code written against the reference (`inst.pred_boxes.tensor`, `inst.scores`, `Instances.cat`, `inst.to('cpu')`,
`Boxes.clip`, ...) runs unchanged.  The bodies are this repository's own:

  * `Boxes` is a thin view over one float32 [N,4] tensor; every geometric method is a single vectorised expression
    against a per-column bound / factor vector (no per-column slicing);
  * `Instances` is a field table with ONE explicit row-count invariant (`_rows`), and one `_map` primitive from which
    `to`, indexing and `cat` are derived; what can be concatenated is decided by a small dispatch table.
"""
import torch

_XYXY = 4


def _as_box_tensor(data):
    """Anything list-like / tensor -> float32 [N,4] on the tensor's own device ([] -> [0,4])."""
    if isinstance(data, torch.Tensor):
        t = data.to(torch.float32)
    else:
        t = torch.tensor(data, dtype=torch.float32) if len(data) else torch.zeros((0, _XYXY), dtype=torch.float32)
    if t.numel() == 0:
        t = t.reshape(0, _XYXY)
    assert t.dim() == 2 and t.size(-1) == _XYXY, t.size()
    return t


class Boxes:
    """N boxes, absolute (x1, y1, x2, y2), float32.  `.tensor` is THE storage: in-place methods (`clip`, `scale`)
    modify it, like the reference's."""

    def __init__(self, tensor):
        self.tensor = _as_box_tensor(tensor)

    # -- construction / movement --------------------------------------------------------------------------------
    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, device):
        return Boxes(self.tensor.to(device))

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def cat(boxes_list):
        assert isinstance(boxes_list, (list, tuple)) and len(boxes_list) > 0
        assert all(isinstance(b, Boxes) for b in boxes_list)
        return Boxes(torch.cat([b.tensor for b in boxes_list], dim=0))

    # -- geometry -------------------------------------------------------------------------------------------------
    def _wh(self):
        return self.tensor[:, 2:] - self.tensor[:, :2]          # [N,2] = (width, height)

    def area(self):
        return self._wh().prod(dim=1)

    def nonempty(self, threshold=0):
        return (self._wh() > threshold).all(dim=1)

    def get_centers(self):
        return self.tensor.view(-1, 2, 2).mean(dim=1)           # midpoint of the two corners

    def clip(self, box_size):
        assert torch.isfinite(self.tensor).all(), "Box tensor contains infinite or NaN!"
        h, w = box_size
        hi = self.tensor.new_tensor([w, h, w, h])
        torch.minimum(self.tensor.clamp_(min=0), hi, out=self.tensor)

    def scale(self, scale_x, scale_y):
        self.tensor.mul_(self.tensor.new_tensor([scale_x, scale_y, scale_x, scale_y]))

    def inside_box(self, box_size, boundary_threshold=0):
        h, w = box_size
        t = boundary_threshold
        lo_ok = (self.tensor[..., :2] >= -t).all(dim=-1)
        hi_ok = (self.tensor[..., 2:] < self.tensor.new_tensor([w + t, h + t])).all(dim=-1)
        return lo_ok & hi_ok

    # -- container protocol -------------------------------------------------------------------------------------------
    def __getitem__(self, item):
        picked = self.tensor[item]
        if isinstance(item, int):
            picked = picked.view(1, -1)                         # Boxes[i] stays a (1-row) Boxes
        assert picked.dim() == 2, "Indexing on Boxes with {} failed to return a matrix!".format(item)
        return Boxes(picked)

    def __len__(self):
        return self.tensor.shape[0]

    def __iter__(self):
        yield from self.tensor                                  # rows, as [4] tensors

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"


def _cat_values(values):
    """Concatenate one field's per-Instances values: tensors, Python lists, or anything exposing `type(v).cat`."""
    first = values[0]
    if isinstance(first, torch.Tensor):
        return torch.cat(values, dim=0)
    if isinstance(first, list):
        return [x for v in values for x in v]
    cat = getattr(type(first), "cat", None)
    if cat is None:
        raise ValueError("Unsupported type {} for concatenation".format(type(first)))
    return cat(values)


class Instances:
    """Per-image table of equally long fields (`pred_boxes`, `scores`, `pred_classes`, `class_logits`, `prob_score`,
    `vars`, ...) plus the (height, width) they refer to.  Fields are attributes: `inst.scores = t`, `inst.scores`."""

    def __init__(self, image_size, **kwargs):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        object.__setattr__(self, "_rows", None)     # the one length every field must have (None: no field yet)
        for name, value in kwargs.items():
            self.set(name, value)

    @property
    def image_size(self):
        return self._image_size

    # -- attribute <-> field routing -------------------------------------------------------------------------------
    def __setattr__(self, name, value):
        if name.startswith("_"):
            object.__setattr__(self, name, value)
        else:
            self.set(name, value)

    def __getattr__(self, name):            # only reached when normal lookup fails
        fields = self.__dict__.get("_fields", {})
        if name in fields:
            return fields[name]
        raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))

    # -- field table ---------------------------------------------------------------------------------------------------
    def set(self, name, value):
        n = len(value)
        assert self._rows is None or self._rows == n, \
            "Adding a field of length {} to a Instances of length {}".format(n, self._rows)
        self._fields[name] = value
        object.__setattr__(self, "_rows", n)

    def has(self, name):
        return name in self._fields

    def get(self, name):
        return self._fields[name]

    def remove(self, name):
        del self._fields[name]
        if not self._fields:
            object.__setattr__(self, "_rows", None)

    def get_fields(self):
        return self._fields

    def _map(self, fn):
        """New Instances of the same image with fn applied to every field."""
        out = Instances(self._image_size)
        for name, value in self._fields.items():
            out.set(name, fn(value))
        return out

    def to(self, device):
        return self._map(lambda v: v.to(device) if hasattr(v, "to") else v)

    def __getitem__(self, item):
        if type(item) is int:
            n = len(self)
            if not -n <= item < n:
                raise IndexError("Instances index out of range!")
            item = slice(item % n, item % n + 1)                # keep a 1-row table
        return self._map(lambda v: v[item])

    def __len__(self):
        if self._rows is None:
            raise NotImplementedError("Empty Instances does not support __len__!")
        return self._rows

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    @staticmethod
    def cat(instance_lists):
        assert len(instance_lists) > 0 and all(isinstance(i, Instances) for i in instance_lists)
        first = instance_lists[0]
        if len(instance_lists) == 1:
            return first
        assert all(i.image_size == first.image_size for i in instance_lists[1:])
        out = Instances(first.image_size)
        for name in first._fields:
            out.set(name, _cat_values([i.get(name) for i in instance_lists]))
        return out

    def __str__(self):
        return "{}(num_instances={}, image_height={}, image_width={}, fields=[{}])".format(
            type(self).__name__, len(self), self._image_size[0], self._image_size[1], ", ".join(self._fields))

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
