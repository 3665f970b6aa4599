"""``SampleList``: batch dict-of-tensors with attribute access and ``.to(device)``
(mirror of the parts of ``pythia/common/sample.py:58-326`` that the model boundary uses)."""
from collections import OrderedDict

import torch


class SampleList(OrderedDict):
    def __init__(self, fields=None):
        super().__init__()
        if fields:
            items = fields.items() if hasattr(fields, "items") else fields      # mapping or (key, value) tuples
            for k, v in items:
                self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def add_field(self, k, v):
        self[k] = v

    def fields(self):
        return list(self.keys())

    def get_batch_size(self):
        for v in self.values():
            if torch.is_tensor(v) and v.dim() > 0:
                return v.size(0)
        return 0

    def to(self, device, non_blocking=True):
        out = SampleList()
        for k, v in self.items():
            out[k] = v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v
        return out
