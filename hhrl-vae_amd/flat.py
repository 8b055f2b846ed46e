"""Flat fp32 parameter buffers with named views.

Every network keeps its parameters, gradients and Adam moments in ONE
contiguous device buffer each, in the reference's `parameters()` order, so the
optimiser and the Polyak update are single coalesced streams over HBM and a
gradient all-reduce is one message.  Named views into the buffer use the
reference's `state_dict` keys, which keeps checkpoints interchangeable.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch

ALIGN = 64  # floats: every tensor starts on a 256-byte boundary

Shape = Tuple[int, ...]


def _numel(shape: Shape) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n


class Layout:
    """Offsets of named tensors inside a flat buffer."""

    def __init__(self, shapes: Sequence[Tuple[str, Shape]]):
        self.shapes: "OrderedDict[str, Shape]" = OrderedDict((n, tuple(int(x) for x in s)) for n, s in shapes)
        self.offsets: Dict[str, int] = {}
        off = 0
        for name, shp in self.shapes.items():
            self.offsets[name] = off
            off += -(-_numel(shp) // ALIGN) * ALIGN
        self.size = max(off, ALIGN)
        self.names: List[str] = list(self.shapes)

    def numel(self) -> int:
        """Number of real (unpadded) parameters."""
        return sum(_numel(s) for s in self.shapes.values())


class FlatBuffer:
    """One zero-initialised flat fp32 tensor plus per-name views (padding
    stays zero forever: Adam and Polyak map 0 -> 0)."""

    def __init__(self, layout: Layout, device):
        self.layout = layout
        # `full` = the buffer plus a tail of ALIGN scratch floats right behind it: a data-parallel step
        # puts its loss there so that gradient and loss travel in one all-reduce
        self.full = torch.zeros(layout.size + ALIGN, dtype=torch.float32, device=device)
        self.data = self.full[:layout.size]
        self.views: Dict[str, torch.Tensor] = {}
        for name, shp in layout.shapes.items():
            off = layout.offsets[name]
            self.views[name] = self.data[off: off + _numel(shp)].view(shp)
        self._base = self.data.data_ptr()

    def ptr(self, name: str) -> int:
        return self._base + 4 * self.layout.offsets[name]

    @property
    def base_ptr(self) -> int:
        return self._base

    @property
    def tail_ptr(self) -> int:
        return self._base + 4 * self.layout.size

    def __getitem__(self, name: str) -> torch.Tensor:
        return self.views[name]

    def load(self, state: Dict[str, "np.ndarray | torch.Tensor"], strict: bool = True) -> None:
        missing = [n for n in self.layout.names if n not in state]
        extra = [n for n in state if n not in self.layout.shapes]
        if strict and (missing or extra):
            raise KeyError("state_dict mismatch: missing %s, unexpected %s" % (missing, extra))
        for name in self.layout.names:
            if name not in state:
                continue
            src = torch.as_tensor(np.asarray(state[name]) if not isinstance(state[name], torch.Tensor) else state[name])
            if tuple(src.shape) != self.layout.shapes[name]:
                raise ValueError("%s: shape %s, expected %s" % (name, tuple(src.shape), self.layout.shapes[name]))
            self.views[name].copy_(src.to(dtype=torch.float32))

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        """Detached CPU copies under the reference's key names."""
        return OrderedDict((n, self.views[n].detach().cpu().clone()) for n in self.layout.names)

    def copy_from(self, other: "FlatBuffer") -> None:
        self.data.copy_(other.data)
