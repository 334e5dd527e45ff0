"""GCNConv with the reference's constructor / forward / parameter surface
(gcn_conv.py:10-108), executing on libcalhip.

``GCNConv(in, out, improved=False, cached=False, bias=True, edge_norm=True,
gfn=False)(x, edge_index, edge_weight=None)``; parameters ``weight [in, out]``
(glorot) and ``bias [out]`` (zeros).  Extra keyword-only arguments (``plan``,
``relu``) let a caller share one GraphPlan across layers and fuse the ReLU the
models apply right after the conv (model.py:95,112-113).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch.nn import Parameter

from . import ops
from .plan import GraphPlan


def glorot(t: torch.Tensor):
    """PyG inits.glorot (call site gcn_conv.py:40)."""
    if t is not None:
        a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
        with torch.no_grad():
            t.uniform_(-a, a)


def zeros(t: Optional[torch.Tensor]):
    if t is not None:
        with torch.no_grad():
            t.fill_(0)


class GCNConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, improved=False, cached=False, bias=True,
                 edge_norm=True, gfn=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.improved = improved
        self.cached = cached
        self.cached_result = None
        self.edge_norm = edge_norm
        self.gfn = gfn
        self.message_mask = None
        self.weight = Parameter(torch.empty(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        zeros(self.bias)
        self.cached_result = None

    def forward(self, x, edge_index, edge_weight=None, *, plan: Optional[GraphPlan] = None,
                relu: bool = False):
        x = ops.matmul(x, self.weight, relu=relu and self.gfn)  # gcn_conv.py:75 (fp32 MFMA GEMM)
        if self.gfn:                                            # gcn_conv.py:76-77
            return x
        if not self.edge_norm:
            raise NotImplementedError("edge_norm=False is never used by the CAL models")
        if plan is None:
            if self.cached and self.cached_result is not None:  # gcn_conv.py:79,91
                plan = self.cached_result
            else:
                plan = GraphPlan(edge_index, x.size(0))
                self.cached_result = plan if self.cached else None
        return ops.gcn_aggregate(x, plan, edge_weight, self.bias, self.improved, relu)

    def __repr__(self):
        return "{}({}, {})".format(self.__class__.__name__, self.in_channels, self.out_channels)
