"""GATConv (GATv1, concat heads) with the PyG-1.x parameter surface the
reference's ``GATConv(hidden, hidden // head, heads=head, dropout=dropout)``
call sites (model.py:340,390) rely on: ``weight [in, heads*out]``,
``att [1, heads, 2*out]`` (target half first), ``bias [heads*out]``.

State dicts written by PyG 2.x (``lin_src.weight``/``lin.weight``, ``att_src``,
``att_dst``) are converted on load.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.nn import Parameter

from . import ops
from .gcn_conv import glorot, zeros
from .plan import GraphPlan


class GATConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2,
                 dropout=0.0, bias=True):
        super().__init__()
        if not concat:
            raise NotImplementedError("concat=False is never used by the CAL models")
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.weight = Parameter(torch.empty(in_channels, heads * out_channels))
        self.att = Parameter(torch.empty(1, heads, 2 * out_channels))
        if bias:
            self.bias = Parameter(torch.empty(heads * out_channels))
        else:
            self.register_parameter("bias", None)
        self._calls = 0
        self.seed: Optional[int] = None       # fixed seed for the attention dropout (tests)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        glorot(self.att)
        zeros(self.bias)

    def _next_seed(self) -> int:
        if self.seed is not None:
            return int(self.seed)
        self._calls += 1
        return int(torch.initial_seed() * 1000003 + id(self) % 65521 * 7919 + self._calls) & ((1 << 63) - 1)

    def forward(self, x, edge_index, *, plan: Optional[GraphPlan] = None, relu: bool = False):
        z = ops.matmul(x, self.weight)
        if plan is None:
            plan = GraphPlan(edge_index, x.size(0))
        p = float(self.dropout) if self.training else 0.0
        seed = self._next_seed() if p > 0 else 0
        self.last_seed = seed
        return ops.gat_aggregate(z, self.att, self.bias, plan, self.heads, self.negative_slope, p, seed, relu)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kw):
        # PyG >= 2.0 layouts -> the 1.x layout used here
        for src in ("lin_src.weight", "lin.weight", "lin_l.weight"):
            k = prefix + src
            if k in state_dict:
                state_dict[prefix + "weight"] = state_dict.pop(k).t().contiguous()
        for drop in ("lin_dst.weight", "lin_r.weight"):
            state_dict.pop(prefix + drop, None)
        ks, kd = prefix + "att_src", prefix + "att_dst"
        if ks in state_dict and kd in state_dict:
            state_dict[prefix + "att"] = torch.cat([state_dict.pop(kd), state_dict.pop(ks)], dim=-1)
        super()._load_from_state_dict(state_dict, prefix, *args, **kw)

    def __repr__(self):
        return "{}({}, {}, heads={})".format(self.__class__.__name__, self.in_channels,
                                             self.out_channels, self.heads)
