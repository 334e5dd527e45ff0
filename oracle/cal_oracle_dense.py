"""Independent dense-adjacency restatement -- TEST INFRASTRUCTURE ONLY.

Second, structurally different formulation of the graph operators, used only to
pin ``oracle/cal_oracle.py`` (the reference has no tests,
so two independent restatements must agree).  **PARITY UNPINNED** against the
true reference for the same reasons given in ``cal_oracle.py``.

Everything is expressed through an ``[N, N]`` matrix:

* GCN (gcn_conv.py:44-104):  ``out = (D_r^-1/2 (A_w + L) D_c^-1/2)^T (X W) + b``
  where ``A_w[r, c]`` = summed weight of non-loop edges r->c, ``L`` = loop
  weight * I, and the degree is the *row* (source) sum -- the same vector is
  used on both sides, as the reference does.
* GAT (PyG 1.x GATConv): dense masked softmax over incoming edges + self loop,
  with edge multiplicities kept as counts.
* add-pool: ``P @ X`` with the ``[B, N]`` membership matrix.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def dense_weighted_adj(edge_index: Tensor, num_nodes: int, edge_weight: Optional[Tensor],
                       dtype) -> Tensor:
    row, col = edge_index
    w = torch.ones(row.numel(), dtype=dtype) if edge_weight is None else edge_weight.view(-1)
    keep = (row != col).to(dtype)
    a = torch.zeros(num_nodes, num_nodes, dtype=dtype)
    # accumulate (parallel edges add up), self loops dropped
    a = a.index_put((row, col), w * keep, accumulate=True)
    return a


def gcn_conv_dense(x: Tensor, edge_index: Tensor, weight: Tensor, bias: Optional[Tensor],
                   edge_weight: Optional[Tensor] = None, improved: bool = False) -> Tensor:
    n = x.size(0)
    a = dense_weighted_adj(edge_index, n, edge_weight, x.dtype)
    a = a + torch.eye(n, dtype=x.dtype) * (2.0 if improved else 1.0)
    deg = a.sum(dim=1)                       # row (source) degree, gcn_conv.py:65-66
    dis = deg.pow(-0.5)
    dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)
    a_hat = dis.view(-1, 1) * a * dis.view(1, -1)
    out = a_hat.t() @ (x @ weight)           # aggregate at the target (col) index
    return out if bias is None else out + bias


def gat_conv_dense(x: Tensor, edge_index: Tensor, weight: Tensor, att: Tensor,
                   bias: Optional[Tensor], heads: int, negative_slope: float = 0.2) -> Tensor:
    """Eval-mode GATv1 (no dropout)."""
    n = x.size(0)
    d = weight.size(1) // heads
    z = (x @ weight).view(n, heads, d)
    a_dst = (z * att[0, :, :d]).sum(-1)      # [N, K]  (x_i half)
    a_src = (z * att[0, :, d:]).sum(-1)      # [N, K]  (x_j half)
    row, col = edge_index
    cnt = torch.zeros(n, n, dtype=x.dtype)   # cnt[i(target), j(source)]
    keep = (row != col).to(x.dtype)
    cnt = cnt.index_put((col, row), keep, accumulate=True)
    cnt = cnt + torch.eye(n, dtype=x.dtype)
    e = F.leaky_relu(a_dst.view(n, 1, heads) + a_src.view(1, n, heads), negative_slope)
    mask = (cnt > 0).view(n, n, 1)
    e_m = e.masked_fill(~mask, float("-inf"))
    mx = e_m.max(dim=1, keepdim=True).values
    p = torch.exp(e_m - mx) * cnt.view(n, n, 1)
    alpha = p / (p.sum(dim=1, keepdim=True) + 1e-16)
    out = torch.einsum("ijk,jkd->ikd", alpha, z).reshape(n, heads * d)
    return out if bias is None else out + bias


def global_add_pool_dense(x: Tensor, batch: Tensor, size: int) -> Tensor:
    p = F.one_hot(batch, size).to(x.dtype).t()
    return p @ x


def edge_attention_dense(x: Tensor, edge_index: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """model.py:97-104 via per-node projections P = x W1^T, Q = x W2^T."""
    h = x.size(1)
    p = x @ w[:, :h].t()
    q = x @ w[:, h:].t()
    row, col = edge_index
    return F.softmax(p[row] + q[col] + b, dim=-1)
