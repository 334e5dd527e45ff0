"""autograd wrappers around the C-ABI entry points (operator-level boundary).

Every Function hands its tensors to ``plan._c``, which calls straight into the
C ABI: CUDA tensors -> libcalhip.so (HIP kernels, torch's current stream), CPU
tensors -> libcalhost.so (the plain-C++ host implementation of the same
symbols).  Tensors of one call must live on one device.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd import Function

from . import _lib
from .plan import GraphPlan, _c, _q


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (got %s)" % (name, t.dtype))
    return t.contiguous()


def _empty(n: int, dev) -> torch.Tensor:
    return torch.empty(max(int(n), 4), dtype=torch.float32, device=dev)


class _GCNAggregate(Function):
    """norm + propagate + bias (+ReLU): gcn_conv.py:79-104 after the dense x @ W."""

    @staticmethod
    def forward(ctx, h, w, bias, plan: GraphPlan, loop_w: float, relu: bool):
        h = _f32(h, "h")
        N, H = h.shape
        if N != plan.N:
            raise ValueError("feature rows (%d) != plan nodes (%d)" % (N, plan.N))
        if w is None:
            dis, norm = plan.unit_norm(loop_w)
        else:
            w = _f32(w, "edge_weight").view(-1)
            if w.numel() != plan.E:
                raise AssertionError("edge_weight.size(0) != edge_index.size(1)")   # gcn_conv.py:54
            dis, norm = _empty(N, h.device), _empty(plan.E, h.device)
            _c("cal_gcn_norm_fwd", plan.rowptr_src, plan.eid_src, plan.row32, plan.col32,
                      w, loop_w, N, plan.E, dis, norm)
        if bias is not None:
            bias = _f32(bias, "bias")
        out = torch.empty_like(h)
        _c("cal_spmm_fwd", plan.rowptr_dst, plan.nbr_dst, plan.eid_dst, norm, dis,
                  loop_w, h, bias, int(relu), out, N, H)
        ctx.plan, ctx.loop_w, ctx.relu, ctx.has_bias = plan, loop_w, relu, bias is not None
        ctx.has_w = w is not None
        need_h = ctx.has_w and ctx.needs_input_grad[1]
        ctx.save_for_backward(h if need_h else None, w, dis, norm, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, gout):
        h, w, dis, norm, out = ctx.saved_tensors
        plan, loop_w = ctx.plan, ctx.loop_w
        gout = _f32(gout, "grad_out")
        N, H = gout.shape
        dev = gout.device
        need_bias = ctx.has_bias and ctx.needs_input_grad[2]
        dz = torch.empty_like(gout) if ctx.relu else gout
        dbias = torch.empty(H, dtype=torch.float32, device=dev) if need_bias else None
        if ctx.relu or need_bias:
            part = _empty(_q(gout, "cal_colsum_parts", N) * H, dev) if need_bias else None
            _c("cal_relu_bwd_colsum", gout, out if ctx.relu else None,
                      dz if ctx.relu else None, dbias, part, N, H)
        dh = None
        if ctx.needs_input_grad[0]:
            dh = torch.empty_like(gout)
            _c("cal_spmm_fwd", plan.rowptr_src, plan.nbr_src, plan.eid_src, norm, dis,
                      loop_w, dz, None, 0, dh, N, H)
        dw = None
        if ctx.has_w and ctx.needs_input_grad[1]:
            dw = _empty(plan.E, dev)[:plan.E]
            gn, gself, ddeg = _empty(plan.E, dev), _empty(N, dev), _empty(N, dev)
            _c("cal_gcn_norm_bwd", plan.rowptr_dst, plan.nbr_dst, plan.eid_dst,
                      plan.rowptr_src, plan.nbr_src, plan.eid_src, plan.row32, plan.col32,
                      w, dis, loop_w, h, dz, gn, gself, ddeg, dw,
                      N, plan.E, H)
        return dh, dw, dbias, None, None, None


def gcn_aggregate(h, plan: GraphPlan, edge_weight=None, bias=None, improved: bool = False,
                  relu: bool = False):
    return _GCNAggregate.apply(h, edge_weight, bias, plan, 2.0 if improved else 1.0, relu)


class _EdgeAttention(Function):
    """model.py:97-104 -> [2, E] (row 0: edge_weight_c, row 1: edge_weight_o)."""

    @staticmethod
    def forward(ctx, x, W, b, plan: GraphPlan):
        x, W, b = _f32(x, "x"), _f32(W, "edge_att_mlp.weight"), _f32(b, "edge_att_mlp.bias")
        N, H = x.shape
        if W.shape != (2, 2 * H):
            raise ValueError("edge_att_mlp.weight must be [2, 2*hidden]")
        att = _empty(2 * plan.E, x.device)[:2 * plan.E].view(2, plan.E)
        pq = _empty(4 * N, x.device)
        _c("cal_edge_att_fwd", x, W, b, plan.row32, plan.col32, pq, att,
                  N, plan.E, H)
        ctx.plan = plan
        ctx.save_for_backward(x, W, att)
        return att

    @staticmethod
    def backward(ctx, datt):
        x, W, att = ctx.saved_tensors
        plan = ctx.plan
        datt = _f32(datt, "grad")
        N, H = x.shape
        dx = torch.empty_like(x)
        dW = torch.empty_like(W)
        db = torch.empty(2, dtype=torch.float32, device=x.device)
        ws = _empty(_q(x, "cal_edge_att_bwd_ws", N, plan.E, H), x.device)
        _c("cal_edge_att_bwd", x, W, att, datt, plan.rowptr_src, plan.eid_src,
                  plan.rowptr_dst, plan.eid_dst, dx, 0, dW, db, ws, N, plan.E, H)
        return dx, dW, db, None


def edge_attention(x, weight, bias, plan: GraphPlan):
    return _EdgeAttention.apply(x, weight, bias, plan)


class _NodeAttentionSplit(Function):
    """model.py:106-111 -> (xc, xo, node_att)."""

    @staticmethod
    def forward(ctx, x, Wn, bn):
        x, Wn, bn = _f32(x, "x"), _f32(Wn, "node_att_mlp.weight"), _f32(bn, "node_att_mlp.bias")
        N, H = x.shape
        att = _empty(2 * N, x.device)[:2 * N].view(N, 2)
        xc, xo = torch.empty_like(x), torch.empty_like(x)
        _c("cal_node_att_split_fwd", x, Wn, bn, att, xc, xo, N, H)
        ctx.save_for_backward(x, Wn, att)
        ctx.mark_non_differentiable(att)
        return xc, xo, att

    @staticmethod
    def backward(ctx, dxc, dxo, _datt):
        x, Wn, att = ctx.saved_tensors
        N, H = x.shape
        dxc = torch.zeros_like(x) if dxc is None else _f32(dxc, "grad_xc")
        dxo = torch.zeros_like(x) if dxo is None else _f32(dxo, "grad_xo")
        dx, dWn = torch.empty_like(x), torch.empty_like(Wn)
        dbn = torch.empty(2, dtype=torch.float32, device=x.device)
        ws = _empty(_q(x, "cal_node_att_bwd_ws", N, H), x.device)
        _c("cal_node_att_split_bwd", x, Wn, att, dxc, dxo, dx, dWn, dbn,
                  ws, N, H)
        return dx, dWn, dbn


def node_attention_split(x, weight, bias):
    return _NodeAttentionSplit.apply(x, weight, bias)


class _AddPool(Function):
    """global_add_pool (model.py:115-116)."""

    @staticmethod
    def forward(ctx, x, plan: GraphPlan):
        x = _f32(x, "x")
        N, H = x.shape
        B, S = plan.B, plan.pool_splits()
        out = torch.empty(B, H, dtype=torch.float32, device=x.device)
        part = _empty(S * B * H, x.device) if S > 1 else None
        _c("cal_add_pool_fwd", x, plan.gptr, out, part, B, H, S)
        ctx.plan, ctx.N = plan, N
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _f32(dout, "grad")
        H = dout.size(1)
        dx = torch.empty(ctx.N, H, dtype=torch.float32, device=dout.device)
        _c("cal_add_pool_bwd", dout, ctx.plan.batch, dx, ctx.N, H)
        return dx, None


def add_pool(x, plan: GraphPlan):
    if plan.batch is None:
        raise ValueError("plan was built without a batch vector")
    return _AddPool.apply(x, plan)


class _GATAggregate(Function):
    """GATConv after z = x @ W: scores, edge softmax, dropout, aggregation, bias (+ReLU)."""

    @staticmethod
    def forward(ctx, z, att, bias, plan: GraphPlan, heads: int, slope: float, p: float, seed: int,
                relu: bool):
        z, att = _f32(z, "z"), _f32(att, "att")
        N, H = z.shape
        K = int(heads)
        D = H // K
        if att.numel() != K * 2 * D:
            raise ValueError("att must have heads * 2 * out_channels elements")
        if bias is not None:
            bias = _f32(bias, "bias")
        dev = z.device
        out = torch.empty_like(z)
        adst, asrc, mx, den = (_empty(N * K, dev) for _ in range(4))
        _c("cal_gat_fwd", plan.rowptr_dst, plan.nbr_dst, plan.eid_dst, z, att, bias,
                  int(relu), slope, p, seed, out, adst, asrc, mx, den, N, plan.E, K, D)
        ctx.plan, ctx.K, ctx.D, ctx.slope, ctx.p, ctx.seed, ctx.relu = plan, K, D, slope, p, seed, relu
        ctx.has_bias = bias is not None
        ctx.att_shape = att.shape
        ctx.save_for_backward(z, att, adst, asrc, mx, den, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, gout):
        z, att, adst, asrc, mx, den, out = ctx.saved_tensors
        plan, K, D = ctx.plan, ctx.K, ctx.D
        gout = _f32(gout, "grad")
        N, H = gout.shape
        dev = gout.device
        need_bias = ctx.has_bias and ctx.needs_input_grad[2]
        g = torch.empty_like(gout) if ctx.relu else gout
        dbias = torch.empty(H, dtype=torch.float32, device=dev) if need_bias else None
        if ctx.relu or need_bias:
            part = _empty(_q(gout, "cal_colsum_parts", N) * H, dev) if need_bias else None
            _c("cal_relu_bwd_colsum", gout, out if ctx.relu else None,
                      g if ctx.relu else None, dbias, part, N, H)
        dz = torch.empty_like(gout)
        datt = torch.empty(K * 2 * D, dtype=torch.float32, device=dev)
        ws = _empty(_q(gout, "cal_gat_bwd_ws", N, plan.E, K, D), dev)
        _c("cal_gat_bwd", plan.rowptr_dst, plan.nbr_dst, plan.eid_dst, plan.rowptr_src,
                  plan.nbr_src, plan.eid_src, z, att, adst, asrc, mx, den, g,
                  ctx.slope, ctx.p, ctx.seed, dz, datt, ws, N, plan.E, K, D)
        return dz, datt.view(ctx.att_shape), dbias, None, None, None, None, None, None


def gat_aggregate(z, att, bias, plan: GraphPlan, heads: int, negative_slope: float = 0.2,
                  dropout: float = 0.0, seed: int = 0, relu: bool = False):
    return _GATAggregate.apply(z, att, bias, plan, heads, negative_slope, dropout, seed, relu)


def gat_dropout_mask(seed: int, plan: GraphPlan, heads: int, p: float) -> torch.Tensor:
    """The keep mask ([E + N, heads] of 0/1) the kernels derive from ``seed`` (for tests)."""
    m = torch.empty(plan.E + plan.N, heads, dtype=torch.float32, device=plan.device)
    _c("cal_gat_dropout_mask", seed, plan.E, plan.N, heads, p, m)
    return m


class _Linear(Function):
    """Dense layer on the fp32 MFMA GEMM (cal_gemm): y = x @ W (+b) for a GCN-style weight
    [in, out] (gcn_conv.py:75) or y = x @ W^T (+b) for an nn.Linear weight [out, in]
    (model.py:46-75), optional fused ReLU."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_in: bool, relu: bool):
        x, weight = _f32(x, "x"), _f32(weight, "weight")
        M, K = x.shape
        N = weight.size(0) if out_in else weight.size(1)
        if (weight.size(1) if out_in else weight.size(0)) != K:
            raise ValueError("weight shape does not match the input width")
        if bias is not None:
            bias = _f32(bias, "bias")
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        _c("cal_gemm", 0, 1 if out_in else 0, x, weight, y, bias, int(relu), None, M, N, K)
        ctx.out_in, ctx.relu, ctx.has_bias = out_in, relu, bias is not None
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = _f32(gy, "grad")
        M, N = gy.shape
        K = x.size(1)
        dev = gy.device
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        g = torch.empty_like(gy) if ctx.relu else gy
        db = torch.empty(N, dtype=torch.float32, device=dev) if need_b else None
        if ctx.relu or need_b:
            part = _empty(_q(gy, "cal_colsum_parts", M) * N, dev) if need_b else None
            _c("cal_relu_bwd_colsum", gy, y if ctx.relu else None, g if ctx.relu else None,
                      db, part, M, N)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=torch.float32, device=dev)
            # out_in: dx = g @ W ([N,K] as stored);  else dx = g @ W^T (W stored [K,N])
            _c("cal_gemm", 0, 0 if ctx.out_in else 1, g, weight, dx, None, 0, None, M, K, N)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            if ctx.out_in:      # dW [N,K] = g^T @ x
                ws = _empty(_q(gy, "cal_gemm_ws", N, K, M), dev)
                _c("cal_gemm", 1, 0, g, x, dw, None, 0, ws, N, K, M)
            else:               # dW [K,N] = x^T @ g
                ws = _empty(_q(gy, "cal_gemm_ws", K, N, M), dev)
                _c("cal_gemm", 1, 0, x, g, dw, None, 0, ws, K, N, M)
        return dx, dw, db, None, None


def linear(x, weight, bias=None, relu: bool = False):
    """nn.Linear semantics (weight [out, in]) on the MFMA GEMM."""
    return _Linear.apply(x, weight, bias, True, relu)


def matmul(x, weight, relu: bool = False):
    """x @ weight with weight [in, out] (gcn_conv.py:75) on the MFMA GEMM."""
    return _Linear.apply(x, weight, None, False, relu)


class _GINAggregate(Function):
    """PyG GINConv's aggregation, (1 + eps) * x + sum_{j -> i} x_j (self loops of the input dropped),
    on the CSR aggregation kernel with unit coefficients (call site model.py:188)."""

    @staticmethod
    def forward(ctx, x, plan: GraphPlan, eps: float):
        x = _f32(x, "x")
        N, H = x.shape
        ones_n, ones_e = plan.ones()
        out = torch.empty_like(x)
        _c("cal_spmm_fwd", plan.rowptr_dst, plan.nbr_dst, plan.eid_dst, ones_e, ones_n,
                  1.0 + eps, x, None, 0, out, N, H)
        ctx.plan, ctx.eps = plan, eps
        return out

    @staticmethod
    def backward(ctx, g):
        g = _f32(g, "grad")
        N, H = g.shape
        plan = ctx.plan
        ones_n, ones_e = plan.ones()
        dx = torch.empty_like(g)
        _c("cal_spmm_fwd", plan.rowptr_src, plan.nbr_src, plan.eid_src, ones_e, ones_n,
                  1.0 + ctx.eps, g, None, 0, dx, N, H)
        return dx, None, None


def gin_aggregate(x, plan: GraphPlan, eps: float = 0.0):
    return _GINAggregate.apply(x, plan, eps)
