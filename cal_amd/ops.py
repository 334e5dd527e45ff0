"""autograd wrappers around the libcalhip entry points (operator-level boundary).

Every Function checks that its tensors live on the GPU and calls straight into
the C ABI on torch's current stream; there is no CPU implementation.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd import Function

from . import _lib
from .plan import GraphPlan, _p, _stream


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.CalError("%s: %s is on %s -- cal_amd kernels run on the GPU only (no CPU fallback)"
                            % ("cal_amd.ops", name, t.device))
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
            _lib.call("cal_gcn_norm_fwd", _p(plan.rowptr_src), _p(plan.eid_src), _p(plan.row32), _p(plan.col32),
                      _p(w), loop_w, N, plan.E, _p(dis), _p(norm), _stream())
        if bias is not None:
            bias = _f32(bias, "bias")
        out = torch.empty_like(h)
        _lib.call("cal_spmm_fwd", _p(plan.rowptr_dst), _p(plan.nbr_dst), _p(plan.eid_dst), _p(norm), _p(dis),
                  loop_w, _p(h), _p(bias), int(relu), _p(out), N, H, _stream())
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
            part = _empty(_lib.query("cal_colsum_parts", N) * H, dev) if need_bias else None
            _lib.call("cal_relu_bwd_colsum", _p(gout), _p(out) if ctx.relu else None,
                      _p(dz) if ctx.relu else None, _p(dbias), _p(part), N, H, _stream())
        dh = None
        if ctx.needs_input_grad[0]:
            dh = torch.empty_like(gout)
            _lib.call("cal_spmm_fwd", _p(plan.rowptr_src), _p(plan.nbr_src), _p(plan.eid_src), _p(norm), _p(dis),
                      loop_w, _p(dz), None, 0, _p(dh), N, H, _stream())
        dw = None
        if ctx.has_w and ctx.needs_input_grad[1]:
            dw = _empty(plan.E, dev)[:plan.E]
            gn, gself, ddeg = _empty(plan.E, dev), _empty(N, dev), _empty(N, dev)
            _lib.call("cal_gcn_norm_bwd", _p(plan.rowptr_dst), _p(plan.nbr_dst), _p(plan.eid_dst),
                      _p(plan.rowptr_src), _p(plan.nbr_src), _p(plan.eid_src), _p(plan.row32), _p(plan.col32),
                      _p(w), _p(dis), loop_w, _p(h), _p(dz), _p(gn), _p(gself), _p(ddeg), _p(dw),
                      N, plan.E, H, _stream())
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
        _lib.call("cal_edge_att_fwd", _p(x), _p(W), _p(b), _p(plan.row32), _p(plan.col32), _p(pq), _p(att),
                  N, plan.E, H, _stream())
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
        ws = _empty(_lib.query("cal_edge_att_bwd_ws", N, plan.E, H), x.device)
        _lib.call("cal_edge_att_bwd", _p(x), _p(W), _p(att), _p(datt), _p(plan.rowptr_src), _p(plan.eid_src),
                  _p(plan.rowptr_dst), _p(plan.eid_dst), _p(dx), 0, _p(dW), _p(db), _p(ws), N, plan.E, H, _stream())
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
        _lib.call("cal_node_att_split_fwd", _p(x), _p(Wn), _p(bn), _p(att), _p(xc), _p(xo), N, H, _stream())
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
        ws = _empty(_lib.query("cal_node_att_bwd_ws", N, H), x.device)
        _lib.call("cal_node_att_split_bwd", _p(x), _p(Wn), _p(att), _p(dxc), _p(dxo), _p(dx), _p(dWn), _p(dbn),
                  _p(ws), N, H, _stream())
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
        _lib.call("cal_add_pool_fwd", _p(x), _p(plan.gptr), _p(out), _p(part), B, H, S, _stream())
        ctx.plan, ctx.N = plan, N
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _f32(dout, "grad")
        H = dout.size(1)
        dx = torch.empty(ctx.N, H, dtype=torch.float32, device=dout.device)
        _lib.call("cal_add_pool_bwd", _p(dout), _p(ctx.plan.batch), _p(dx), ctx.N, H, _stream())
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
        _lib.call("cal_gat_fwd", _p(plan.rowptr_dst), _p(plan.nbr_dst), _p(plan.eid_dst), _p(z), _p(att), _p(bias),
                  int(relu), slope, p, seed, _p(out), _p(adst), _p(asrc), _p(mx), _p(den), N, plan.E, K, D, _stream())
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
            part = _empty(_lib.query("cal_colsum_parts", N) * H, dev) if need_bias else None
            _lib.call("cal_relu_bwd_colsum", _p(gout), _p(out) if ctx.relu else None,
                      _p(g) if ctx.relu else None, _p(dbias), _p(part), N, H, _stream())
        dz = torch.empty_like(gout)
        datt = torch.empty(K * 2 * D, dtype=torch.float32, device=dev)
        ws = _empty(_lib.query("cal_gat_bwd_ws", N, plan.E, K, D), dev)
        _lib.call("cal_gat_bwd", _p(plan.rowptr_dst), _p(plan.nbr_dst), _p(plan.eid_dst), _p(plan.rowptr_src),
                  _p(plan.nbr_src), _p(plan.eid_src), _p(z), _p(att), _p(adst), _p(asrc), _p(mx), _p(den), _p(g),
                  ctx.slope, ctx.p, ctx.seed, _p(dz), _p(datt), _p(ws), N, plan.E, K, D, _stream())
        return dz, datt.view(ctx.att_shape), dbias, None, None, None, None, None, None


def gat_aggregate(z, att, bias, plan: GraphPlan, heads: int, negative_slope: float = 0.2,
                  dropout: float = 0.0, seed: int = 0, relu: bool = False):
    return _GATAggregate.apply(z, att, bias, plan, heads, negative_slope, dropout, seed, relu)


def gat_dropout_mask(seed: int, plan: GraphPlan, heads: int, p: float) -> torch.Tensor:
    """The keep mask ([E + N, heads] of 0/1) the kernels derive from ``seed`` (for tests)."""
    m = torch.empty(plan.E + plan.N, heads, dtype=torch.float32, device=plan.device)
    _lib.call("cal_gat_dropout_mask", seed, plan.E, plan.N, heads, p, _p(m), _stream())
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
        _lib.call("cal_gemm", 0, 1 if out_in else 0, _p(x), _p(weight), _p(y), _p(bias), int(relu), None, M, N, K, _stream())
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
            part = _empty(_lib.query("cal_colsum_parts", M) * N, dev) if need_b else None
            _lib.call("cal_relu_bwd_colsum", _p(gy), _p(y) if ctx.relu else None, _p(g) if ctx.relu else None,
                      _p(db), _p(part), M, N, _stream())
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=torch.float32, device=dev)
            # out_in: dx = g @ W ([N,K] as stored);  else dx = g @ W^T (W stored [K,N])
            _lib.call("cal_gemm", 0, 0 if ctx.out_in else 1, _p(g), _p(weight), _p(dx), None, 0, None, M, K, N, _stream())
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            if ctx.out_in:      # dW [N,K] = g^T @ x
                ws = _empty(_lib.query("cal_gemm_ws", N, K, M), dev)
                _lib.call("cal_gemm", 1, 0, _p(g), _p(x), _p(dw), None, 0, _p(ws), N, K, M, _stream())
            else:               # dW [K,N] = x^T @ g
                ws = _empty(_lib.query("cal_gemm_ws", K, N, M), dev)
                _lib.call("cal_gemm", 1, 0, _p(x), _p(g), _p(dw), None, 0, _p(ws), K, N, M, _stream())
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
        _lib.call("cal_spmm_fwd", _p(plan.rowptr_dst), _p(plan.nbr_dst), _p(plan.eid_dst), _p(ones_e), _p(ones_n),
                  1.0 + eps, _p(x), None, 0, _p(out), N, H, _stream())
        ctx.plan, ctx.eps = plan, eps
        return out

    @staticmethod
    def backward(ctx, g):
        g = _f32(g, "grad")
        N, H = g.shape
        plan = ctx.plan
        ones_n, ones_e = plan.ones()
        dx = torch.empty_like(g)
        _lib.call("cal_spmm_fwd", _p(plan.rowptr_src), _p(plan.nbr_src), _p(plan.eid_src), _p(ones_e), _p(ones_n),
                  1.0 + ctx.eps, _p(g), None, 0, _p(dx), N, H, _stream())
        return dx, None, None


def gin_aggregate(x, plan: GraphPlan, eps: float = 0.0):
    return _GINAggregate.apply(x, plan, eps)
