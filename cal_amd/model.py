"""CausalGCN / CausalGAT / CausalGIN with the reference's nn.Module surface
(model.py:12-450): same constructor arguments (``args`` namespace with
``layers, hidden, with_random, without_node_attention,
without_edge_attention, fc_num, cat_or_add``), same sub-module / state-dict
names, same ``forward(data, eval_random=True) -> (xc_logis, xo_logis,
xco_logis)`` log-probabilities, so ``opts.get_model`` (opts.py:85-119) and the
loops in train_causal.py work unchanged.

The graph operators run on libcalhip through ``cal_amd.ops``; one GraphPlan is
built per batch and shared by every layer.
"""
from __future__ import annotations

import os
import random
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import BatchNorm1d, Linear, ReLU, Sequential

from . import ops
from .gat_conv import GATConv
from .gcn_conv import GCNConv
from .plan import plan_of


class _CausalBase(torch.nn.Module):
    """Everything after the backbone: soft masks, causal/trivial convs, pooling,
    three readouts (model.py:46-75, 97-164)."""

    #: CausalGCN gates the shuffle on args.with_random as well (model.py:149-151);
    #: CausalGAT / CausalGIN do not (model.py:435-436, 298-299).
    _gate_on_with_random = False

    def _build_head(self, hidden, hidden_out, GConv):
        self.edge_att_mlp = nn.Linear(hidden * 2, 2)
        self.node_att_mlp = nn.Linear(hidden, 2)
        self.bnc = BatchNorm1d(hidden)
        self.bno = BatchNorm1d(hidden)
        self.context_convs = GConv(hidden, hidden)
        self.objects_convs = GConv(hidden, hidden)
        # context mlp
        self.fc1_bn_c = BatchNorm1d(hidden)
        self.fc1_c = Linear(hidden, hidden)
        self.fc2_bn_c = BatchNorm1d(hidden)
        self.fc2_c = Linear(hidden, hidden_out)
        # object mlp
        self.fc1_bn_o = BatchNorm1d(hidden)
        self.fc1_o = Linear(hidden, hidden)
        self.fc2_bn_o = BatchNorm1d(hidden)
        self.fc2_o = Linear(hidden, hidden_out)
        # random mlp
        if self.args.cat_or_add == "cat":
            self.fc1_bn_co = BatchNorm1d(hidden * 2)
            self.fc1_co = Linear(hidden * 2, hidden)
        elif self.args.cat_or_add == "add":
            self.fc1_bn_co = BatchNorm1d(hidden)
            self.fc1_co = Linear(hidden, hidden)
        else:
            assert False
        self.fc2_bn_co = BatchNorm1d(hidden)
        self.fc2_co = Linear(hidden, hidden_out)

    def _init_bn(self):
        for m in self.modules():                                   # model.py:80-83
            if isinstance(m, torch.nn.BatchNorm1d):
                torch.nn.init.constant_(m.weight, 1)
                torch.nn.init.constant_(m.bias, 0.0001)

    # -- hooks -----------------------------------------------------------------
    def _backbone(self, x, edge_index, plan):
        raise NotImplementedError

    #: CausalGCN / CausalGAT: run forward/backward on the native step engine (cal_amd/csrc/engine.hip)
    #: behind this same nn.Module / autograd surface.  Set False for the operator-level path.
    use_engine = os.environ.get("CAL_AMD_ENGINE", "1") != "0"

    def engine(self):
        """The model's StepEngine (created on first use) when its parameters are on the GPU and the engine covers this
        variant; else ``None``."""
        p0 = next(self.parameters())
        return self._engine_for(p0)

    def _engine_for(self, x):
        from . import engine as eng_mod
        if not (self.use_engine and x.is_cuda and isinstance(self, (CausalGCN, CausalGAT, CausalGIN)) and eng_mod.supported(self)):
            return None
        eng = getattr(self, "_engine", None)
        p0 = next(self.parameters())
        if eng is not None:
            lo = eng.flat_p.data_ptr()
            if not (lo <= p0.data_ptr() < lo + 4 * eng.flat_p.numel()):
                eng = None                                   # parameters were moved (.to/.cuda): rebuild
        if eng is None:
            eng = eng_mod.StepEngine(self)
            object.__setattr__(self, "_engine", eng)
        return eng

    def forward(self, data, eval_random=True, perm=None, _objects_raw=False):
        x = data.x if data.x is not None else data.feat
        # (_objects_raw: CausalGIN's train_type = "irm" also returns the objects head's raw logits -- operator-level path)
        eng = None if _objects_raw else self._engine_for(x)
        if eng is not None:
            from .engine import engine_forward_autograd
            if perm is None:
                perm = eng.perm_stage().put(self.intervention_list(int(data.num_graphs), eval_random))
            else:
                perm = perm.to(x.device) if perm.is_cuda else eng.perm_stage().put(perm)  # (pinned ring: a pageable H2D copy blocks the host)
            if self.training and torch.is_grad_enabled():
                return engine_forward_autograd(eng, data, perm)
            eng.forward(data, perm, training=self.training)
            return eng.logp_copy()
        edge_index = data.edge_index
        plan = plan_of(data)
        x = self.bn_feat(x)
        x = self.conv_feat(x, edge_index, relu=True)
        x = self._backbone(x, edge_index, plan)

        if getattr(self, "without_edge_attention", False):
            edge_att = torch.full((2, plan.E), 0.5, dtype=x.dtype, device=x.device)   # model.py:100
        else:
            edge_att = ops.edge_attention(x, self.edge_att_mlp.weight, self.edge_att_mlp.bias, plan)
        edge_weight_c, edge_weight_o = edge_att[0], edge_att[1]

        if getattr(self, "without_node_attention", False):
            xc = 0.5 * x                                                                # model.py:107-111
            xo = 0.5 * x
        else:
            xc, xo, _ = ops.node_attention_split(x, self.node_att_mlp.weight, self.node_att_mlp.bias)
        xc = self.context_convs(self.bnc(xc), edge_index, edge_weight_c, plan=plan, relu=True)
        xo = self.objects_convs(self.bno(xo), edge_index, edge_weight_o, plan=plan, relu=True)

        xc = ops.add_pool(xc, plan)
        xo = ops.add_pool(xo, plan)

        xc_logis = self.context_readout_layer(xc)
        xo_logis = self.objects_readout_layer(xo, "irm" if _objects_raw else "base")
        xco_logis = self.random_readout_layer(xc, xo, eval_random=eval_random, perm=perm)
        return xc_logis, xo_logis, xco_logis

    def context_readout_layer(self, x):
        x = self.fc1_bn_c(x)
        x = ops.linear(x, self.fc1_c.weight, self.fc1_c.bias, relu=True)     # Linear + ReLU on the MFMA GEMM
        x = self.fc2_bn_c(x)
        x = ops.linear(x, self.fc2_c.weight, self.fc2_c.bias)
        return F.log_softmax(x, dim=-1)

    def objects_readout_layer(self, x, train_type="base"):
        """model.py:136-143; CausalGIN's variant (model.py:281-292) also hands back the raw logits for train_type = "irm"."""
        x = self.fc1_bn_o(x)
        x = ops.linear(x, self.fc1_o.weight, self.fc1_o.bias, relu=True)     # Linear + ReLU on the MFMA GEMM
        x = self.fc2_bn_o(x)
        x = ops.linear(x, self.fc2_o.weight, self.fc2_o.bias)
        if train_type == "irm":
            return x, F.log_softmax(x, dim=-1)
        return F.log_softmax(x, dim=-1)

    def intervention_list(self, num, eval_random):
        """model.py:147-152: Python ``random.shuffle`` of range(num), gated as the reference gates it (the list itself)."""
        l = [i for i in range(num)]
        gate = eval_random and (self.with_random if self._gate_on_with_random else True)
        if gate:
            random.shuffle(l)
        return l

    def intervention_index(self, num, eval_random):
        return torch.tensor(self.intervention_list(num, eval_random))

    def random_readout_layer(self, xc, xo, eval_random, perm=None):
        num = xc.shape[0]
        random_idx = self.intervention_index(num, eval_random) if perm is None else perm
        random_idx = random_idx.to(xc.device)
        if self.args.cat_or_add == "cat":
            x = torch.cat((xc[random_idx], xo), dim=1)
        else:
            x = xc[random_idx] + xo
        x = self.fc1_bn_co(x)
        x = ops.linear(x, self.fc1_co.weight, self.fc1_co.bias, relu=True)     # Linear + ReLU on the MFMA GEMM
        x = self.fc2_bn_co(x)
        x = ops.linear(x, self.fc2_co.weight, self.fc2_co.bias)
        return F.log_softmax(x, dim=-1)


class CausalGCN(_CausalBase):
    """model.py:12-164."""
    _gate_on_with_random = True

    def __init__(self, num_features, num_classes, args, gfn=False, collapse=False, residual=False,
                 res_branch="BNConvReLU", global_pool="sum", dropout=0, edge_norm=True):
        super().__init__()
        num_conv_layers = args.layers
        hidden = args.hidden
        self.args = args
        self.dropout = dropout
        self.with_random = args.with_random
        self.without_node_attention = args.without_node_attention
        self.without_edge_attention = args.without_edge_attention
        GConv = partial(GCNConv, edge_norm=edge_norm, gfn=gfn)
        self.num_classes = num_classes
        self.fc_num = args.fc_num
        self.bn_feat = BatchNorm1d(num_features)
        self.conv_feat = GCNConv(num_features, hidden, gfn=True)
        self.bns_conv = torch.nn.ModuleList()
        self.convs = torch.nn.ModuleList()
        for _ in range(num_conv_layers):
            self.bns_conv.append(BatchNorm1d(hidden))
            self.convs.append(GConv(hidden, hidden))
        self._build_head(hidden, num_classes, GConv)
        self._init_bn()

    def _backbone(self, x, edge_index, plan):
        for i, conv in enumerate(self.convs):                     # model.py:93-95
            x = self.bns_conv[i](x)
            x = conv(x, edge_index, plan=plan, relu=True)
        return x


class CausalGAT(_CausalBase):
    """model.py:315-450."""

    def __init__(self, num_features, num_classes, args, head=4, dropout=0.2):
        super().__init__()
        num_conv_layers = args.layers
        hidden = args.hidden
        self.args = args
        self.dropout = dropout
        self.with_random = getattr(args, "with_random", True)
        GConv = partial(GCNConv, edge_norm=True, gfn=False)
        self.num_classes = num_classes
        self.fc_num = args.fc_num
        self.bn_feat = BatchNorm1d(num_features)
        self.conv_feat = GCNConv(num_features, hidden, gfn=True)
        self.bns_conv = torch.nn.ModuleList()
        self.convs = torch.nn.ModuleList()
        for _ in range(num_conv_layers):
            self.bns_conv.append(BatchNorm1d(hidden))
            self.convs.append(GATConv(hidden, int(hidden / head), heads=head, dropout=dropout))
        self._build_head(hidden, num_classes, GConv)
        self._init_bn()

    def _backbone(self, x, edge_index, plan):
        for i, conv in enumerate(self.convs):                     # model.py:388-390
            x = self.bns_conv[i](x)
            x = conv(x, edge_index, plan=plan, relu=True)
        return x


class GINConv(torch.nn.Module):
    """PyG GINConv(nn): nn((1 + eps) * x + sum_j x_j), eps = 0 fixed (model.py:188)."""

    def __init__(self, nn_module, eps=0.0):
        super().__init__()
        self.nn = nn_module
        self.initial_eps = float(eps)
        self.register_buffer("eps", torch.tensor([float(eps)]))     # PyG keeps eps in the state dict

    def forward(self, x, edge_index, *, plan=None):
        from .plan import GraphPlan
        if plan is None:
            plan = GraphPlan(edge_index, x.size(0))
        out = ops.gin_aggregate(x, plan, self.initial_eps)
        lin1, bn, _, lin2, _ = self.nn                               # model.py:189-194
        out = ops.linear(out, lin1.weight, lin1.bias)
        out = torch.relu(bn(out))
        return ops.linear(out, lin2.weight, lin2.bias, relu=True)


class CausalGIN(_CausalBase):
    """model.py:166-313 (SURVEY.md section 8f "next")."""

    def __init__(self, num_features, num_classes, args, gfn=False, edge_norm=True):
        super().__init__()
        hidden = args.hidden
        self.args = args
        self.with_random = getattr(args, "with_random", True)
        GConv = partial(GCNConv, edge_norm=edge_norm, gfn=gfn)
        self.num_classes = num_classes
        self.fc_num = args.fc_num
        self.bn_feat = BatchNorm1d(num_features)
        self.conv_feat = GCNConv(num_features, hidden, gfn=True)
        self.bns_conv = torch.nn.ModuleList()
        self.convs = torch.nn.ModuleList()
        for _ in range(args.layers):
            self.convs.append(GINConv(Sequential(Linear(hidden, hidden), BatchNorm1d(hidden), ReLU(),
                                                 Linear(hidden, hidden), ReLU())))
        self._build_head(hidden, num_classes, GConv)
        self._init_bn()

    def _backbone(self, x, edge_index, plan):
        for conv in self.convs:                                   # model.py:244-245
            x = conv(x, edge_index, plan=plan)
        return x

    def forward(self, data, eval_random=True, train_type="base", perm=None):
        """model.py:234-264: ``forward(data, eval_random=True, train_type="base")``.  ``train_type="irm"`` makes the objects
        head return ``(raw logits, log-probs)`` instead of the log-probs (model.py:281-292) -- dead in the reference's own loops
        (train_causal.py:177,212 never pass it) but part of the module surface; that variant runs on the operator-level path,
        where the raw logits are differentiable."""
        if torch.is_tensor(train_type):                               # (a caller of the base-class order: forward(data, eval_random, perm))
            perm, train_type = train_type, "base"
        return super().forward(data, eval_random, perm=perm, _objects_raw=(train_type == "irm"))
