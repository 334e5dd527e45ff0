"""Python handle of the native CausalGCN / CausalGAT step engine (cal_amd/csrc/engine.hip).

``StepEngine(model)`` re-homes the module's parameters into one flat buffer
(so the module, its state_dict and any torch optimizer keep working on the same
memory), binds gradients / Adam state, owns the device workspace and exposes

* ``forward(batch, perm, training)`` -> three ``[B, C]`` log-prob tensors,
* ``train_step(batch, perm)``       -> forward + 3-term loss + backward (+ Adam),

each as ONE C call that enqueues the fused kernels on torch's current stream
(hipGraph-capturable).  ``supported(model)`` says what is covered: CausalGCN and
CausalGAT, ``cat_or_add`` "add" or "cat", with or without the node / edge
attention and CausalGIN (every causal variant ``opts.get_model`` builds).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib
from .plan import _p, _stream

_BN_ORDER_TAIL = ["bnc", "bno", "fc1_bn_c", "fc2_bn_c", "fc1_bn_o", "fc2_bn_o", "fc1_bn_co", "fc2_bn_co"]


def _gat_heads(model) -> int:
    """Heads of a GATConv backbone (CausalGAT, model.py:340,390); 0 for the GCNConv backbone."""
    from .gat_conv import GATConv
    convs = list(model.convs)
    return int(convs[0].heads) if convs and isinstance(convs[0], GATConv) else 0


def _is_gin(model) -> bool:
    from .model import CausalGIN
    return isinstance(model, CausalGIN)


def supported(model) -> bool:
    from .model import CausalGCN, CausalGAT, CausalGIN, GINConv
    from .gcn_conv import GCNConv
    a = model.args
    h = a.hidden
    if not (isinstance(model, (CausalGCN, CausalGAT, CausalGIN)) and a.cat_or_add in ("add", "cat")
            and h % 4 == 0 and h <= 256 and a.layers <= 6 and model.num_classes <= 64):
        return False
    # the engine hard-wires the normalised, non-improved GCNConv with a bias (gcn_conv.py:72-92 defaults): a model
    # built with gfn=True / edge_norm=False / improved=True keeps the operator-level path
    gcn_like = [model.context_convs, model.objects_convs] + [c for c in model.convs if isinstance(c, GCNConv)]
    if any(c.gfn or not c.edge_norm or c.improved or c.bias is None for c in gcn_like):
        return False
    if not (model.conv_feat.gfn and model.conv_feat.bias is not None):
        return False
    if isinstance(model, CausalGAT):
        k = _gat_heads(model)
        d = h // k if k else 0
        return (k > 0 and h % k == 0 and d % 4 == 0 and (d // 4) & (d // 4 - 1) == 0
                and all(c.heads == k and c.bias is not None for c in model.convs))
    if isinstance(model, CausalGIN):
        # GINConv(Sequential(Linear, BatchNorm1d, ReLU, Linear, ReLU)), eps = 0 (model.py:188-194)
        return all(isinstance(c, GINConv) and c.initial_eps == 0.0 and len(c.nn) == 5 and c.nn[0].bias is not None
                   and c.nn[3].bias is not None for c in model.convs)
    return True


def _slot_names(layers: int, gin: bool = False):
    names = ["bn_feat.weight", "bn_feat.bias", "conv_feat.weight"]
    for i in range(layers):
        if gin:     # cal_engine_set_gin: BatchNorm, first Linear, second Linear of GINConv's Sequential (model.py:189-194)
            names += [f"convs.{i}.nn.1.weight", f"convs.{i}.nn.1.bias", f"convs.{i}.nn.0.weight", f"convs.{i}.nn.0.bias",
                      f"convs.{i}.nn.3.weight", f"convs.{i}.nn.3.bias"]
        else:
            names += [f"bns_conv.{i}.weight", f"bns_conv.{i}.bias", f"convs.{i}.weight", f"convs.{i}.bias"]
    names += ["edge_att_mlp.weight", "edge_att_mlp.bias", "node_att_mlp.weight", "node_att_mlp.bias",
              "bnc.weight", "bnc.bias", "bno.weight", "bno.bias",
              "context_convs.weight", "context_convs.bias", "objects_convs.weight", "objects_convs.bias"]
    for h in ("c", "o", "co"):
        names += [f"fc1_bn_{h}.weight", f"fc1_bn_{h}.bias", f"fc1_{h}.weight", f"fc1_{h}.bias",
                  f"fc2_bn_{h}.weight", f"fc2_bn_{h}.bias", f"fc2_{h}.weight", f"fc2_{h}.bias"]
    return names


def _layout_of(batch, B: int) -> dict:
    """Layout facts the per-graph kernels need -- node / edge ranges per graph, the largest graph, no self loops.
    ``cal_amd.data.Batch`` and ``DeviceDataset.collate`` record them while collating; a foreign batch exposing only
    the reference's protocol (``x``/``feat``, ``edge_index``, sorted ``batch``, ``num_graphs`` -- SURVEY.md 8b; PyG >= 2
    adds ``ptr``) gets them derived here once on the device (one host read-back for the two bounds) and cached on the
    object, instead of silently falling to the unfused kernels."""
    have = all(hasattr(batch, k) for k in ("max_nodes", "max_edges", "edge_ptr", "no_self_loops")) and hasattr(batch, "ptr")
    if have and (int(batch.max_nodes or 0) > 0 or B == 0):
        return dict(ptr=batch.ptr, edge_ptr=batch.edge_ptr, no_self_loops=bool(batch.no_self_loops),
                    max_nodes=int(batch.max_nodes or 0), max_edges=int(batch.max_edges or 0))
    ei, bvec = batch.edge_index, batch.batch
    key = (ei.data_ptr(), bvec.data_ptr(), int(ei.size(1)), int(bvec.numel()), B)
    lay = getattr(batch, "_cal_layout", None)
    if lay is not None and lay["key"] == key:
        return lay
    dev = bvec.device
    counts = torch.bincount(bvec, minlength=B)[:B] if bvec.numel() else torch.zeros(B, dtype=torch.long, device=dev)
    ptr = torch.zeros(B + 1, dtype=torch.long, device=dev)
    ptr[1:] = torch.cumsum(counts, 0)
    lay = dict(key=key, ptr=ptr, edge_ptr=None, no_self_loops=False, max_nodes=0, max_edges=0)
    if ei.size(1) > 0 and B > 0:
        gid = bvec[ei[0]]
        grouped = bool((gid[1:] >= gid[:-1]).all().item()) and bool((bvec[ei[1]] == gid).all().item())
        if grouped:                                       # graph b owns a contiguous run of edge columns
            eptr = torch.searchsorted(gid, torch.arange(B + 1, device=dev, dtype=torch.long)).contiguous()
            sizes = torch.stack([counts.max(), (eptr[1:] - eptr[:-1]).max(), (ei[0] == ei[1]).any().long()]).tolist()
            lay.update(edge_ptr=eptr, max_nodes=int(sizes[0]), max_edges=int(sizes[1]), no_self_loops=not sizes[2])
    elif B > 0:
        lay.update(edge_ptr=torch.zeros(B + 1, dtype=torch.long, device=dev), no_self_loops=True,
                   max_nodes=int(counts.max().item()), max_edges=0)
    try:
        batch._cal_layout = lay
    except Exception:
        pass
    return lay


class StepEngine:
    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, flat=None, deterministic: Optional[bool] = None):
        """``deterministic``: every cross-row sum (BatchNorm statistics, their backward sums) is added in a fixed order -- steps are
        bit-reproducible from run to run and across identical data-parallel replicas.  The default path adds those sums with fp64
        atomics into four accumulator rows (order-dependent in the last bits, ~1e-16 relative; a few launches per step faster).
        ``None`` follows ``torch.are_deterministic_algorithms_enabled()``; the environment variable CAL_AMD_STRIPED=0 selects
        the same mode process-wide."""
        if not supported(model):
            raise ValueError("StepEngine covers CausalGCN / CausalGAT / CausalGIN (hidden % 4 == 0, <= 256, <= 6 layers, <= 64 classes)")
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise _lib.CalError("StepEngine needs the model on the GPU (no CPU fallback)")
        self.model = model
        self.device = p0.device
        a = model.args
        self.F = model.bn_feat.num_features
        self.H, self.C, self.L = a.hidden, model.num_classes, a.layers
        from .trainer import flatten_parameters
        self.flat_p, self.flat_g = flat if flat is not None else flatten_parameters(model)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.lr = torch.full((1,), float(lr), dtype=torch.float32, device=self.device)
        h = _lib.lib().cal_engine_create(self.F, self.H, self.C, self.L)
        if not h:
            raise _lib.CalError("cal_engine_create failed: " + _lib.lib().cal_last_error().decode())
        self._h = ctypes.c_void_p(h)
        if deterministic is None:
            deterministic = torch.are_deterministic_algorithms_enabled()
        self.deterministic = bool(deterministic)
        if self.deterministic:
            _lib.call("cal_engine_set_deterministic", self._h, 1)
        _loss_flag(self.device)              # (the fused loss's label flag exists before any step is captured)
        # model variants: cat readout (2H-wide co head), the two ablation flags (model.py:65-69,99-107)
        _lib.call("cal_engine_set_options", self._h, int(a.cat_or_add == "cat"),
                  int(bool(getattr(model, "without_node_attention", False))),
                  int(bool(getattr(model, "without_edge_attention", False))))
        self.gin = _is_gin(model)
        if self.gin:
            _lib.call("cal_engine_set_gin", self._h, 1)
        # parameter offsets in slot order
        base = self.flat_p.data_ptr()
        params = dict(model.named_parameters())
        offs = []
        for n in _slot_names(self.L, self.gin):
            p = params[n]
            off = (p.data_ptr() - base) // 4
            assert 0 <= off and off + p.numel() <= self.flat_p.numel(), n
            offs.append(off)
        assert len(offs) == _lib.query("cal_engine_num_param_slots", self._h)
        backbone_bns = [c.nn[1] for c in model.convs] if self.gin else list(model.bns_conv)
        bns = [model.bn_feat] + backbone_bns + [getattr(model, n) for n in _BN_ORDER_TAIL]
        ptrs = []
        for bn in bns:
            ptrs += [bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr()]
        self._offs = (ctypes.c_int64 * len(offs))(*offs)
        self._bn = (ctypes.c_int64 * len(ptrs))(*ptrs)
        _lib.call("cal_engine_bind", self._h, _p(self.flat_p), _p(self.flat_g), _p(self.exp_avg), _p(self.exp_avg_sq),
                  _p(self.step_count), _p(self.lr), self.flat_p.numel(), self._offs, self._bn,
                  betas[0], betas[1], eps, weight_decay)
        # GATConv backbone: att offsets, dropout seeds, the per-step device counter the masks are keyed by
        self.heads = _gat_heads(model)
        self._gat_seeds = None
        if self.heads:
            self.gat_ctr = torch.zeros(1, dtype=torch.int64, device=self.device)
            self._gat_base = int(torch.initial_seed() * 1000003 + 7919) & ((1 << 62) - 1)
            att = []
            for conv in model.convs:
                off = (conv.att.data_ptr() - base) // 4
                assert 0 <= off and off + conv.att.numel() <= self.flat_p.numel()
                att.append(off)
            self._att_offs = (ctypes.c_int64 * len(att))(*att)
            self._sync_gat()
        #: False withholds the batch layout from the engine: generic CSR build + unfused GEMM / aggregation kernels
        #: (what a batch too large for the per-graph kernels runs anyway); tests compare the two paths with it
        self.fused = True
        #: False keeps one graph per workgroup for batches of small graphs too (tests compare the two)
        #: True: pack when it pays (below); "force": whenever the batch carries tiles (tests); False: never
        self.tiles = {"0": False, "force": "force"}.get(os.environ.get("CAL_AMD_TILES", "1"), True)
        self.tile_min_units = 256
        self._tiles = (0, 0)
        self._identity = {}             # identity permutations by batch size (eval passes, models that do not shuffle)
        self._ws: Optional[torch.Tensor] = None
        self.ws_generation = 0          # bumped whenever the workspace is re-allocated (captured graphs check it)
        self._cap = (0, 0, 0)
        self._bounds = (0, 0)
        self._ptrs = (0, 0)
        self.wc, self.wo, self.wco = float(getattr(a, "c", 0.5)), float(getattr(a, "o", 1.0)), float(getattr(a, "co", 0.5))
        from .optim import register_engine
        register_engine(self)

    def set_adam(self, beta1: float, beta2: float, eps: float, weight_decay: float):
        """Adam hyper-parameters of an attached optimizer object (cal_amd/optim.py); the learning rate is ``self.lr``."""
        _lib.call("cal_engine_set_adam", self._h, float(beta1), float(beta2), float(eps), float(weight_decay))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().cal_engine_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _sync_gat(self):
        """(Re)send the GAT settings when a layer's fixed dropout seed (GATConv.seed, tests) or p changed.  With no
        fixed seed the masks are keyed by (base seed + layer, device step counter): fresh per step even inside a
        replayed hipGraph; with fixed seeds the counter is off and layer i uses exactly GATConv.seed."""
        convs = list(self.model.convs)
        key = tuple((c.seed, float(c.dropout), float(c.negative_slope)) for c in convs)
        if key == self._gat_seeds:
            return
        fixed = any(c.seed is not None for c in convs)
        seeds = [int(c.seed) if c.seed is not None else (self._gat_base + 0x9E3779B1 * (i + 1)) & ((1 << 63) - 1)
                 for i, c in enumerate(convs)]
        arr = (ctypes.c_uint64 * len(seeds))(*seeds)
        _lib.call("cal_engine_set_gat", self._h, self.heads, float(convs[0].dropout), float(convs[0].negative_slope),
                  self._att_offs, arr, None if fixed else _p(self.gat_ctr))
        self._gat_seeds = key
        self.gat_layer_seeds = seeds
        self.gat_fixed = fixed

    # ---------------------------------------------------------------- workspace
    def reserve(self, N: int, E: int, B: int):
        """(Re)allocate the workspace for batches up to (N nodes, E edges, B graphs)."""
        cn, ce, cb = self._cap
        if N <= cn and E <= ce and B <= cb and self._ws is not None:
            return
        N, E, B = max(N, cn), max(E, ce), max(B, cb)
        # what the steps run so far flagged travels with us (the old status word dies with the old workspace; a
        # synchronising read, but growth is rare), and captured graphs of the old workspace are stale from here on
        sticky = 0
        if self._ws is not None:
            st = self.buffer("status", 4, torch.int32)
            sticky = int((st[0] | st[1]).item())
        self.ws_generation += 1
        nbytes = _lib.query("cal_engine_workspace_bytes", self._h, N, E, B)
        self._ws = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=self.device)
        assert self._ws.data_ptr() % 256 == 0
        _lib.call("cal_engine_set_workspace", self._h, _p(self._ws), nbytes, N, E, B)
        self._cap = (N, E, B)
        self.buffer("status", 4, torch.int32).zero_()       # [0] = latest step, [1] = sticky OR of the earlier steps
        if sticky:
            self.buffer("status", 4, torch.int32)[1] = sticky
        if self.gin:
            self.buffer("ones", N).fill_(1.0)                 # unit aggregation coefficients of the GINConv layers

    _STATUS_BITS = ((1, "edge_index has entries outside [0, num_nodes)"),
                    (2, "batch vector is not sorted / has ids outside [0, num_graphs), or ptr / edge_ptr do not match it"),
                    (8, "a graph exceeds the per-graph bounds (max_nodes / max_edges) the batch declared"),
                    (16, "an edge leaves its graph's node range (edge_ptr / ptr are stale)"),
                    (32, "edge_index has self loops although the batch declared no_self_loops"),
                    (64, "the peer-memory gradient exchange timed out waiting for another rank"),
                    (256, "the one-launch readout timed out at an in-kernel barrier (its workgroups were not co-resident: "
                          "another process is holding CUs; CAL_AMD_RO_STEP=0 runs the readout as separate launches)"),
                    (512, "a captured train step was replayed behind a training-mode forward that had no backward (the step's "
                          "BatchNorm accumulators were not clean: nothing was updated; the next step runs normally)"))

    def peek_status(self) -> int:
        """The status words as the latest COMPLETED training step left them (host-mapped mirror written by the step's last
        kernel: no synchronisation, cheap enough for every step).  Non-zero: call ``check_status`` for the message."""
        return _lib.query("cal_engine_peek_status", self._h)

    def check_status(self, reset: bool = True):
        """Synchronising check of the device status word over every step since the last check: the per-graph kernels
        skip graphs that violate the layout facts the batch declared (``max_nodes``, ``max_edges``, ``ptr``,
        ``edge_ptr``, ``no_self_loops``) and flag it there, so a stale attribute would otherwise train on garbage
        silently.  Called where the loops already synchronise (per-epoch statistics read-back, evaluation)."""
        if self._ws is None:
            return
        st = self.buffer("status", 4, torch.int32)
        word = int((st[0] | st[1]).item())
        if reset:
            st[:2].zero_()
        if word:
            for cb in list(getattr(self, "_on_flag", ())):
                cb()                           # (optim.Binding: the flagged steps were not applied -- re-read the device step counter)
            msgs = [m for bit, m in self._STATUS_BITS if word & bit] or ["unknown status bits"]
            raise _lib.CalError("cal_amd engine: invalid batch (status 0x%x): %s" % (word, "; ".join(msgs)))
        # the fused causal loss of statement-by-statement loops flags labels outside [0, C) in a word of its own (advisor, round 5:
        # only train_causal's epoch read it; every caller of check_status -- trainer read-backs, evaluation -- now does)
        check_loss_labels()

    def buffer(self, name: str, numel: int, dtype=torch.float32) -> torch.Tensor:
        off = _lib.query("cal_engine_buffer_offset", self._h, name.encode())
        if off < 0:
            raise KeyError(name)
        if dtype == torch.float64:
            return self._ws[off:off + 2 * numel].view(torch.float64)
        if dtype == torch.int32:
            return self._ws[off:off + numel].view(torch.int32)
        if dtype == torch.int64:
            return self._ws[off:off + 2 * numel].view(torch.int64)
        return self._ws[off:off + numel]

    # --------------------------------------------------------------------- run
    def set_perm_rng(self, seed: int, counter: torch.Tensor):
        """Key of the random-intervention permutations the step draws itself (``train_step(draw_perm=True)``): ``seed`` and
        a device int64 counter that the drawing kernel advances (same stream of permutations as ``cal_randperm``)."""
        assert counter.is_cuda and counter.dtype == torch.int64 and counter.numel() == 1
        self._perm_counter = counter
        _lib.call("cal_engine_set_perm_rng", self._h, int(seed) & ((1 << 64) - 1), _p(counter))

    def drawn_perm(self, B: int) -> torch.Tensor:
        """The permutation the latest ``draw_perm`` step drew (device int64 [B], a view into the workspace)."""
        return self.buffer("perm", B, torch.int64)

    def _run(self, batch, perm, mode: int):
        x = batch.x if getattr(batch, "x", None) is not None else batch.feat
        ei, bvec, y = batch.edge_index, batch.batch, batch.y
        if not (x.is_cuda and ei.is_cuda):
            raise _lib.CalError("StepEngine: batch must be on the GPU (no CPU fallback)")
        N, E, B = x.size(0), ei.size(1), int(batch.num_graphs)
        if x.dtype != torch.float32 or x.size(1) != self.F:
            raise ValueError("features must be float32 [N, %d]" % self.F)
        if (mode & 1) and (B == 1 or N == 1):
            # the reference's error behaviour: torch.nn.BatchNorm1d in training mode refuses a single row (bn_feat over one
            # node, model.py:90; the readout BatchNorms over one graph, model.py:127-131) -- same exception, same text
            raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                             % (torch.Size([1, self.F if N == 1 else self.H]),))
        if self.heads:
            self._sync_gat()
        self.reserve(N, E, B)
        self._last_B = B
        # layout facts of a collated batch -> one-kernel per-graph CSR build (the tensors stay referenced by the batch)
        lay = _layout_of(batch, B) if self.fused else dict(ptr=None, edge_ptr=None, no_self_loops=False, max_nodes=0, max_edges=0)
        nptr, eptr = lay["ptr"], lay["edge_ptr"]
        ok = (lay["no_self_loops"] and torch.is_tensor(nptr) and torch.is_tensor(eptr)
              and nptr.is_cuda and eptr.is_cuda and nptr.dtype == torch.long and eptr.dtype == torch.long
              and nptr.numel() == B + 1 and eptr.numel() == B + 1 and nptr.is_contiguous() and eptr.is_contiguous())
        ptrs = (nptr.data_ptr(), eptr.data_ptr()) if ok else (0, 0)
        bounds = (lay["max_nodes"], lay["max_edges"])
        # small-graph packing: a collated batch of small graphs carries tiles of consecutive graphs (data.pack_tiles); the
        # per-graph kernels then take the TILES' offsets and bounds, and the first graph of every tile (cal_engine_set_tiles)
        # (only when one graph per workgroup would need more than one round of the one-per-CU backward kernels: below that
        #  a tile is just a bigger unit of the same launch -- MUTAG-like B = 64 measured 0.253 ms unpacked, 0.275 ms packed)
        want_tiles = bool(self.tiles)
        if self.tiles is True:
            want_tiles = B * (self.H // 64) > self.tile_min_units
        tp = getattr(batch, "tile_ptr", None) if (ok and want_tiles and self.H in (64, 128)) else None
        tiles = (0, 0)
        if torch.is_tensor(tp) and tp.is_cuda and batch.tile_node_ptr.is_cuda and batch.tile_edge_ptr.is_cuda and tp.numel() >= 2:
            tiles = (tp.data_ptr(), int(tp.numel()) - 1)
            ptrs = (batch.tile_node_ptr.data_ptr(), batch.tile_edge_ptr.data_ptr())
            bounds = (int(batch.tile_max_nodes), int(batch.tile_max_edges))
        if tiles != self._tiles:
            _lib.call("cal_engine_set_tiles", self._h, tiles[0] or None, tiles[1])
            self._tiles = tiles
        if ptrs != self._ptrs:
            _lib.call("cal_engine_set_graph_ptrs", self._h, ptrs[0] or None, ptrs[1] or None)
            self._ptrs = ptrs
        if bounds != self._bounds:
            _lib.call("cal_engine_set_graph_bounds", self._h, bounds[0], bounds[1])
            self._bounds = bounds
        if mode & 16:
            perm = None                                   # drawn by the step's first kernel
        elif perm is None:
            perm = self._identity.get(B)
            if perm is None:
                perm = self._identity[B] = torch.arange(B, device=self.device)
        if y is None:
            y = torch.zeros(B, dtype=torch.long, device=self.device)
        _lib.call("cal_engine_step", self._h, _p(x.contiguous()), _p(ei.contiguous()), _p(bvec.contiguous()),
                  _p(y.view(-1).contiguous()), _p(perm.contiguous()) if perm is not None else None, N, E, B,
                  self.wc, self.wo, self.wco, mode, _stream())
        return B

    def forward(self, batch, perm=None, training: bool = False):
        B = self._run(batch, perm, 1 if training else 0)
        lp = self.buffer("logp", 3 * B * self.C).view(3, B, self.C)
        return lp[0], lp[1], lp[2]

    def logp_copy(self):
        """The latest forward's three [B, C] log-prob outputs as views of ONE private copy (the workspace buffer is
        overwritten by the next step; a caller may keep what a forward returned)."""
        B = self._last_B
        lp = self.buffer("logp", 3 * B * self.C).clone().view(3, B, self.C)
        return lp[0], lp[1], lp[2]

    def train_step(self, batch, perm=None, adam: bool = True, tick: bool = False, draw_perm: bool = False):
        """forward + loss + backward (+ Adam); returns the device stats tensor
        [loss, c_loss, o_loss, co_loss, correct_o] (a view into the workspace).  ``tick`` (with ``adam=False``):
        the update follows a gradient exchange as ``adam_ticked()``; the step advances the Adam step counter.
        ``draw_perm``: the step draws the random-intervention permutation itself (``set_perm_rng``), in its first kernel."""
        self._run(batch, perm, 3 | (4 if adam else (8 if tick else 0)) | (16 if draw_perm else 0))
        return self.buffer("stats", 5)

    def adam(self):
        _lib.call("cal_engine_adam", self._h, _stream())

    def adam_ticked(self):
        """The Adam update of a ``train_step(adam=False, tick=True)``, one launch (after the gradient all-reduce)."""
        _lib.call("cal_engine_adam_ticked", self._h, _stream())

    def set_grad_scale(self, scale: float):
        """Factor on the gradient inside Adam: 1 / world_size turns the all-reduced sum into the replicas' mean."""
        _lib.call("cal_engine_set_grad_scale", self._h, float(scale))

    def perm_stage(self) -> "PermStage":
        """The engine's pinned ring for host-drawn intervention permutations (model.py:147-152: Python's RNG on the host)."""
        st = getattr(self, "_perm_stage", None)
        if st is None:
            st = self._perm_stage = PermStage(self.device)
        return st

    def grad_views(self):
        """(parameters, their views into ``flat_g``) in parameter order -- built once: the autograd node hands these out as
        ``p.grad`` after every backward."""
        gv = getattr(self, "_grad_views", None)
        if gv is None:
            from .trainer import flat_offsets
            params = list(self.model.parameters())
            views = [self.flat_g[off:off + p.numel()].view(p.shape) for p, off in zip(params, flat_offsets(params)[0])]
            gv = self._grad_views = (params, views)
        return gv

    def backward_from(self, batch, dlogp: torch.Tensor):
        """Backward of the LAST training-mode ``forward`` of ``batch`` from an external
        gradient w.r.t. its three log-prob outputs (``dlogp`` [3, B, C]); fills ``flat_g``."""
        x = batch.x if getattr(batch, "x", None) is not None else batch.feat
        N, E, B = x.size(0), batch.edge_index.size(1), int(batch.num_graphs)
        dlogp = dlogp.contiguous()
        assert dlogp.shape == (3, B, self.C) and dlogp.dtype == torch.float32 and dlogp.is_cuda
        _lib.call("cal_engine_backward_from", self._h, _p(x.contiguous()), _p(batch.batch.contiguous()), _p(dlogp),
                  N, E, B, _stream())


class PermStage:
    """Pinned ring for the per-step intervention permutation (a pageable H2D copy would block the host on the GPU)."""

    def __init__(self, device, n=1024, slots=8):
        self.device = device
        self.host = [torch.empty(n, dtype=torch.long).pin_memory() for _ in range(slots)]
        self.dev = [torch.empty(n, dtype=torch.long, device=device) for _ in range(slots)]
        self.ev = [None] * slots
        self.i = 0
        self.host_np = None

    def put(self, perm):
        """``perm``: a CPU int64 tensor, or the Python list ``random.shuffle`` produced (written straight into the pinned slot:
        ``torch.tensor(list)`` + a tensor copy were 35 us of every step of the reference-shaped loop)."""
        is_list = isinstance(perm, list)
        n = len(perm) if is_list else perm.numel()
        if n > self.host[0].numel():
            return (torch.tensor(perm) if is_list else perm).to(self.device)
        k, self.i = self.i, (self.i + 1) % len(self.host)
        if self.ev[k] is not None:
            self.ev[k].synchronize()
        if is_list:
            if self.host_np is None:
                self.host_np = [h.numpy() for h in self.host]
            self.host_np[k][:n] = perm
        else:
            self.host[k][:n].copy_(perm)
        d = self.dev[k][:n]
        d.copy_(self.host[k][:n], non_blocking=True)
        self.ev[k] = torch.cuda.Event()
        self.ev[k].record()
        return d


class _EngineAutograd(torch.autograd.Function):
    """The whole CausalGCN forward as ONE autograd node on the native engine (nn.Module surface):
    forward = cal_engine_step(mode 1), backward = cal_engine_backward_from.  The parameter gradients
    land in the engine's flat gradient buffer, of which every ``p.grad`` is (re)made a view."""

    @staticmethod
    def forward(ctx, eng: "StepEngine", batch, perm, anchor):
        eng.forward(batch, perm, training=True)
        ctx.eng, ctx.batch = eng, batch
        eng._fwd_token = getattr(eng, "_fwd_token", 0) + 1
        ctx.token = eng._fwd_token
        return eng.logp_copy()

    @staticmethod
    def backward(ctx, gc, go, gco):
        eng = ctx.eng
        if ctx.token != eng._fwd_token:
            raise RuntimeError("cal_amd engine: another forward ran before this backward "
                               "(the engine keeps one step's activations)")
        B, C = eng._last_B, eng.C
        g = _as_one_block(gc, go, gco, B, C)
        if g is None:
            z = torch.zeros(B, C, dtype=torch.float32, device=eng.device)
            g = torch.stack([t if t is not None else z for t in (gc, go, gco)]).to(torch.float32)
        # p.grad tensors that already alias the flat buffer hold the gradients of an earlier backward (accumulation)
        # or in-place zeros (zero_grad(set_to_none=False)): backward_from overwrites the buffer, so keep them and add
        params, views = eng.grad_views()
        alias = [p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, views)]
        old = eng.flat_g.clone() if any(alias) else None
        eng.backward_from(ctx.batch, g)
        eng._fwd_token += 1                                   # a second backward would double-count
        for p, view, al in zip(params, views, alias):          # hand the gradients to autograd's owners
            if al:
                off = view.storage_offset()
                view.add_(old[off:off + p.numel()].view(p.shape))
            elif p.grad is None:
                p.grad = view
            else:
                p.grad.add_(view)
        return None, None, None, None


def _as_one_block(gc, go, gco, B, C):
    """The three output gradients as ONE contiguous [3, B, C] tensor without a copy when they are the three slices of one
    (what ``fused_causal_loss`` hands back); else ``None``."""
    if gc is None or go is None or gco is None:
        return None
    base = gc._base
    if base is None or go._base is not base or gco._base is not base or base.dtype != torch.float32 or not base.is_contiguous():
        return None
    if base.numel() != 3 * B * C or not (gc.is_contiguous() and go.is_contiguous() and gco.is_contiguous()):
        return None
    p0, n = base.data_ptr(), 4 * B * C
    if (gc.data_ptr(), go.data_ptr(), gco.data_ptr()) != (p0, p0 + n, p0 + 2 * n):
        return None
    return base.view(3, B, C)


_LOSS_FLAGS = {}         # device -> int32[1]: bit 1 once a fused loss saw a label outside [0, C) (sticky until check_loss_labels)


def _loss_flag(device):
    f = _LOSS_FLAGS.get(device)
    if f is None:
        f = _LOSS_FLAGS[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return f


def check_loss_labels():
    """Raise if a fused causal loss since the last call saw a label outside [0, num_classes) -- ``F.nll_loss`` in the reference
    (train_causal.py:178-180) stops with a device assert there; the fused kernel gives such a graph zero loss and flags it.
    Synchronising: called where the loops read their statistics back anyway."""
    for dev, f in list(_LOSS_FLAGS.items()):
        if int(f.item()) != 0:
            f.zero_()
            raise _lib.CalError("cal_amd: a label outside [0, num_classes) reached the fused causal loss (torch's nll_loss would "
                                "have raised; ignore_index is not supported on the fused path)")


class _FusedCausalLoss(torch.autograd.Function):
    """``loss, c_loss, o_loss, co_loss`` of train_causal.py:176-183 from the three log-prob outputs as ONE launch
    (``cal_causal_loss``), with the gradient w.r.t. the log-probs produced in the same launch: a statement-by-statement loop
    (model(data) -> loss -> backward -> step) then spends one kernel and one autograd node on the loss instead of ~20."""

    @staticmethod
    def forward(ctx, c_logs, o_logs, co_logs, y, wc, wo, wco):
        B, C = c_logs.shape                              # (the three are consecutive [B, C] blocks of one buffer: fused_causal_loss)
        out = torch.empty(4, dtype=torch.float32, device=c_logs.device)
        dl = torch.empty(3, B, C, dtype=torch.float32, device=c_logs.device)
        _lib.call("cal_causal_loss", _p(c_logs), _p(y), B, C, float(wc), float(wo), float(wco), _p(out), _p(dl),
                  _p(_loss_flag(c_logs.device)), _stream())
        ctx.dl, ctx.w = dl, (float(wc), float(wo), float(wco))
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def backward(ctx, g_loss, g_c, g_o, g_co):
        dl = ctx.dl                                    # = d loss / d (c, o, co), the weights folded in
        if g_c is None and g_o is None and g_co is None:
            if g_loss is None:
                return None, None, None, None, None, None, None
            g = dl * g_loss                            # one launch; its three slices go to the model's node as one block
            return g[0], g[1], g[2], None, None, None, None
        # the per-term outputs are used differentiably too: d term_k / d head_k = dl[k] / w_k
        gs = []
        for k, (gt, w) in enumerate(zip((g_c, g_o, g_co), ctx.w)):
            if w == 0.0:
                raise RuntimeError("fused_causal_loss: a loss term with weight 0 is differentiated on its own; compute the loss with torch")
            coef = (g_loss * w if g_loss is not None else 0.0) + (gt if gt is not None else 0.0)
            gs.append(dl[k] / w * coef)
        return gs[0], gs[1], gs[2], None, None, None, None


def fused_causal_loss(c_logs, o_logs, co_logs, y, num_classes, wc, wo, wco):
    """The fused form of the reference's loss when ``(c, o, co)`` are the three [B, C] slices of one contiguous float32 CUDA
    tensor (what an engine-backed model returns); ``None`` otherwise (the caller then uses torch)."""
    if not (torch.is_tensor(c_logs) and c_logs.is_cuda and c_logs.dtype == torch.float32 and c_logs.dim() == 2):
        return None
    B, C = c_logs.shape
    if C != num_classes or y.numel() != B or not y.is_cuda or y.dtype != torch.long:
        return None
    base = c_logs._base
    if base is None or o_logs._base is not base or co_logs._base is not base or not base.is_contiguous() or base.numel() != 3 * B * C:
        return None
    p0, n = base.data_ptr(), 4 * B * C
    if (c_logs.data_ptr(), o_logs.data_ptr(), co_logs.data_ptr()) != (p0, p0 + n, p0 + 2 * n):
        return None
    return _FusedCausalLoss.apply(c_logs, o_logs, co_logs, y.view(-1).contiguous(), wc, wo, wco)


def engine_forward_autograd(eng: "StepEngine", batch, perm):
    """(c, o, co) log-probs with autograd support through the native engine."""
    anchor = next(eng.model.parameters())       # any leaf that requires grad: makes the node differentiable
    return _EngineAutograd.apply(eng, batch, perm, anchor)
