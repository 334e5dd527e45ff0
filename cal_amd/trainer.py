"""CausalTrainer: the train step of train_causal.py:173-192 as a replayable
HIP graph, plus the data-parallel gradient exchange.

MI355X-first choices (DESIGN.md section "step engine"):

* all parameters are views of ONE flat fp32 buffer and all gradients views of
  ONE flat gradient buffer -> Adam is a single fused update over one tensor and
  the data-parallel exchange is a single RCCL all-reduce of ~0.5 MB
  (latency-bound on xGMI, so: one bucket, one call);
* CausalGCN / CausalGAT / CausalGIN run on the native step engine (cal_amd/csrc/engine.hip):
  forward + 3-term loss + backward + Adam are a few dozen fused kernels enqueued
  by ONE C call; other models run the operator-level autograd path
  (cal_amd.ops) with torch's loss / Adam;
* either way the step for one resident, pre-collated batch is captured once
  into a hipGraph and replayed -- no per-kernel host launch cost; a pass over
  several resident batches can share one graph launch (``step_sequence``);
* with more than one rank the RCCL all-reduce is captured inside the same graph
  (between the backward's last kernel and Adam, which applies the 1/world mean),
  so N > 1 differs from N = 1 only by the collective node;
* the GraphPlan (CSR build) is rebuilt inside every step: it is part of the
  work the reference does per step (GCNConv.norm, gcn_conv.py:79-89);
* the random-intervention permutation (model.py:147-152) is drawn on the device
  inside the captured step (``cal_randperm``, keyed by a seed taken from Python's
  RNG); ``step(perm=...)`` / ``device_perm=False`` keep the host-side
  ``random.shuffle`` stream of the reference.
"""
from __future__ import annotations

import os
import random
import warnings
from typing import Dict, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib
from .plan import _p, _stream
from .train_causal import causal_loss


#: floats: every parameter starts on a 16-byte boundary of the flat buffers (the 128 x 128 GEMM and the row kernels read weights
#: with 16 B loads; packed back to back, a 10-element bn_feat.weight put everything behind it on an 8-byte boundary and an odd
#: feature count on a 4-byte one, which the big-batch kernels refuse)
FLAT_ALIGN = 4


def flat_offsets(params):
    """Offsets (in elements) of `params` inside the flat parameter / gradient / moment buffers, and the buffers' length."""
    offs, off = [], 0
    for p in params:
        off = (off + FLAT_ALIGN - 1) // FLAT_ALIGN * FLAT_ALIGN
        offs.append(off)
        off += p.numel()
    return offs, (off + FLAT_ALIGN - 1) // FLAT_ALIGN * FLAT_ALIGN


def flatten_parameters(model: torch.nn.Module):
    """Re-home every parameter (and its .grad) into one buffer (`flat_offsets`; the padding stays zero: zero gradient, zero
    moments, so Adam leaves it alone)."""
    params = [p for p in model.parameters()]
    offs, total = flat_offsets(params)
    dev = params[0].device
    flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
    flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
    for p, off in zip(params, offs):
        n = p.numel()
        flat_p[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat_p[off:off + n].view(p.shape)
        p.grad = flat_g[off:off + n].view(p.shape)
    return flat_p, flat_g


def _batch_ptrs(batch):
    x = batch.x if getattr(batch, "x", None) is not None else batch.feat
    return (x.data_ptr(), batch.edge_index.data_ptr(), batch.batch.data_ptr(), batch.y.data_ptr(),
            int(x.size(0)), int(batch.edge_index.size(1)), int(batch.num_graphs))


#: captured single-step graphs kept per trainer; batches beyond it (a loader yielding fresh batches every step)
#: evict the oldest entry instead of growing without bound
MAX_CAPTURED = 256


class _Captured:
    """Captured step(s) of one resident batch: `graphs[True]` draws the intervention permutation on the
    device inside the graph (cal_randperm), `graphs[False]` reads it from `perm` (uploaded by the host)."""
    __slots__ = ("graphs", "perm", "stats", "batch", "ptrs", "ws_gen")


class _PinnedRing:
    """Pinned staging buffers for the per-step permutation upload: a pageable-memory H2D copy is
    synchronous and would serialise the host with the previous step's GPU work."""

    def __init__(self, n: int, slots: int = 8):
        self.bufs = [torch.empty(n, dtype=torch.long).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.i = 0

    def stage(self, perm: torch.Tensor, dst: torch.Tensor):
        k = self.i
        self.i = (self.i + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
        buf = self.bufs[k][:perm.numel()]
        buf.copy_(perm)
        dst.copy_(buf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev


class CausalTrainer:
    def __init__(self, model, args, lr: float = 1e-3, weight_decay: float = 0.0,
                 use_graph: bool = True, world_size: int = 1, rebuild_plan: bool = True,
                 use_engine: Optional[bool] = None, device_perm: bool = True,
                 force_exchange: bool = False, graph_exchange: Optional[bool] = None,
                 p2p_exchange: Optional[bool] = None, deterministic: Optional[bool] = None):
        """``deterministic``: fixed-order BatchNorm sums -- bit-reproducible steps (``StepEngine(deterministic=...)``)."""
        from . import engine as eng_mod
        self.model, self.args = model, args
        self.use_graph = use_graph
        self.world_size = world_size
        self.rebuild_plan = rebuild_plan
        self.flat_p, self.flat_g = flatten_parameters(model)
        self.flat_p.grad = self.flat_g
        if use_engine is None:
            use_engine = eng_mod.supported(model)
        self.engine = None
        model.use_engine = False        # the trainer drives the engine itself; the module path stays op-level
        if use_engine:
            self.engine = eng_mod.StepEngine(model, lr=lr, weight_decay=weight_decay,
                                             flat=(self.flat_p, self.flat_g), deterministic=deterministic)
            self.engine.wc, self.engine.wo, self.engine.wco = float(args.c), float(args.o), float(args.co)
            self.lr = self.engine.lr
            self.opt = None
        else:
            self.lr = torch.tensor(float(lr), device=self.flat_p.device) if use_graph else float(lr)
            self.opt = torch.optim.Adam([self.flat_p], lr=self.lr, weight_decay=weight_decay,
                                        capturable=use_graph)
        self._graphs: Dict[int, _Captured] = {}
        self._seqs: Dict[tuple, tuple] = {}
        self._pool = torch.cuda.graph_pool_handle() if use_graph else None
        self._opt_graph: Optional[torch.cuda.CUDAGraph] = None
        self.stats = torch.zeros(5, dtype=torch.float32, device=self.flat_p.device)
        self._ring: Optional[_PinnedRing] = None
        self._eager_perm: Optional[torch.Tensor] = None
        # model.py:147-152 draws the permutation with Python's RNG on the host; by default it is drawn on the
        # device instead (keyed by a seed taken from Python's RNG here, so random.seed() still controls it):
        # no host RNG, pinned staging or H2D copy in front of each step's hipGraph.  step(perm=...) overrides.
        self.device_perm = bool(device_perm)
        self._perm_seed = random.getrandbits(63)
        self._perm_counter = torch.zeros(1, dtype=torch.int64, device=self.flat_p.device)
        if self.engine is not None:       # the engine draws inside its first kernel: no launch of its own for the permutation
            self.engine.set_perm_rng(self._perm_seed, self._perm_counter)
        # Data parallel: one all-reduce (sum) of the flat gradient bucket per step, the 1/world mean
        # folded into the engine's Adam kernel.  `exchange_in_graph`: the collective is captured INSIDE the step's
        # hipGraph (RCCL collectives are capturable; gloo is not), so N > 1 keeps one graph launch per step -- and
        # the multi-step sequence graph -- and differs from N = 1 only by the collective node.  Otherwise the step is
        # graph(forward + backward) -> eager all-reduce -> graph(Adam).  `force_exchange` runs the exchange on a
        # one-rank group too (tests the captured-collective path on a single GPU).
        self.exchange = bool((world_size > 1 or force_exchange) and dist.is_available() and dist.is_initialized())
        if world_size > 1 and not self.exchange:
            raise RuntimeError("CausalTrainer(world_size > 1) needs an initialised torch.distributed process group")
        if graph_exchange is None:
            graph_exchange = os.environ.get("CAL_AMD_GRAPH_EXCHANGE", "1") != "0"
        self.exchange_in_graph = bool(self.exchange and self.engine is not None and use_graph and graph_exchange
                                      and dist.get_backend() == "nccl")
        if self.exchange and self.engine is not None:
            self.engine.set_grad_scale(1.0 / max(1, dist.get_world_size()))
        # one-shot peer-memory exchange instead of the collective (cal_amd/p2p.py): a plain kernel node behind the backward
        if p2p_exchange is None:
            p2p_exchange = os.environ.get("CAL_AMD_P2P_EXCHANGE", "0") == "1"
        self.p2p = None
        if p2p_exchange and self.exchange and self.engine is not None:
            from .p2p import P2PExchange
            self.p2p = P2PExchange(self.engine)
            self.exchange_in_graph = bool(use_graph)
        # the optimizer update rides in the same graph as forward/backward (always on one GPU)
        self.fused_opt = self.engine is not None and (not self.exchange or self.exchange_in_graph or not use_graph or self.p2p is not None)
        self.model.train()

    # ------------------------------------------------------------------ pieces
    def check_status(self):
        """Synchronising check that no step since the last call flagged its batch as invalid (StepEngine.check_status)."""
        if self.p2p is not None:
            self.p2p.check()
        if self.engine is not None:
            self.engine.check_status()

    def set_lr(self, lr: float):
        if torch.is_tensor(self.lr):
            self.lr.fill_(lr)
        else:
            for g in self.opt.param_groups:
                g["lr"] = lr

    def _shuffles(self) -> bool:
        return bool(self.args.with_random and (getattr(self.model, "with_random", True)
                                               or not self.model._gate_on_with_random))

    def draw_perm(self, num: int) -> torch.Tensor:
        """model.py:147-152 on the host (Python RNG, like the reference)."""
        l = list(range(num))
        if self._shuffles():
            random.shuffle(l)
        return torch.tensor(l, dtype=torch.long)

    def _device_perm_into(self, dst: torch.Tensor, num: int):
        """model.py:147-152 on the device: dst[:num] <- fresh random permutation (identity when the model
        does not shuffle); enqueued on the current stream, capturable."""
        if self._shuffles():
            _lib.call("cal_randperm", _p(dst), num, self._perm_seed, _p(self._perm_counter), _stream())
        else:
            dst[:num].copy_(torch.arange(num, dtype=torch.long, device=dst.device))

    def _use_device_perm(self, num: int) -> bool:
        return self.device_perm and num <= 4096

    def reserve_for(self, batches: Sequence):
        """Size the engine workspace for the largest of `batches` BEFORE any graph is captured."""
        if self.engine is not None:
            n = max(int(b.batch.numel()) for b in batches)
            e = max(int(b.edge_index.size(1)) for b in batches)
            g = max(int(b.num_graphs) for b in batches)
            self.engine.reserve(n, e, g)

    def _in_engine_draw(self, num: int) -> bool:
        """The engine draws the permutation in its own first kernel (up to 1024 graphs; beyond that cal_randperm)."""
        return self.engine is not None and num <= 1024 and self._shuffles()

    def _fwd_bwd(self, batch, perm, stats, draw: bool = False):
        """One forward + backward (+ fused Adam); returns the device stats tensor.  On the engine path that is a
        view of the engine's own stats buffer (no copy node in the graph): it holds the LATEST step's values.
        ``draw``: the intervention permutation is drawn on the device for this step (inside the engine's first kernel
        when it can, else by cal_randperm into ``perm`` first)."""
        if draw and not self._in_engine_draw(batch.num_graphs):
            self._device_perm_into(perm, batch.num_graphs)
            draw = False
        if self.engine is not None:
            if not self.exchange:
                return self.engine.train_step(batch, perm, adam=True, draw_perm=draw)
            stats = self.engine.train_step(batch, perm, adam=False, tick=True, draw_perm=draw)  # k_finish advances the Adam step
            if self.p2p is not None:                  # publish / wait / sum / Adam: one kernel
                self.p2p.adam()
            elif self.fused_opt:                      # collective + update in stream order (captured with the step)
                dist.all_reduce(self.flat_g)
                self.engine.adam_ticked()
            return stats
        self.flat_g.zero_()
        if self.rebuild_plan:
            batch._plan = None
        c, o, co = self.model(batch, eval_random=self.args.with_random, perm=perm)
        loss, lc, lo, lco = causal_loss(c, o, co, batch.y, self.model.num_classes, self.args)
        loss.backward()
        with torch.no_grad():
            correct = o.max(1)[1].eq(batch.y.view(-1)).sum().to(torch.float32)
            stats.copy_(torch.stack([loss.detach(), lc.detach(), lo.detach(), lco.detach(), correct]))
        return stats

    def _allreduce(self):
        """Gradient exchange outside the step graph (operator-level path, gloo, CAL_AMD_GRAPH_EXCHANGE=0)."""
        if self.exchange and not self.fused_opt:
            dist.all_reduce(self.flat_g)
            if self.engine is None:                   # torch Adam has no gradient factor; the engine's k_adam does
                self.flat_g.mul_(1.0 / dist.get_world_size())

    def _snapshot(self):
        state = {k: v.clone() for k, v in self.model.state_dict().items()}
        extra = None
        if self.engine is not None:
            extra = (self.engine.exp_avg.clone(), self.engine.exp_avg_sq.clone(), self.engine.step_count.clone())
        return state, extra

    def _restore(self, snap):
        state, extra = snap
        with torch.no_grad():
            for k, v in self.model.state_dict().items():
                v.copy_(state[k])
        if extra is not None:
            self.engine.exp_avg.copy_(extra[0])
            self.engine.exp_avg_sq.copy_(extra[1])
            self.engine.step_count.copy_(extra[2])

    def _capture(self, batch, dev_perm: bool, cap: Optional[_Captured] = None) -> _Captured:
        nb = batch.num_graphs
        if cap is None:
            cap = _Captured()
            cap.graphs = {}
            # the graph bakes in the batch's device pointers: keep the batch alive (its id() is the cache key and
            # must not be recycled by a later batch) and remember the pointers to detect in-place replacement
            cap.batch = batch
            cap.ptrs = _batch_ptrs(batch)
            cap.perm = torch.arange(nb, dtype=torch.long, device=self.flat_p.device)
            cap.stats = torch.zeros(5, dtype=torch.float32, device=self.flat_p.device)
        draws = dev_perm and self._shuffles()
        if self.engine is not None:
            x = batch.x if getattr(batch, "x", None) is not None else batch.feat
            cn, ce, cb = self.engine._cap
            if x.size(0) > cn or batch.edge_index.size(1) > ce or nb > cb:
                # growth bumps the engine's workspace generation: graphs captured on the old workspace are evicted and
                # re-captured when their batch comes up again (_captured); reserve_for(all batches) up front avoids that
                self.engine.reserve(x.size(0), batch.edge_index.size(1), nb)
        # warm-up on a side stream (allocator / autograd state); everything it touched is restored
        snap = self._snapshot()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._fwd_bwd(batch, cap.perm, cap.stats, draw=draws)
        torch.cuda.current_stream().wait_stream(s)
        self._restore(snap)
        g = torch.cuda.CUDAGraph()
        failed = None
        try:
            with torch.cuda.graph(g, pool=self._pool):
                cap.stats = self._fwd_bwd(batch, cap.perm, cap.stats, draw=draws)
        except Exception as exc:
            if not self.exchange_in_graph:
                raise
            failed = exc
        if self.exchange_in_graph and self.p2p is None and dist.is_initialized() and dist.get_world_size() > 1 \
                and not getattr(self, "_capture_agreed", False):
            # The ranks must agree on the outcome: a rank that captured the collective would wait inside its graph for a rank
            # that fell back to the eager one -- a hang, not an error.  One eager all-reduce (MAX) of the local verdict at the
            # FIRST capture (every rank's first step; later captures happen at rank-specific times and must not communicate):
            # every rank reaches it (capture itself communicates nothing), and if ANY rank failed, ALL take the split form below.
            self._capture_agreed = True
            torch.cuda.synchronize()
            verdict = torch.tensor([1.0 if failed is not None else 0.0], device=self.flat_p.device)
            dist.all_reduce(verdict, op=dist.ReduceOp.MAX)
            if verdict.item() > 0 and failed is None:
                failed = RuntimeError("another rank could not capture the gradient all-reduce")
        if failed is not None and getattr(self, "_capture_agreed", False) and not getattr(self, "_first_capture_open", True):
            raise failed              # a later capture failed after the ranks had agreed that it works: an error, never a one-sided fallback
        self._first_capture_open = False
        if failed is not None:
            exc = failed
            # the collective refused stream capture (here or on another rank): keep it between two graphs instead
            warnings.warn("cal_amd: gradient all-reduce could not be captured into the step graph (%r); "
                          "running it between the forward/backward graph and the Adam graph" % (exc,))
            self.exchange_in_graph = False
            self.fused_opt = False
            torch.cuda.synchronize()
            self._restore(snap)
            # graphs captured earlier contain the collective + Adam: drop them, they are re-captured in the split form
            self._graphs.clear()
            self._seqs.clear()
            cap.graphs = {}
            return self._capture(batch, dev_perm, cap)
        self._restore(snap)
        cap.graphs[dev_perm] = g
        cap.ws_gen = self.engine.ws_generation if self.engine is not None else 0
        return cap

    def _reset_opt_state(self):
        st = self.opt.state[self.flat_p]
        st["step"].zero_()
        st["exp_avg"].zero_()
        st["exp_avg_sq"].zero_()

    def _build_opt_graph(self):
        """Capture the optimizer update once.  torch Adam's lazy state init must happen outside
        capture, so one throw-away step runs first and everything it touched is restored."""
        saved_g = self.flat_g.clone()
        snap = self._snapshot()
        saved_state = None
        if self.opt is not None:
            fresh = self.flat_p not in self.opt.state or len(self.opt.state[self.flat_p]) == 0
            saved_state = None if fresh else {k: v.clone() for k, v in self.opt.state[self.flat_p].items()}
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.opt.step()
            torch.cuda.current_stream().wait_stream(s)
        self._opt_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._opt_graph, pool=self._pool):
            if self.opt is not None:
                self.opt.step()
            else:
                self.engine.adam_ticked()            # the step graph (mode bit 8) has advanced the counter
        if self.opt is not None:
            if saved_state is None:
                self._reset_opt_state()
            else:
                for k, v in saved_state.items():
                    self.opt.state[self.flat_p][k].copy_(v)
        self._restore(snap)
        self.flat_g.copy_(saved_g)

    def _opt_step(self):
        if self.fused_opt:
            return
        if not self.use_graph:
            if self.opt is not None:
                self.opt.step()
            else:
                self.engine.adam_ticked()
            return
        if self._opt_graph is None:
            self._build_opt_graph()
        self._opt_graph.replay()

    # -------------------------------------------------------------------- step
    def prepare(self, batch):
        """Capture the graphs for a resident batch ahead of the timed region."""
        if self.use_graph:
            self._captured(batch, self._use_device_perm(batch.num_graphs))
            if self._opt_graph is None and not self.fused_opt:
                self._build_opt_graph()

    # A hipGraph launch costs ~10 us of idle GPU between two replays of a ~0.3 ms step.  When the coming steps are
    # known (an epoch over resident batches) they can share ONE graph: step_sequence(batches) runs len(batches)
    # consecutive train steps -- permutation draw, forward, backward, Adam each -- per launch.  Single-GPU engine
    # path only (with more GPUs the gradient all-reduce sits between backward and Adam, outside the graphs).
    def can_sequence(self) -> bool:
        return self.use_graph and self.fused_opt and self.engine is not None

    def step_sequence(self, batches) -> torch.Tensor:
        """len(batches) consecutive train steps (in this order) as one graph launch; returns the device stats tensor
        of the LAST step.  The captured sequence is cached per tuple of batch objects."""
        if self.p2p is not None:
            self.p2p.check()
        if not self.can_sequence() or not all(self._use_device_perm(b.num_graphs) for b in batches):
            stats = None
            for b in batches:
                stats = self.step(b)
            return stats
        if len(batches) > MAX_CAPTURED // 2:      # a sequence must not evict its own single-step captures: run it in chunks
            stats = None
            for s0 in range(0, len(batches), MAX_CAPTURED // 2):
                stats = self.step_sequence(batches[s0:s0 + MAX_CAPTURED // 2])
            return stats
        key = tuple(id(b) for b in batches)
        seq = self._seqs.get(key)
        if seq is not None and len(seq) > 3 and seq[3] != self.engine.ws_generation:
            del self._seqs[key]
            seq = None
        if seq is None:
            # per-batch state (perm buffers, warm-up) comes from the single-step capture; the captures are taken from
            # _captured()'s return value and pinned for the loop (an eviction while preparing a later batch of the same
            # sequence used to leave `self._graphs[id(b)]` dangling)
            self._pinned = set(key)
            try:
                caps = [self._captured(b, self._use_device_perm(b.num_graphs)) for b in batches]
            finally:
                self._pinned = set()
            snap = self._snapshot()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._pool):
                for b, cap in zip(batches, caps):
                    stats = self._fwd_bwd(b, cap.perm, cap.stats, draw=self._shuffles())
            self._restore(snap)
            seq = self._seqs[key] = (g, stats, list(batches), self.engine.ws_generation)
        seq[0].replay()
        return seq[1]

    def _captured(self, batch, dev_perm: bool) -> _Captured:
        cap = self._graphs.get(id(batch))
        gen = self.engine.ws_generation if self.engine is not None else 0
        if cap is not None and getattr(cap, "ws_gen", gen) != gen:
            # the engine's workspace was re-allocated after this capture: its kernels have the old buffers baked in
            self._evict(id(batch))
            cap = None
        if cap is not None and (cap.batch is not batch or cap.ptrs != _batch_ptrs(batch)):
            # same id() but another object, or the batch's tensors were replaced: the baked-in pointers are stale
            self._evict(id(batch))
            cap = None
        if cap is None or dev_perm not in cap.graphs:
            if cap is None and len(self._graphs) >= MAX_CAPTURED:
                pinned = getattr(self, "_pinned", ())
                victim = next((k for k in self._graphs if k not in pinned), None)
                if victim is not None:
                    self._evict(victim)
            cap = self._capture(batch, dev_perm, cap)
            self._graphs[id(batch)] = cap
        return cap

    def _evict(self, key):
        self._graphs.pop(key, None)
        for k in [k for k in self._seqs if key in k]:
            del self._seqs[k]

    def _upload_perm(self, perm: torch.Tensor, dst: torch.Tensor):
        if perm.is_cuda:
            dst.copy_(perm)
            return
        if self._ring is None or self._ring.bufs[0].numel() < perm.numel():
            self._ring = _PinnedRing(max(perm.numel(), 1024))
        self._ring.stage(perm, dst)

    def step(self, batch, perm: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One train step on a device-resident batch; returns the device stats
        tensor [loss, c_loss, o_loss, co_loss, correct_o] (no host sync)."""
        if self.p2p is not None:
            self.p2p.check()          # an exchange that timed out left the parameters untouched: say so before stepping on (host-mapped word, no sync)
        if self.engine is not None and self.engine.peek_status():
            self.engine.check_status()        # an earlier step flagged its batch (host-mapped mirror, no sync): raise with the message
        nb = batch.num_graphs
        dev = perm is None and self._use_device_perm(nb)
        if perm is None and not dev:
            perm = self.draw_perm(nb)
        if self.use_graph:
            cap = self._captured(batch, dev)
            if self._opt_graph is None and not self.fused_opt:
                self._build_opt_graph()
            if not dev:
                self._upload_perm(perm, cap.perm)
            cap.graphs[dev].replay()
            stats = cap.stats
        else:
            if not dev and perm.is_cuda:
                dperm = perm
            else:
                if self._eager_perm is None or self._eager_perm.numel() < nb:
                    self._eager_perm = torch.empty(max(nb, 1), dtype=torch.long, device=self.flat_p.device)
                dperm = self._eager_perm[:nb]
                if not dev:
                    self._upload_perm(perm, dperm)
                elif not self._shuffles():
                    self._device_perm_into(dperm, nb)           # identity
            stats = self._fwd_bwd(batch, dperm, self.stats, draw=dev and self._shuffles())
        self._allreduce()
        self._opt_step()
        return stats
