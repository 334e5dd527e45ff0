"""CausalTrainer: the train step of train_causal.py:173-192 as a replayable
HIP graph, plus the data-parallel gradient exchange.

MI355X-first choices (DESIGN.md section "step engine"):

* all parameters are views of ONE flat fp32 buffer and all gradients views of
  ONE flat gradient buffer -> Adam is a single fused update over one tensor and
  the data-parallel exchange is a single RCCL all-reduce of ~0.5 MB
  (latency-bound on xGMI, so: one bucket, one call);
* forward + 3-term loss + backward for one resident, pre-collated batch is
  captured once into a hipGraph (the kernels are enqueued through the C ABI on
  the capturing stream) and replayed -- no per-op host launch cost;
* the GraphPlan (CSR build) is rebuilt inside every step: it is part of the
  work the reference does per step (GCNConv.norm, gcn_conv.py:79-89);
* the random-intervention permutation (model.py:147-152) stays a host-side
  ``random.shuffle`` and is uploaded into a static device buffer before replay.
"""
from __future__ import annotations

import random
from typing import Dict, Optional

import torch
import torch.distributed as dist

from .train_causal import causal_loss


def flatten_parameters(model: torch.nn.Module):
    """Re-home every parameter (and its .grad) into one contiguous buffer."""
    params = [p for p in model.parameters()]
    total = sum(p.numel() for p in params)
    dev = params[0].device
    flat_p = torch.empty(total, dtype=torch.float32, device=dev)
    flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
    off = 0
    for p in params:
        n = p.numel()
        flat_p[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat_p[off:off + n].view(p.shape)
        p.grad = flat_g[off:off + n].view(p.shape)
        off += n
    return flat_p, flat_g


class _Captured:
    __slots__ = ("graph", "perm", "stats")


class CausalTrainer:
    def __init__(self, model, args, lr: float = 1e-3, weight_decay: float = 0.0,
                 use_graph: bool = True, world_size: int = 1, rebuild_plan: bool = True):
        self.model, self.args = model, args
        self.use_graph = use_graph
        self.world_size = world_size
        self.rebuild_plan = rebuild_plan
        self.flat_p, self.flat_g = flatten_parameters(model)
        self.flat_p.grad = self.flat_g
        self.lr = torch.tensor(float(lr), device=self.flat_p.device) if use_graph else float(lr)
        self.opt = torch.optim.Adam([self.flat_p], lr=self.lr, weight_decay=weight_decay,
                                    capturable=use_graph)
        self._graphs: Dict[int, _Captured] = {}
        self._pool = torch.cuda.graph_pool_handle() if use_graph else None
        self._opt_graph: Optional[torch.cuda.CUDAGraph] = None
        self.stats = torch.zeros(5, dtype=torch.float32, device=self.flat_p.device)
        self.model.train()

    # ------------------------------------------------------------------ pieces
    def set_lr(self, lr: float):
        if torch.is_tensor(self.lr):
            self.lr.fill_(lr)
        else:
            for g in self.opt.param_groups:
                g["lr"] = lr

    def draw_perm(self, num: int) -> torch.Tensor:
        """model.py:147-152 on the host (Python RNG, like the reference)."""
        l = list(range(num))
        if self.args.with_random and (getattr(self.model, "with_random", True)
                                      or not self.model._gate_on_with_random):
            random.shuffle(l)
        return torch.tensor(l, dtype=torch.long)

    def _fwd_bwd(self, batch, perm, stats):
        self.flat_g.zero_()
        if self.rebuild_plan:
            batch._plan = None
        c, o, co = self.model(batch, eval_random=self.args.with_random, perm=perm)
        loss, lc, lo, lco = causal_loss(c, o, co, batch.y, self.model.num_classes, self.args)
        loss.backward()
        with torch.no_grad():
            correct = o.max(1)[1].eq(batch.y.view(-1)).sum().to(torch.float32)
            stats.copy_(torch.stack([loss.detach(), lc.detach(), lo.detach(), lco.detach(), correct]))

    def _allreduce(self):
        if self.world_size > 1:
            dist.all_reduce(self.flat_g)
            self.flat_g.mul_(1.0 / self.world_size)

    def _capture(self, batch) -> _Captured:
        cap = _Captured()
        nb = batch.num_graphs
        cap.perm = torch.arange(nb, dtype=torch.long, device=self.flat_p.device)
        cap.stats = torch.zeros(5, dtype=torch.float32, device=self.flat_p.device)
        # warm-up on a side stream (allocator + autograd state), restoring BN statistics after
        bn_state = {k: v.clone() for k, v in self.model.state_dict().items()
                    if "running_" in k or "num_batches" in k}
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._fwd_bwd(batch, cap.perm, cap.stats)
        torch.cuda.current_stream().wait_stream(s)
        self.model.load_state_dict(bn_state, strict=False)
        cap.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cap.graph, pool=self._pool):
            self._fwd_bwd(batch, cap.perm, cap.stats)
        self.model.load_state_dict(bn_state, strict=False)
        return cap

    def _reset_opt_state(self):
        st = self.opt.state[self.flat_p]
        st["step"].zero_()
        st["exp_avg"].zero_()
        st["exp_avg_sq"].zero_()

    def _build_opt_graph(self):
        """Capture Adam's update once.  Its lazy state init must happen outside
        capture, so one throw-away step runs first and everything it touched is
        restored."""
        saved_p, saved_g = self.flat_p.clone(), self.flat_g.clone()
        fresh = self.flat_p not in self.opt.state or len(self.opt.state[self.flat_p]) == 0
        saved_state = None
        if not fresh:
            saved_state = {k: v.clone() for k, v in self.opt.state[self.flat_p].items()}
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.opt.step()
        torch.cuda.current_stream().wait_stream(s)
        self._opt_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._opt_graph):
            self.opt.step()
        if saved_state is None:
            self._reset_opt_state()
        else:
            for k, v in saved_state.items():
                self.opt.state[self.flat_p][k].copy_(v)
        self.flat_p.copy_(saved_p)
        self.flat_g.copy_(saved_g)

    def _opt_step(self):
        if not self.use_graph:
            self.opt.step()
            return
        if self._opt_graph is None:
            self._build_opt_graph()
        self._opt_graph.replay()

    # -------------------------------------------------------------------- step
    def prepare(self, batch):
        """Capture the graphs for a resident batch ahead of the timed region."""
        if self.use_graph:
            if id(batch) not in self._graphs:
                self._graphs[id(batch)] = self._capture(batch)
            if self._opt_graph is None:
                self._build_opt_graph()

    def step(self, batch, perm: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One train step on a device-resident batch; returns the device stats
        tensor [loss, c_loss, o_loss, co_loss, correct_o] (no host sync)."""
        if perm is None:
            perm = self.draw_perm(batch.num_graphs)
        if self.use_graph:
            cap = self._graphs.get(id(batch))
            if cap is None:
                self.prepare(batch)
                cap = self._graphs[id(batch)]
            cap.perm.copy_(perm, non_blocking=True)
            cap.graph.replay()
            stats = cap.stats
        else:
            self._fwd_bwd(batch, perm.to(self.flat_p.device, non_blocking=True), self.stats)
            stats = self.stats
        self._allreduce()
        self._opt_step()
        return stats
