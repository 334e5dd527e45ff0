"""The reference's optimizer object on the step engine.

The reference trains with ``torch.optim.Adam(model.parameters(), lr=..., weight_decay=...)`` and steps it once per
mini-batch (train_causal.py:21,76,192); ``CosineAnnealingLR`` rewrites ``param_groups[0]["lr"]`` once per epoch
(train_causal.py:22,29).  The step engine owns a fused Adam over the model's flat parameter buffer (inside the last
kernel of ``cal_engine_step``, or ``cal_engine_adam`` as one launch).  This module connects the two without changing
what a caller sees:

* ``bind(optimizer, model)`` -- when ``optimizer`` is a plain single-group Adam over exactly the parameters of an
  engine-backed model, its per-parameter state (``exp_avg``, ``exp_avg_sq``) is re-homed into views of the engine's flat
  moment buffers and its ``step`` counters are kept in sync, so the engine's update and ``optimizer.step()`` act on the same
  memory and either can run next.  Returns the binding (or ``None``: the caller keeps the generic path).
* ``EngineAdam`` -- ``torch.optim.Adam`` subclass whose ``step()`` is ONE engine launch when the gradients sit in the engine's
  flat gradient buffer (after ``loss.backward()`` through the engine's autograd node); otherwise it is ``Adam.step()``.
  LR schedulers, ``param_groups`` and ``state_dict()`` work as with the parent class.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

_ENGINES = weakref.WeakSet()


def register_engine(engine) -> None:
    """Called by StepEngine.__init__: lets an optimizer find the engine that owns its parameters."""
    _ENGINES.add(engine)


def _engine_of(params):
    p0 = params[0]
    if not p0.is_cuda:
        return None
    a = p0.data_ptr()
    for eng in list(_ENGINES):
        lo = eng.flat_p.data_ptr()
        if lo <= a < lo + 4 * eng.flat_p.numel():
            return eng
    return None


def _plain_adam(opt) -> bool:
    if type(opt) not in (torch.optim.Adam, EngineAdam) or len(opt.param_groups) != 1:
        return False
    g = opt.param_groups[0]
    return not (g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable")
                or g.get("fused") or g.get("decoupled_weight_decay") or torch.is_tensor(g["lr"]))


class Binding:
    """An optimizer attached to a StepEngine (see ``bind``)."""

    def __init__(self, opt, engine, params):
        self.opt, self.engine, self.params = opt, engine, params
        self._hp = None
        self._lr = None
        self.pending = 0                    # engine-side Adam steps not yet written to the optimizer's `step` counters
        self.engine_step = 0.0              # what the engine's device step counter holds
        self.grads_mixed = False            # a step has seen gradients outside the engine's flat buffer
        from .trainer import flat_offsets
        self._views = [(off, p.numel()) for p, off in zip(params, flat_offsets(params)[0])]
        hooks = getattr(engine, "_on_flag", None)
        if hooks is None:
            hooks = engine._on_flag = []
        hooks[:] = [self.after_flag]            # one binding per engine at a time

    def after_flag(self):
        """``check_status`` found flagged steps: the device applied none of them and did not count them (k_finish), while
        ``stepped()`` did -- bring the host-side counters back to what the device holds (synchronising; the raise follows)."""
        s = float(self.engine.step_count.item())
        drift = self.engine_step - s
        if drift > 0:
            self.pending = max(0, self.pending - int(round(drift)))
            self.engine_step = s

    def sync_hparams(self):
        """betas / eps / weight_decay / lr of the optimizer's group -> the engine (lr is a device float: an async fill)."""
        g = self.opt.param_groups[0]
        hp = (float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))
        if hp != self._hp:
            self.engine.set_adam(*hp)
            self._hp = hp
        lr = float(g["lr"])
        if lr != self._lr:
            self.engine.lr.fill_(lr)
            self._lr = lr

    def stepped(self, k: int = 1):
        self.pending += k
        self.engine_step += k
        self.opt._opt_called = True         # what torch's LR schedulers look at to warn about "scheduler before optimizer"

    def rehome(self, full: bool = False):
        """``optimizer.load_state_dict()`` after the bind replaces ``exp_avg`` / ``exp_avg_sq`` by fresh tensors: the engine
        would keep stepping on its old moments and a mid-run restore would be silently ignored.  Whenever the optimizer's
        state no longer aliases the engine's flat moment buffers, copy it in and re-home it (cheap test on the first
        parameter -- a restore replaces every entry at once; ``full`` tests all)."""
        eng = self.engine
        base_m, base_v = eng.exp_avg.data_ptr(), eng.exp_avg_sq.data_ptr()

        def aliased(p, off):
            s = self.opt.state.get(p)
            return bool(s) and s["exp_avg"].data_ptr() == base_m + 4 * off and s["exp_avg_sq"].data_ptr() == base_v + 4 * off
        if not self.params:
            return
        probe = zip(self.params, self._views) if full else [(self.params[0], self._views[0])]
        if all(aliased(p, off) for p, (off, _) in probe):
            return
        self.pending = 0                    # the restored state carries its own step counters
        for p, (off, n) in zip(self.params, self._views):
            s = self.opt.state.get(p)
            m = eng.exp_avg[off:off + n].view(p.shape)
            v = eng.exp_avg_sq[off:off + n].view(p.shape)
            if not s:
                m.zero_(); v.zero_()
                self.opt.state[p] = {"step": torch.tensor(0.0, dtype=torch.float32), "exp_avg": m, "exp_avg_sq": v}
                continue
            if not aliased(p, off):
                m.copy_(s["exp_avg"]); v.copy_(s["exp_avg_sq"])
                s["exp_avg"], s["exp_avg_sq"] = m, v
            if torch.is_tensor(s["step"]) and s["step"].is_cuda:
                s["step"] = s["step"].detach().to("cpu", torch.float32)

    def resync(self):
        """Before a run of engine-side steps: the optimizer may have been stepped the plain way in between (its own
        counters moved; the moments are shared memory) or had a state dict loaded (``rehome``).  The optimizer's per-parameter
        ``step`` counters are brought up to date lazily (``flush``: state_dict(), a plain step, a re-bind) -- forty CPU tensor
        increments per step were a third of ``EngineAdam.step()``'s host time."""
        self.rehome()
        s = float(self.opt.state[self.params[0]]["step"]) + self.pending
        if s != self.engine_step:
            self.flush()
            self.engine.step_count.fill_(s)
            self.engine_step = s
        self.sync_hparams()

    def flush(self):
        """Bring the optimizer's per-parameter `step` counters up to date with the engine's."""
        if self.pending:
            for p in self.params:
                self.opt.state[p]["step"] += self.pending
            self.pending = 0

    def grads_in_flat(self, full: bool = True) -> bool:
        """Every ``p.grad`` is the parameter's view of the engine's flat gradient buffer.  ``full=False`` probes the first and the
        last parameter only (the engine's autograd node installs all views together; zero_grad / a foreign backward replace
        them together)."""
        base = self.engine.flat_g.data_ptr()
        probe = zip(self.params, self._views) if full else ((self.params[0], self._views[0]), (self.params[-1], self._views[-1]))
        for p, (off, _) in probe:
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                return False
        return True


def bind(opt, model) -> Optional[Binding]:
    """Attach ``opt`` to the step engine of ``model`` (creating the engine if the model supports one); ``None`` when the
    optimizer is not a plain Adam over exactly the model's parameters, or the model has no engine."""
    b = getattr(opt, "_cal_binding", None)
    if b is not None and b.engine is getattr(model, "_engine", None) and b.engine is not None:
        b.rehome(full=True)
        return b
    if not _plain_adam(opt):
        return None
    params = list(model.parameters())
    gp = opt.param_groups[0]["params"]
    if len(gp) != len(params) or any(a is not b_ for a, b_ in zip(gp, params)):
        return None
    get = getattr(model, "engine", None)
    eng = get() if callable(get) else None
    if eng is None:
        return None
    # Adam state: either untouched (fresh optimizer) or complete with one common step count
    states = [opt.state.get(p, {}) for p in params]
    have = [len(s) > 0 for s in states]
    if any(have) and not all(have):
        return None
    step = 0.0
    if all(have) and params:
        steps = {float(s["step"]) for s in states}
        if len(steps) != 1 or any(torch.is_tensor(s["step"]) and s["step"].is_cuda for s in states):
            return None
        step = steps.pop()
    b = Binding(opt, eng, params)
    for p, s, (off, n) in zip(params, states, b._views):
        m = eng.exp_avg[off:off + n].view(p.shape)
        v = eng.exp_avg_sq[off:off + n].view(p.shape)
        if s:
            m.copy_(s["exp_avg"])
            v.copy_(s["exp_avg_sq"])
            s["exp_avg"], s["exp_avg_sq"] = m, v
        else:
            m.zero_()
            v.zero_()
            opt.state[p] = {"step": torch.tensor(0.0, dtype=torch.float32), "exp_avg": m, "exp_avg_sq": v}
    eng.step_count.fill_(step)
    b.engine_step = step
    b.sync_hparams()
    opt._cal_binding = b
    return b


class EngineAdam(torch.optim.Adam):
    """``Adam(params, lr, betas, eps, weight_decay)`` (train_causal.py:21,76) whose ``step()`` is the engine's one-launch
    Adam when the parameters live in a StepEngine and their gradients in its flat gradient buffer."""

    def _binding(self) -> Optional[Binding]:
        b = getattr(self, "_cal_binding", None)
        if b is not None:
            return b
        params = self.param_groups[0]["params"] if len(self.param_groups) == 1 else []
        eng = _engine_of(params) if params else None
        if eng is None or eng.model is None:
            return None
        return bind(self, eng.model)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        b = self._binding()
        # every parameter's .grad must alias its slice of the flat buffer (~40 pointer compares): a frozen layer (grad None) or
        # a replaced gradient in the middle of the list sends the step to torch's Adam, which skips / uses it as torch would
        if b is None or not b.grads_in_flat():
            if b is not None:
                b.flush()
                b.grads_mixed = True                # some gradients live outside the flat buffer: check all of them from now on
            super().step()
            return loss                             # (the next engine-side step resyncs the device counter)
        b.resync()
        b.engine.adam()
        b.stepped()
        return loss

    def zero_grad(self, set_to_none: bool = True):
        """``optimizer.zero_grad()`` of the reference loop (train_causal.py:175).  Same effect as the parent's for the default
        ``set_to_none=True``; its per-parameter bookkeeping (profiler scope, foreach grouping) was 34 us of host time per step."""
        b = getattr(self, "_cal_binding", None)
        if set_to_none and b is not None:
            for p in b.params:
                p.grad = None
            return
        super().zero_grad(set_to_none)

    def state_dict(self):
        b = getattr(self, "_cal_binding", None)
        if b is not None:
            b.flush()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        b = getattr(self, "_cal_binding", None)
        if b is not None:                           # the loaded moments are fresh tensors: move them into the engine's buffers
            b.rehome(full=True)
