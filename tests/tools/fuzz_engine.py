"""Random-shape sweep of the step engine against the CPU oracle (test infrastructure, run by hand on a GPU box):

    python tests/tools/fuzz_engine.py [seconds] [seed]

Draws (hidden, layers, features, classes, graph sizes) at random -- including sizes around the per-graph kernels' tile
edges (1, 2, 31..33, 63..65, 127..129) -- and a model variant (GCN / GAT / GIN backbone, add / cat readout, the two
ablation flags), runs one training step through
cal_engine_step and through oracle.cal_oracle.CpuTrainer and reports every case whose logits / losses / gradients differ
farther from the same step in fp64 than 8x the fp32 oracle's own distance (floor 1e-4 of the tensor's scale).

Every numeric report is then RESOLVED mechanically (round-5 review item 4; ``resolve``) -- nothing is attributed by eye:

  flips   the engine's ReLU decisions are read back from its activation buffers (``h``, ``hco``, ``y1``, GIN's inner
          BatchNorm from ``gt1``; cal_engine_buffer_offset) and compared, site by site, with the sign of the fp64 oracle's
          pre-activations.  Every element that differs is printed as (site, row, column, |pre-activation| / scale of the
          site); the fp64 step is then re-evaluated WITH THE ENGINE'S MASKS at every site, and the report is resolved when
          every tensor falls back inside the bound of the sweep (8 x the fp32 oracle's own distance, floor 1e-4 of the scale).
  cond    the fp32 oracle itself, with its inputs and parameters moved by one ulp (relative +-2^-23 noise, eight draws, four
          of them with BatchNorm sums in fp32 as a CUDA device evaluates the reference instead of torch-CPU's double
          accumulators), moves as far from the fp64 step as the engine is: the case is ill-conditioned (BatchNorm over 2-3
          pooled rows, ...), the engine is inside 8 x that spread.
  UNRESOLVED  neither: a defect until shown otherwise.

    python tests/tools/fuzz_engine.py [seconds] [seed]
    python tests/tools/fuzz_engine.py --case 5504146          (replay ONE case of seed 5504 and resolve it verbosely)"""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_engine as T                     # noqa: E402  (helpers: _ragged_batch, _engine, _args)
from oracle import cal_oracle as O              # noqa: E402

EDGE_SIZES = [1, 2, 3, 7, 15, 16, 17, 31, 32, 33, 48, 57, 63, 64, 65, 96, 97, 127, 128, 129, 140]


def one_case(rng):
    hidden = rng.choice([16, 32, 48, 64, 80, 128, 128, 128, 256])
    layers = rng.choice([0, 1, 2, 3, 3, 4])
    nfeat = rng.choice([1, 3, 7, 10, 10, 37, 64, 65, 139])
    ncls = rng.choice([2, 3, 4, 4, 10])
    nb = rng.choice([2, 3, 5, 16, 17, 40, 100, 128, 129, 200, 300])      # (B = 1: BatchNorm over one graph raises in the reference)
    big = rng.random() < 0.3
    sizes = []
    for _ in range(nb):
        if rng.random() < 0.25:
            n = rng.choice(EDGE_SIZES)
        else:
            n = rng.randint(1, 64 if not big else 128)
        if not big:
            n = min(n, 64)
        sizes.append(n)
    if sum(sizes) > 6000:                        # keep the oracle (fp32 + fp64 step) in seconds
        sizes = sizes[: max(2, 6000 // max(sizes))]
    return hidden, layers, nfeat, ncls, sizes


def one_variant(rng, hidden):
    """(model, oracle / args keywords): every variant opts.get_model can build."""
    name = rng.choice(["CausalGCN", "CausalGCN", "CausalGAT", "CausalGIN"])
    if name == "CausalGAT" and hidden % 4:
        name = "CausalGCN"
    kw = {}
    if rng.random() < 0.3:
        kw["cat_or_add"] = "cat"
    if rng.random() < 0.15:
        kw["without_node_attention"] = True
    if rng.random() < 0.15:
        kw["without_edge_attention"] = True
    return name, kw


def run(case, seed, name="CausalGCN", kw=None, autograd=False, want_ctx=False):
    """autograd: through the nn.Module surface (forward by the engine, the loss by torch, cal_engine_backward_from) instead
    of the one-call training step."""
    from cal_amd import model as M
    from cal_amd.engine import StepEngine
    kw = dict(kw or {})
    hidden, layers, nfeat, ncls, sizes = case
    torch.manual_seed(seed)
    b = T._ragged_batch(seed, nfeat, sizes)
    bd = T._ragged_batch(seed, nfeat, sizes).to(T.DEV)
    b.y = b.y % ncls
    bd.y = bd.y % ncls
    sd = O.init_state(name, nfeat, ncls, hidden=hidden, layers=layers, heads=4, cat_or_add=kw.get("cat_or_add", "add"))
    g = torch.Generator().manual_seed(7)
    for k in list(sd):
        if k.endswith(".bias") or ("bn" in k and k.endswith(".weight")) or k.endswith(".nn.1.weight"):
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
    m = getattr(M, name)(nfeat, ncls, T._args(hidden=hidden, layers=layers, **kw))
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=name != "CausalGIN")
    m = m.to(T.DEV).train()
    if name == "CausalGAT":
        for c in m.convs:
            c.dropout = 0.0
    eng = StepEngine(m, lr=1e-3)
    okw = dict(layers=layers, heads=4, gat_dropout=0.0, **kw)
    ctx = {}
    B = len(sizes)
    perm = torch.randperm(B)
    tr = O.CpuTrainer(name, {k: v.clone() for k, v in sd.items()}, ncls, lr=1e-3, **okw)
    loss, lc, lo, lco, logits = tr.step(b.x, b.edge_index, b.batch, b.y, perm=perm)
    # the same step in fp64: what separates a defect from the conditioning of the case (BatchNorm over 2 graphs, ...)
    sd64 = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    tr64 = O.CpuTrainer(name, sd64, ncls, lr=1e-3, **okw)
    loss64, _, _, _, logits64 = tr64.step(b.x.double(), b.edge_index, b.batch, b.y, perm=perm)
    if autograd:
        m.zero_grad()
        out = m(bd, perm=perm.to(T.DEV))
        loss_t = O.causal_loss(*out, bd.y, ncls)[0]
        loss_t.backward()
        stats = np.array([loss_t.detach().item()])
        lp = torch.stack([o.detach().cpu() for o in out])
    else:
        stats = eng.train_step(bd, perm.to(T.DEV), adam=True).cpu().numpy()       # Adam inside k_finish (or k_adam behind it)
        eng.check_status()
        lp = eng.buffer("logp", 3 * B * ncls).view(3, B, ncls).cpu()
    src = getattr(m, "_engine", None) if autograd else eng          # (module surface: the model's own engine ran the forward)
    masks = engine_masks(src if src is not None else eng, name, sd, layers, hidden, int(bd.batch.numel()), B)      # (before the eval forward below overwrites the buffers)
    bad = []
    judged = []

    def judge(name, mine, ref32, ref64, floor, absolute=None):
        e_mine = (mine.double() - ref64).abs().max().item()
        e_ref = (ref32.double() - ref64).abs().max().item()
        scale = ref64.abs().max().item()
        judged.append((name, mine.double().clone(), e_ref, floor))
        if not e_mine <= max(8.0 * e_ref, floor * max(scale, 1.0)):
            bad.append("%s: engine %.3g vs fp32 oracle %.3g off the fp64 step (scale %.3g)" % (name, e_mine, e_ref, scale))
        # north_star's bound is ABSOLUTE (1e-4 on the logits): wherever the fp32 oracle itself is well inside it (a quarter),
        # the engine must be inside it too, whatever the scale of the log-probabilities
        elif absolute is not None and e_ref < 0.25 * absolute and not e_mine < absolute:
            bad.append("%s: engine %.3g off the fp64 step, above the absolute bound %.1g (fp32 oracle %.3g)" % (name, e_mine, absolute, e_ref))

    for hd in range(3):
        judge("logits head %d" % hd, lp[hd], logits[hd].detach(), logits64[hd].detach(), 1e-4, absolute=1e-4)
    judge("loss", torch.tensor(float(stats[0])), loss, loss64, 1e-4)
    for k, p in m.named_parameters():
        gref = tr.sd[k].grad
        if gref is not None:
            judge("grad " + k, p.grad.cpu(), gref, tr64.sd[k].grad, 1e-4)
        elif float(p.grad.abs().max()) != 0.0:       # switched-off attention MLP / conv_feat.bias: no gradient in the reference
            bad.append("grad %s: %.3g where the reference has none" % (k, float(p.grad.abs().max())))
        if gref is not None and not autograd:
            # the Adam update, where the gradient is not numerically zero (lr * sign(g) of a 1e-9 gradient is rounding noise)
            # and the engine's gradient agrees with the oracle's to begin with
            mask = (gref.abs() > 1e-5) & ((p.grad.cpu() - gref).abs() <= 1e-5 + 1e-2 * gref.abs())
            if mask.any() and not torch.allclose(p.detach().cpu()[mask], tr.sd[k].detach()[mask], atol=5e-5, rtol=1e-3):
                bad.append("param %s after Adam: %.3g" % (k, (p.detach().cpu()[mask] - tr.sd[k].detach()[mask]).abs().max().item()))
    if not autograd:
        # eval-mode forward of the stepped model (running statistics, post-Adam parameters) against the oracle on the same state
        m.eval()
        sde = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if not k.endswith(".eps")}
        fkw = {k: v for k, v in okw.items() if k != "gat_dropout"}
        ref = O.causal_forward(name, sde, b.x, b.edge_index, b.batch, perm=perm, training=False, **fkw)
        out = eng.forward(bd, perm.to(T.DEV), training=False)
        for hd, (r, t) in enumerate(zip(ref, out)):
            d = (r.detach() - t.cpu()).abs().max().item()
            if not d <= 2e-4 * max(1.0, r.abs().max().item()):
                bad.append("eval logits head %d: %.3g (scale %.3g)" % (hd, d, r.abs().max().item()))
    ctx.update(name=name, sd=sd, ncls=ncls, okw=okw, b=b, perm=perm, masks=masks, judged=judged, params=[k for k, _ in m.named_parameters()])
    if want_ctx:
        return bad, ctx
    return bad



# ------------------------------------------------------------------------------------------------------------------------
# mechanical resolution of a numeric report
# ------------------------------------------------------------------------------------------------------------------------
class _ReluShim:
    """Stands in for ``torch.nn.functional`` inside oracle.cal_oracle for one forward: ``relu`` records the pre-activation of
    every call (the sites come in program order: feature layer, backbone layers -- GIN: inner then outer --, context, objects,
    readouts c / o / co) and, when ``force`` holds a mask for the site, multiplies by that mask instead of clamping."""

    def __init__(self, force=None, bn32=False):
        self.pre, self.force, self.bn32 = [], force, bn32

    def batch_norm(self, x, rm, rv, w, b, training, momentum, eps):
        """bn32: BatchNorm the way a CUDA device evaluates the reference -- ``acc_type<float, true>`` is float, so the batch
        statistics, the normalisation and (through autograd) every backward sum are plain fp32; torch's CPU kernel, which the
        oracle otherwise runs, accumulates them in DOUBLE (``acc_type<float, false>``) and is that much closer to the fp64 step
        than any fp32 device evaluation can be when the batch is 2-3 rows deep."""
        if not (self.bn32 and training and x.dtype == torch.float32):
            return torch.nn.functional.batch_norm(x, rm, rv, w, b, training, momentum, eps)
        mean = x.mean(0)
        xc = x - mean
        var = (xc * xc).mean(0)
        with torch.no_grad():
            n = x.size(0)
            rm.mul_(1 - momentum).add_(momentum * mean.detach())
            rv.mul_(1 - momentum).add_(momentum * var.detach() * (n / max(n - 1, 1)))
        return xc * torch.rsqrt(var + eps) * w + b

    def __getattr__(self, n):
        return getattr(torch.nn.functional, n)

    def relu(self, x):
        k = len(self.pre)
        self.pre.append(x.detach().clone())
        if self.force is not None and self.force[k] is not None:
            return x * self.force[k].to(x.dtype)
        return torch.nn.functional.relu(x)


def site_names(name, layers):
    s = ["feature layer"]
    for i in range(layers):
        s += (["GIN layer %d inner (after BatchNorm)" % i, "GIN layer %d outer" % i] if name == "CausalGIN" else ["backbone layer %d" % i])
    return s + ["context conv", "objects conv", "readout c fc1", "readout o fc1", "readout co fc1"]


def engine_masks(eng, name, sd, layers, hidden, N, B):
    """The engine's ReLU decisions of its LATEST training forward, one {0,1} tensor per site (CPU) in the oracle's site order:
    post-activation buffers > 0.  GIN's inner ReLU sits behind a BatchNorm the fused kernels re-evaluate from ``gt1`` in both
    passes: the same arithmetic is repeated here (fp64 column sums of the fp32 values, fp32 scale / shift, one fma)."""
    H = hidden
    h = eng.buffer("h", (layers + 1) * N * H).view(layers + 1, N, H).cpu()
    hco = eng.buffer("hco", 2 * N * H).view(2, N, H).cpu()
    y1 = eng.buffer("y1", 3 * B * H).view(3, B, H).cpu()
    out = [(h[0] > 0)]
    for i in range(layers):
        if name == "CausalGIN":
            t1 = eng.buffer("gt1", layers * N * H).view(layers, N, H)[i].cpu()
            t1d = t1.double()
            mean = t1d.mean(0)
            var = ((t1d * t1d).mean(0) - mean * mean).clamp_min(0.0)
            rstd = 1.0 / torch.sqrt(var.float() + 1e-5)
            sc = sd["convs.%d.nn.1.weight" % i].float() * rstd
            sh = sd["convs.%d.nn.1.bias" % i].float() - mean.float() * sc
            out.append(torch.addcmul(sh, t1, sc) > 0)
        out.append(h[i + 1] > 0)
    out += [hco[0] > 0, hco[1] > 0, y1[0] > 0, y1[1] > 0, y1[2] > 0]
    return out


def _step64(ctx, force=None, x=None, dtype=torch.float64, sd=None, bn32=False):
    """One oracle train step (fp64 unless ``dtype`` says otherwise) on the report's case; returns (shim, trainer, logits)."""
    sdx = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in (sd or ctx["sd"]).items()}
    tr = O.CpuTrainer(ctx["name"], sdx, ctx["ncls"], lr=1e-3, **ctx["okw"])
    b = ctx["b"]
    shim = _ReluShim(force, bn32)
    keep = O.F
    O.F = shim
    try:
        out = tr.step((b.x if x is None else x).to(dtype), b.edge_index, b.batch, b.y, perm=ctx["perm"])
    finally:
        O.F = keep
    return shim, tr, out


def _values(ctx, tr, out):
    """The judged tensors of a step in the order ``run`` judged them: three logits, the loss, every gradient the reference has."""
    vals = {"logits head %d" % hd: out[4][hd].detach().double() for hd in range(3)}
    vals["loss"] = out[0].detach().double()
    for k in ctx["params"]:
        if tr.sd[k].grad is not None:
            vals["grad " + k] = tr.sd[k].grad.detach().double()
    return vals


def resolve(ctx, verbose=True, log=print):
    """Classify a numeric report: returns ("flips" | "cond" | "UNRESOLVED", lines)."""
    lines = []
    names = site_names(ctx["name"], ctx["okw"]["layers"])
    shim, tr64, out64 = _step64(ctx)
    assert len(shim.pre) == len(names) == len(ctx["masks"]), (len(shim.pre), len(names), len(ctx["masks"]))
    nflip = 0
    masks = list(ctx["masks"])
    for k, (z, mk) in enumerate(zip(shim.pre, masks)):
        diff = (z > 0) != mk
        n = int(diff.sum())
        if n > 0.01 * z.numel() + 4:
            # not this forward's activations: the one-launch readout (k_ro_step) keeps y1 in LDS and never stores it -- the site is
            # left to the oracle's own ReLU (a flip THERE would stay unexplained and show as "cond" or UNRESOLVED below)
            lines.append("   site %-34s not readable from the engine (%d of %d signs differ: the buffer was not written by this forward)"
                         % (names[k], n, z.numel()))
            masks[k] = None
            continue
        if n:
            nflip += n
            scale = z.abs().max().item()
            idx = diff.nonzero()
            worst = (z.abs()[diff] / max(scale, 1e-300)).max().item()
            lines.append("   site %-34s %4d flipped of %d; max |z| / scale %.2e (scale %.3g); first: %s" % (
                names[k], n, z.numel(), worst, scale,
                ", ".join("(row %d, col %d, z %.2e)" % (int(r), int(c), z[int(r), int(c)].item()) for r, c in idx[:3].tolist())))
    ref = _values(ctx, tr64, out64)
    res = {}
    if nflip:
        _, trm, outm = _step64(ctx, force=masks)
        res = _values(ctx, trm, outm)
    # conditioning: the fp32 oracle with its inputs AND parameters moved by one ulp (relative noise of +-2^-23, eight draws): the
    # rounding a different -- equally valid -- fp32 evaluation order injects at every layer, not only at the input
    spread = {}
    g = torch.Generator().manual_seed(99)

    def ulp(t):
        return t * (1.0 + (torch.rand(t.shape, generator=g) - 0.5) * 2.0 ** -22)

    for draw in range(8):
        xp = ulp(ctx["b"].x)
        sdp = {k: (ulp(v) if v.is_floating_point() and "running" not in k else v) for k, v in ctx["sd"].items()}
        _, trp, outp = _step64(ctx, x=xp, dtype=torch.float32, sd=sdp, bn32=draw >= 4)       # (four of them with fp32 BatchNorm sums)
        for k, v in _values(ctx, trp, outp).items():
            spread[k] = max(spread.get(k, 0.0), (v - ref[k]).abs().max().item())
    verdicts = []
    for nm, mine, e_ref, floor in ctx["judged"]:
        scale = ref[nm].abs().max().item()
        e_mine = (mine - ref[nm]).abs().max().item()
        bound = max(8.0 * e_ref, floor * max(scale, 1.0))
        if e_mine <= bound:
            continue
        e_forced = (mine - res[nm]).abs().max().item() if nm in res else float("inf")
        by_flips = e_forced <= bound
        by_cond = e_mine <= 8.0 * max(spread.get(nm, 0.0), e_ref)
        verdicts.append("flips" if by_flips else ("cond" if by_cond else "UNRESOLVED"))
        lines.append("   %-34s engine %.3g off the fp64 step (fp32 oracle %.3g, bound %.3g); with the engine's masks %.3g; "
                     "fp32 oracle under one-ulp inputs %.3g  -> %s" % (nm, e_mine, e_ref, bound, e_forced, spread.get(nm, 0.0), verdicts[-1]))
    verdict = "UNRESOLVED" if "UNRESOLVED" in verdicts else ("flips" if "flips" in verdicts else ("cond" if verdicts else "flips"))
    if verbose:
        for ln in lines:
            log(ln)
    return verdict, lines


def replay(seed, n):
    """The n-th case (1-based) of the sweep with this seed, exactly as main() draws it."""
    rng = random.Random(seed)
    for k in range(1, n + 1):
        case = one_case(rng)
        name, kw = one_variant(rng, case[0])
        ag = rng.random() < 0.4
    return case, name, kw, ag


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--case":
        full = int(sys.argv[2])
        case, name, kw, ag = replay(full // 1000, full % 1000)
        bad, ctx = run(case, full, name, kw, autograd=ag, want_ctx=True)
        print("case %d: %s%s %s hidden=%d layers=%d nfeat=%d ncls=%d B=%d sizes[:12]=%s" % (
            full, name, " (module surface)" if ag else "", kw, case[0], case[1], case[2], case[3], len(case[4]), case[4][:12]))
        for ln in bad:
            print("   report: " + ln)
        if bad:
            print("   => " + resolve(ctx)[0])
        return
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    t0 = time.time()
    n = nbad = 0
    tally = {"flips": 0, "cond": 0, "UNRESOLVED": 0, "declined": 0}
    while time.time() - t0 < budget:
        case = one_case(rng)
        name, kw = one_variant(rng, case[0])
        ag = rng.random() < 0.4
        n += 1
        ctx = None
        try:
            bad, ctx = run(case, seed * 1000 + n, name, kw, autograd=ag, want_ctx=True)
        except Exception as ex:                  # noqa: BLE001
            bad = ["exception: %r" % (ex,)]
        if bad:
            nbad += 1
            h, l, f, c, sizes = case
            print("MISMATCH %s%s %s hidden=%d layers=%d nfeat=%d ncls=%d B=%d sizes[:12]=%s seed=%d: %s"
                  % (name, " (module surface)" if ag else "", kw, h, l, f, c, len(sizes), sizes[:12], seed * 1000 + n, "; ".join(bad[:4])), flush=True)
            if ctx is None:
                tally["declined"] += 1           # the engine declined the shape (ValueError): not a numeric report
            else:
                try:
                    verdict, _ = resolve(ctx)
                except Exception as ex:          # noqa: BLE001
                    verdict = "UNRESOLVED"
                    print("   resolve failed: %r" % (ex,))
                tally[verdict] += 1
                print("   => %s" % verdict, flush=True)
    print("fuzz: %d cases, %d mismatching, %.0f s; numeric reports resolved by ReLU flips %d, by conditioning %d, UNRESOLVED %d; "
          "shapes the engine declines %d" % (n, nbad, time.time() - t0, tally["flips"], tally["cond"], tally["UNRESOLVED"], tally["declined"]))


if __name__ == "__main__":
    main()
