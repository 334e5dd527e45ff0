"""Random-shape sweep of the step engine against the CPU oracle (test infrastructure, run by hand on a GPU box):

    python tests/tools/fuzz_engine.py [seconds] [seed]

Draws (hidden, layers, features, classes, graph sizes) at random -- including sizes around the per-graph kernels' tile
edges (1, 2, 31..33, 63..65, 127..129) -- and a model variant (GCN / GAT / GIN backbone, add / cat readout, the two
ablation flags), runs one training step through
cal_engine_step and through oracle.cal_oracle.CpuTrainer and reports every case whose logits / losses / gradients differ
farther from the same step in fp64 than 8x the fp32 oracle's own distance (floor 1e-4 of the tensor's scale).

Reading a report: a mismatch confined to ONE row of a weight gradient (and what lies below it) with every other tensor at
1e-7 is a ReLU whose pre-activation sits within rounding of zero and flipped -- not a defect; errors spread over all
tensors of a head / layer are."""
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


def run(case, seed, name="CausalGCN", kw=None, autograd=False):
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
    bad = []

    def judge(name, mine, ref32, ref64, floor, absolute=None):
        e_mine = (mine.double() - ref64).abs().max().item()
        e_ref = (ref32.double() - ref64).abs().max().item()
        scale = ref64.abs().max().item()
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
    return bad


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    t0 = time.time()
    n = nbad = 0
    while time.time() - t0 < budget:
        case = one_case(rng)
        name, kw = one_variant(rng, case[0])
        ag = rng.random() < 0.4
        n += 1
        try:
            bad = run(case, seed * 1000 + n, name, kw, autograd=ag)
        except Exception as ex:                  # noqa: BLE001
            bad = ["exception: %r" % (ex,)]
        if bad:
            nbad += 1
            h, l, f, c, sizes = case
            print("MISMATCH %s%s %s hidden=%d layers=%d nfeat=%d ncls=%d B=%d sizes[:12]=%s seed=%d: %s"
                  % (name, " (module surface)" if ag else "", kw, h, l, f, c, len(sizes), sizes[:12], seed * 1000 + n, "; ".join(bad[:4])), flush=True)
    print("fuzz: %d cases, %d mismatching, %.0f s" % (n, nbad, time.time() - t0))


if __name__ == "__main__":
    main()
