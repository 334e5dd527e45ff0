"""The random-shape sweep of fuzz_engine.py for the HOST library (libcalhost.so, CPU tensors; runs without a GPU):

    python tests/tools/fuzz_host.py [seconds] [seed]

The nn.Module path of cal_amd.model on CPU tensors routes every operator (and its autograd backward) to the plain-C++
twin of the C-ABI; one forward + external loss + backward per case against the oracle, judged like fuzz_engine.py
(8x the fp32 oracle's own distance from the fp64 step, floor 1e-4 of the tensor's scale)."""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_engine as F                          # noqa: E402
import test_gpu_engine as T                      # noqa: E402
from oracle import cal_oracle as O               # noqa: E402


def run(case, seed, name, kw):
    from cal_amd import model as M
    hidden, layers, nfeat, ncls, sizes = case
    torch.manual_seed(seed)
    b = T._ragged_batch(seed, nfeat, sizes)
    b.y = b.y % ncls
    sd = O.init_state(name, nfeat, ncls, hidden=hidden, layers=layers, heads=4, cat_or_add=kw.get("cat_or_add", "add"))
    g = torch.Generator().manual_seed(7)
    for k in list(sd):
        if k.endswith(".bias") or ("bn" in k and k.endswith(".weight")) or k.endswith(".nn.1.weight"):
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
    m = getattr(M, name)(nfeat, ncls, T._args(hidden=hidden, layers=layers, **kw))
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=name != "CausalGIN")
    m.train()
    if name == "CausalGAT":
        for c in m.convs:
            c.dropout = 0.0
    okw = dict(layers=layers, heads=4, gat_dropout=0.0, **kw)
    perm = torch.randperm(len(sizes))
    tr = O.CpuTrainer(name, {k: v.clone() for k, v in sd.items()}, ncls, lr=1e-3, **okw)
    loss, _, _, _, logits = tr.step(b.x, b.edge_index, b.batch, b.y, perm=perm)
    sd64 = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    tr64 = O.CpuTrainer(name, sd64, ncls, lr=1e-3, **okw)
    loss64, _, _, _, logits64 = tr64.step(b.x.double(), b.edge_index, b.batch, b.y, perm=perm)
    out = m(b, perm=perm)
    loss_t = O.causal_loss(*out, b.y, ncls)[0]
    loss_t.backward()
    bad = []

    def judge(nm, mine, ref32, ref64, floor):
        e_mine = (mine.double() - ref64).abs().max().item()
        e_ref = (ref32.double() - ref64).abs().max().item()
        scale = ref64.abs().max().item()
        if not e_mine <= max(8.0 * e_ref, floor * max(scale, 1.0)):
            bad.append("%s: host %.3g vs fp32 oracle %.3g off the fp64 step (scale %.3g)" % (nm, e_mine, e_ref, scale))

    for hd in range(3):
        judge("logits head %d" % hd, out[hd].detach(), logits[hd].detach(), logits64[hd].detach(), 1e-4)
    judge("loss", loss_t.detach(), loss, loss64, 1e-4)
    for k, p in m.named_parameters():
        gref = tr.sd[k].grad
        if gref is not None:
            judge("grad " + k, p.grad, gref, tr64.sd[k].grad, 1e-4)
        elif p.grad is not None and float(p.grad.abs().max()) != 0.0:
            bad.append("grad %s: %.3g where the reference has none" % (k, float(p.grad.abs().max())))
    return bad


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    t0 = time.time()
    n = nbad = 0
    while time.time() - t0 < budget:
        case = F.one_case(rng)
        h, l, f, c, sizes = case
        if sum(sizes) > 2500:                     # the host library is a plain-loop implementation
            sizes = sizes[: max(2, 2500 // max(sizes))]
            case = (h, l, f, c, sizes)
        name, kw = F.one_variant(rng, h)
        n += 1
        try:
            bad = run(case, seed * 1000 + n, name, kw)
        except Exception as ex:                   # noqa: BLE001
            bad = ["exception: %r" % (ex,)]
        if bad:
            nbad += 1
            print("MISMATCH %s %s hidden=%d layers=%d nfeat=%d ncls=%d B=%d sizes[:12]=%s seed=%d: %s"
                  % (name, kw, h, l, f, c, len(sizes), sizes[:12], seed * 1000 + n, "; ".join(bad[:4])), flush=True)
    print("fuzz_host: %d cases, %d mismatching, %.0f s" % (n, nbad, time.time() - t0))


if __name__ == "__main__":
    main()
