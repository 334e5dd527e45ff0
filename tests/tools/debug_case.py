"""Ad-hoc: for one fuzz case, compare the engine's readout-side intermediates (pooled rows, d pooled) with the oracle in fp32 / fp64.
    python tests/tools/debug_case.py 6601011"""
import importlib.util
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("fz", os.path.join(here, "fuzz_engine.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
O, T = fz.O, fz.T


def main():
    full = int(sys.argv[1])
    case, name, kw, ag = fz.replay(full // 1000, full % 1000)
    hidden, layers, nfeat, ncls, sizes = case
    from cal_amd import model as M
    from cal_amd.engine import StepEngine
    torch.manual_seed(full)
    b = T._ragged_batch(full, nfeat, sizes)
    bd = T._ragged_batch(full, nfeat, sizes).to(T.DEV)
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
    eng = StepEngine(m, lr=1e-3)
    B = len(sizes)
    perm = torch.randperm(B)
    okw = dict(layers=layers, heads=4, gat_dropout=0.0, **kw)
    res = {}
    for dt in (torch.float32, torch.float64):
        sdx = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        for k in O.trainable(sdx):
            sdx[k].requires_grad_(True)
        logits, im = O.causal_forward(name, sdx, b.x.to(dt), b.edge_index, b.batch, perm=perm, training=True, return_intermediates=True, **okw)
        for t in im.values():
            t.retain_grad()
        loss = O.causal_loss(*logits, b.y, ncls)[0]
        loss.backward()
        res[dt] = dict(pool=torch.stack([im["xc_pool"], im["xo_pool"]]).detach(), dpool=torch.stack([im["xc_pool"].grad, im["xo_pool"].grad]),
                       logits=torch.stack([l.detach() for l in logits]), dxc=im["xc"].grad, dxo=im["xo"].grad,
                       gw=sdx["context_convs.weight"].grad, x=im["x"].detach(), dx=im["x"].grad)
    eng.train_step(bd, perm.to(T.DEV), adam=False)
    eng.check_status()
    H = hidden
    N = int(bd.batch.numel())
    mine = dict(pool=eng.buffer("pooled", 2 * B * H).view(2, B, H).cpu(), dpool=eng.buffer("dpool", 2 * B * H).view(2, B, H).cpu(),
                logits=eng.buffer("logp", 3 * B * ncls).view(3, B, ncls).cpu(), gw=m.context_convs.weight.grad.cpu(),
                x=eng.buffer("h", (layers + 1) * N * H).view(layers + 1, N, H)[layers].cpu())
    r64 = res[torch.float64]
    for k in ("x", "pool", "logits", "dpool", "gw"):
        for br in range(mine[k].size(0) if k in ("pool", "dpool", "logits") else 1):
            a = mine[k][br] if k in ("pool", "dpool", "logits") else mine[k]
            r32 = res[torch.float32][k][br] if k in ("pool", "dpool", "logits") else res[torch.float32][k]
            r = r64[k][br] if k in ("pool", "dpool", "logits") else r64[k]
            print("%-7s[%d] scale %.3g   engine - fp64 %.3g   fp32 oracle - fp64 %.3g" % (k, br, r.abs().max().item(), (a.double() - r).abs().max().item(), (r32.double() - r).abs().max().item()))
    # the co head alone: the engine's x_co against the oracle's, and the readout re-evaluated from the ENGINE's x_co
    xco_e = eng.buffer("xco", B * H).view(B, H).cpu()
    xco64 = r64["pool"][0][perm] + r64["pool"][1]
    print("x_co: engine - fp64 %.3g (scale %.3g)" % ((xco_e.double() - xco64).abs().max().item(), xco64.abs().max().item()))
    sd64 = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    sd32 = {k: v.clone() for k, v in sd.items()}
    lg64 = O._readout(xco_e.double(), sd64, "co", True)
    lg32 = O._readout(xco_e.clone(), sd32, "co", True)
    print("co logits from the engine's x_co: fp64 readout vs engine %.3g; fp32 torch readout vs fp64 readout %.3g; fp64 readout vs fp64 step %.3g" % (
        (lg64 - mine["logits"][2].double()).abs().max().item(), (lg32.double() - lg64).abs().max().item(), (lg64 - r64["logits"][2]).abs().max().item()))
    # the same readout in fp32 with BatchNorm as x * sc + sh (the engine's form) instead of (x - mean) * rstd * g + b
    def ro_scale_shift(x, tag):
        def bn(x, name):
            m = x.double().mean(0); v = (x.double() * x.double()).mean(0) - m * m
            rstd = 1.0 / torch.sqrt(v.clamp_min(0).float() + 1e-5)
            sc = sd[name + ".weight"] * rstd; sh = sd[name + ".bias"] - m.float() * sc
            return x * sc + sh
        y = torch.relu(torch.nn.functional.linear(bn(x, "fc1_bn_" + tag), sd["fc1_%s.weight" % tag], sd["fc1_%s.bias" % tag]))
        z = torch.nn.functional.linear(bn(y, "fc2_bn_" + tag), sd["fc2_%s.weight" % tag], sd["fc2_%s.bias" % tag])
        return torch.log_softmax(z, -1)
    print("co logits, fp32 torch with scale/shift BatchNorm vs fp64 readout: %.3g" % ((ro_scale_shift(xco_e.clone(), "co").double() - lg64).abs().max().item()))
    y1_64 = torch.nn.functional.linear(O._bn(xco_e.double(), {k: v.clone() for k, v in sd64.items()}, "fc1_bn_co", True), sd64["fc1_co.weight"], sd64["fc1_co.bias"])
    v2 = torch.relu(y1_64).var(0, unbiased=False)
    print("fc2_bn_co input: batch variance min %.3g, 5 smallest %s; |mean| max %.3g" % (v2.min().item(), v2.topk(5, largest=False).values.tolist(), torch.relu(y1_64).mean(0).abs().max().item()))
    if os.environ.get("CAL_AMD_RO_STEP") == "0":
        y1_e = eng.buffer("y1", 3 * B * H).view(3, B, H)[2].cpu()
        y1r = torch.relu(y1_64)
        err = (y1_e.double() - y1r).abs()
        print("y1 (co): engine - fp64 %.3g (scale %.3g); worst columns %s" % (err.max().item(), y1r.abs().max().item(), err.max(0).values.topk(4)))
        zl_e = eng.buffer("zl", 3 * B * ncls).view(3, B, ncls)[2].cpu()
        sdt = {k: v.clone() for k, v in sd64.items()}
        zl64 = torch.nn.functional.linear(O._bn(y1r, sdt, "fc2_bn_co", True), sd64["fc2_co.weight"], sd64["fc2_co.bias"])
        print("zl (co): engine - fp64 %.3g" % (zl_e.double() - zl64).abs().max().item())
        sdt = {k: v.clone() for k, v in sd64.items()}
        zl_from_e = torch.nn.functional.linear(O._bn(y1_e.double(), sdt, "fc2_bn_co", True), sd64["fc2_co.weight"], sd64["fc2_co.bias"])
        print("zl (co) in fp64 from the ENGINE's y1: vs engine %.3g, vs fp64 chain %.3g" % ((zl_e.double() - zl_from_e).abs().max().item(), (zl_from_e - zl64).abs().max().item()))
        v2e = y1_e.double().var(0, unbiased=False)
        small = (v2e < 1e-6) & (v2e > 0)
        print("columns of y1 with 0 < batch variance < 1e-6: %d; their values:" % int(small.sum()), y1_e[:, small].t().tolist()[:6], "fp64:", y1r[:, small].t().tolist()[:6])
    # condition of the readout BatchNorms: smallest batch variance per head input
    for br, nm in enumerate(("xc_pool", "xo_pool")):
        v = r64["pool"][br].var(0, unbiased=False)
        print("%s: batch variance min %.3g median %.3g; mean |x| %.3g" % (nm, v.min().item(), v.median().item(), r64["pool"][br].abs().mean().item()))
    d = (mine["dpool"][0].double() - r64["dpool"][0]).abs()
    print("d pooled (context): worst columns", d.max(0).values.topk(5))
    print("variance there", r64["pool"][0].var(0, unbiased=False)[d.max(0).values.topk(5).indices])


if __name__ == "__main__":
    main()
