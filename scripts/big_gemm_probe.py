"""Stand-alone timing + correctness probe of the node-level GEMMs at the config-5 shape [160k,256]x[256,256] through the public
cal_gemm entry: NN, NT and the TN weight gradient, each timed alone (events around one launch, a cache-flushing copy before it)
and as the back-to-back NT -> TN pair the backward issues.
usage on the GPU box: PYTHONPATH=. python scripts/big_gemm_probe.py"""
import torch
from cal_amd import _lib
from cal_amd.plan import _p, _stream
h = _lib.lib()
M, N, K = 160000, 256, 256
x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.05; y = torch.empty(M, N, device="cuda")
g = torch.randn(M, N, device="cuda")
wt = w.t().contiguous()
dw = torch.empty(K, N, device="cuda")
ws = torch.empty(max(_lib.query("cal_gemm_ws", K, N, M), 4), device="cuda")
junk = torch.empty(96 << 20, device="cuda")
def nn(): _lib.call("cal_gemm", 0, 0, _p(x), _p(w), _p(y), None, 0, None, M, N, K, _stream())
def nt(): _lib.call("cal_gemm", 0, 1, _p(g), _p(wt), _p(y), None, 0, None, M, N, K, _stream())
def tn(): _lib.call("cal_gemm", 1, 0, _p(x), _p(g), _p(dw), None, 0, _p(ws), K, N, M, _stream())
def timed(fs, it=8):
    tot = [0.0] * len(fs)
    for _ in range(it + 1):
        junk.add_(1.0)                                   # 384 MB through the caches
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(fs) + 1)]
        ev[0].record()
        for i, f in enumerate(fs):
            f(); ev[i + 1].record()
        torch.cuda.synchronize()
        if _ > 0:
            for i in range(len(fs)): tot[i] += ev[i].elapsed_time(ev[i + 1]) * 1e3 / it
    return tot
fl = 2 * M * N * K / 1e6
import sys
if len(sys.argv) > 1:
    mode = sys.argv[1]
    if mode == "zeros": x.zero_(); g.zero_()
    if mode == "coarse": x.copy_((x * 8).round() / 8); g.copy_((g * 8).round() / 8)       # few mantissa bits set
    if mode == "uniform": x.uniform_(-1, 1); g.uniform_(-1, 1)
    print("operand data:", mode)
for name, fs in (("NN", [nn]), ("NT", [nt]), ("TN", [tn]), ("NT,TN", [nt, tn]), ("NN,NT,TN", [nn, nt, tn])):
    t = timed(fs)
    print("%-9s" % name, "  ".join("%7.1f us (%5.1f TF)" % (v, fl / v) for v in t))
nn(); torch.cuda.synchronize(); print("max err NN", (y - x @ w).abs().max().item())
nt(); torch.cuda.synchronize(); print("max err NT", (y - g @ w).abs().max().item())
tn(); torch.cuda.synchronize(); print("max err TN", (dw - x.t() @ g).abs().max().item(), "scale", (x.t() @ g).abs().max().item())
