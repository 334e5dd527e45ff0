"""Stand-alone timing + correctness probe of the 128x128 GEMM (gemm_big.hip) at the config-5 shape [160k,256]x[256,256],
NN and NT, through the public cal_gemm entry.  (The phase knock-outs and per-workgroup timestamps quoted in DESIGN.md were
taken with temporary debug hooks in the kernel; they are not part of the committed source.)
usage on the GPU box: PYTHONPATH=. python scripts/big_gemm_probe.py"""
import ctypes, sys, torch
from cal_amd import _lib
from cal_amd.plan import _p, _stream
h = _lib.lib()
M, N, K = 160000, 256, 256
x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.05; y = torch.empty(M, N, device="cuda")
wt = w.t().contiguous()
def run(tb, bmat):
    _lib.call("cal_gemm", 0, tb, _p(x), _p(bmat), _p(y), None, 0, None, M, N, K, _stream())
def t(tb, bmat, it=10):
    run(tb, bmat); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): run(tb, bmat)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
a, b = t(0, w), t(1, wt)
print("NN %7.1f us (%5.1f TF)   NT %7.1f us (%5.1f TF)" % (a, 2 * M * N * K / a / 1e6, b, 2 * M * N * K / b / 1e6))
run(0, w); torch.cuda.synchronize()
print("max err NN", (y - x @ w).abs().max().item())
run(1, wt); torch.cuda.synchronize()
print("max err NT", (y - x @ w).abs().max().item())
