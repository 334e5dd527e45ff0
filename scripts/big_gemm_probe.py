"""Probe of the 128x128 GEMM (gemm_big.hip) at the config-5 shape; with a CAL_BIG_DEBUG build the debug bits
knock out one phase at a time (1 C stores, 2 global loads in the loop, 4 LDS stores in the loop, 8 MFMAs)."""
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
dbg = getattr(h, "cal_debug_big", None)
if dbg: print("occupancy (WGs/CU) of k_gemm_big<NN>:", h.cal_debug_big_occ())
print(torch.cuda.get_device_properties(0))
for bits in ([0, 1, 2, 4, 6, 8, 9, 14, 15] if dbg else [0]):
    if dbg: dbg(bits)
    a, b = t(0, w), t(1, wt)
    print("dbg %2d: NN %7.1f us (%5.1f TF)   NT %7.1f us (%5.1f TF)" % (bits, a, 2 * M * N * K / a / 1e6, b, 2 * M * N * K / b / 1e6))
run(0, w); torch.cuda.synchronize()
print("max err NN", (y - x @ w).abs().max().item())
run(1, wt); torch.cuda.synchronize()
print("max err NT", (y - x @ w).abs().max().item())
