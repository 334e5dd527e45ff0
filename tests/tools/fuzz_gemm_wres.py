"""Random-shape sweep of the weight-resident GEMMs (gemm_wres.hip) through the public cal_gemm entry against torch in fp64
(test infrastructure, run by hand on a GPU box):  python tests/tools/fuzz_gemm_wres.py [cases] [seed]
M in [16384, 70000] (whole and partial last 32-row blocks), K, N in {128, 256}; NN, NT (bias / ReLU on and off) and the
256 x 256 TN gradient over a node axis of the same length.  Reports max |err| relative to the result's scale."""
import random, sys, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
from cal_amd import _lib
from cal_amd.plan import _p, _stream
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
VERBOSE = len(sys.argv) > 3
rng = random.Random(seed); torch.manual_seed(seed)
worst = {"NN": 0.0, "NT": 0.0, "TN": 0.0}; bad = 0
for c in range(cases):
    M = rng.choice([16384, 16385, 16415, 20000, 20011, rng.randrange(16384, 70000)])
    K, N = rng.choice([128, 256]), rng.choice([128, 256])
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.1
    b = torch.randn(N, device="cuda") if rng.random() < 0.5 else None
    relu = int(rng.random() < 0.5)
    ref = x.double() @ w.double()
    if b is not None: ref = ref + b.double()
    if relu: ref = ref.clamp_min(0)
    y = torch.empty(M, N, device="cuda")
    if VERBOSE: print("case", c, M, K, N, "bias" if b is not None else "-", relu, flush=True)
    for name, tb, bm in (("NN", 0, w), ("NT", 1, w.t().contiguous())):
        y.fill_(float("nan"))
        _lib.call("cal_gemm", 0, tb, _p(x), _p(bm), _p(y), None if b is None else _p(b), relu, None, M, N, K, _stream())
        if VERBOSE: torch.cuda.synchronize(); print("  ", name, "ok", flush=True)
        e = ((y.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
        worst[name] = max(worst[name], e)
        if not e < 2e-6: bad += 1; print("MISMATCH", name, M, K, N, "bias" if b is not None else "", "relu" if relu else "", e)
    g = torch.randn(M, 256, device="cuda"); x2 = torch.randn(M, 256, device="cuda")
    dw = torch.full((256, 256), float("nan"), device="cuda")
    ws = torch.empty(max(_lib.query("cal_gemm_ws", 256, 256, M), 4), device="cuda")
    _lib.call("cal_gemm", 1, 0, _p(x2), _p(g), _p(dw), None, 0, _p(ws), 256, 256, M, _stream())
    if VERBOSE: torch.cuda.synchronize(); print("   TN ok", flush=True)
    r2 = x2.double().t() @ g.double()
    e = ((dw.double() - r2).abs().max() / r2.abs().max()).item()
    worst["TN"] = max(worst["TN"], e)
    if not e < 2e-5: bad += 1; print("MISMATCH TN", M, e)
print("fuzz_gemm_wres: %d cases (seed %d), %d mismatching; worst relative error NN %.2e NT %.2e TN %.2e" % (cases, seed, bad, worst["NN"], worst["NT"], worst["TN"]))
