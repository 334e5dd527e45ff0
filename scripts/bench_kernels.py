"""Real (graph-replayed, un-profiled) latency of individual C-ABI kernels: n dependent launches are
captured into one hipGraph; per-launch time = replay time / n (includes the ~1.5 us boundary)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cal_amd import _lib
from cal_amd.plan import _p, _stream, GraphPlan

def graph_time(fn, n=50, reps=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n * 1e6

def gemm(M, N, K, ta=0, tb=0):
    A = torch.randn((K, M) if ta else (M, K), device="cuda")
    B = torch.randn((N, K) if tb else (K, N), device="cuda")
    C = torch.empty(M, N, device="cuda")
    ws = torch.empty(max(_lib.query("cal_gemm_ws", M, N, K), 4), device="cuda")
    f = lambda: _lib.call("cal_gemm", ta, tb, _p(A), _p(B), _p(C), None, 0, _p(ws), M, N, K, _stream())
    t = graph_time(f)
    t2 = graph_time(lambda: torch.matmul(A.t() if ta else A, B.t() if tb else B, out=C))
    t3 = float("nan")
    if not ta:
        C2 = torch.empty_like(C)
        f3 = lambda: _lib.call("cal_gemm_ks", tb, _p(A), _p(B), _p(C2), None, 0, M, N, K, _stream())
        t3 = graph_time(f3)
        assert torch.allclose(C2, (A @ (B.t() if tb else B)), atol=1e-3, rtol=1e-3)
    print("gemm M=%d N=%d K=%d ta=%d tb=%d: %.2f us/launch (%.1f TF)   ks %.2f us   rocBLAS %.2f us" % (M, N, K, ta, tb, t, 2*M*N*K/t/1e6, t3, t2))

def spmm(N, H, deg):
    import numpy as np
    rng = np.random.default_rng(0)
    src = rng.integers(0, N, N * deg); dst = np.repeat(np.arange(N), deg)
    ei = torch.from_numpy(np.stack([src, dst])).cuda()
    p = GraphPlan(ei, N)
    dis, norm = p.unit_norm()
    h = torch.randn(N, H, device="cuda"); out = torch.empty_like(h)
    f = lambda: _lib.call("cal_spmm_fwd", _p(p.rowptr_dst), _p(p.nbr_dst), _p(p.eid_dst), _p(norm), _p(dis), 1.0, _p(h), None, 0, _p(out), N, H, _stream())
    t = graph_time(f)
    byts = 2 * N * H * 4 + (N * deg + N) * 8 + (N + 1) * 4
    print("spmm N=%d H=%d deg=%d: %.2f us/launch  (%.0f GB/s algorithmic)" % (N, H, deg, t, byts / t / 1e3))

if __name__ == "__main__":
    x = torch.zeros(64, device="cuda")
    print("tiny torch kernel: %.2f us/launch" % graph_time(lambda: x.add_(1)))
    for shape in [(7315, 128, 128, 0, 0), (7315, 128, 128, 0, 1), (128, 128, 7315, 1, 0), (128, 128, 128, 0, 1), (128, 4, 128, 0, 1),
                  (7315, 128, 32, 0, 0), (7315, 128, 512, 0, 0), (160000, 256, 256, 0, 0)]:
        gemm(*shape)
    spmm(7315, 128, 4); spmm(160000, 256, 5)


def gat(N, H, K, deg):
    import numpy as np
    rng = np.random.default_rng(0)
    per = max(N // 32, 1)
    dst = np.repeat(np.arange(N), deg); src = (dst // per) * per + rng.integers(0, per, N * deg)
    p = GraphPlan(torch.from_numpy(np.stack([src, dst])).cuda(), N)
    D = H // K
    z = torch.randn(N, H, device="cuda"); att = torch.randn(K, 2 * D, device="cuda") * 0.2
    out = torch.empty_like(z)
    bufs = [torch.empty(N * K, device="cuda") for _ in range(4)]
    f = lambda: _lib.call("cal_gat_fwd", _p(p.rowptr_dst), _p(p.nbr_dst), _p(p.eid_dst), _p(z), _p(att), None, 1, 0.2, 0.0, 0,
                          _p(out), _p(bufs[0]), _p(bufs[1]), _p(bufs[2]), _p(bufs[3]), N, p.E, K, D, _stream())
    t = graph_time(f, n=20)
    Ep = N * deg + N
    byts = 2 * N * H * 4 + Ep * 8 + (N + 1) * 4 + 3 * Ep * K * 4
    print("gat_fwd N=%d H=%d K=%d deg=%d: %.2f us/launch  (%.0f GB/s algorithmic)" % (N, H, K, deg, t, byts / t / 1e3))
    g = torch.randn_like(z); dz = torch.empty_like(z); datt = torch.empty(K * 2 * D, device="cuda")
    ws = torch.empty(_lib.query("cal_gat_bwd_ws", N, p.E, K, D), device="cuda")
    fb = lambda: _lib.call("cal_gat_bwd", _p(p.rowptr_dst), _p(p.nbr_dst), _p(p.eid_dst), _p(p.rowptr_src), _p(p.nbr_src), _p(p.eid_src),
                           _p(z), _p(att), _p(bufs[0]), _p(bufs[1]), _p(bufs[2]), _p(bufs[3]), _p(g), 0.2, 0.0, 0, _p(dz), _p(datt), _p(ws),
                           N, p.E, K, D, _stream())
    tb = graph_time(fb, n=20)
    print("gat_bwd (5 kernels)          : %.2f us/call" % tb)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "gat":
    gat(1148, 128, 4, 2); gat(7315, 128, 4, 4); gat(160000, 256, 4, 5)
