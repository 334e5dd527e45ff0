"""Launch the CSR aggregation kernel at the config-5 (HBM stress) and config-2 shapes for rocprofv3 --pmc."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cal_amd import _lib
from cal_amd.plan import GraphPlan, _p, _stream
for (N, H, deg) in [(160000, 256, 5), (7315, 128, 4)]:
    rng = np.random.default_rng(0)
    src = rng.integers(0, N, N * deg); dst = np.repeat(np.arange(N), deg)
    p = GraphPlan(torch.from_numpy(np.stack([src, dst])).cuda(), N)
    dis, norm = p.unit_norm()
    h = torch.randn(N, H, device="cuda"); out = torch.empty_like(h)
    for _ in range(10):
        _lib.call("cal_spmm_fwd", _p(p.rowptr_dst), _p(p.nbr_dst), _p(p.eid_dst), _p(norm), _p(dis), 1.0, _p(h), None, 0, _p(out), N, H, _stream())
    torch.cuda.synchronize()
    print("algorithmic bytes", N, H, 2 * N * H * 4 + (N * deg + N) * 8 + (N + 1) * 4)
