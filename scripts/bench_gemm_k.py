import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cal_amd import _lib
from cal_amd.plan import _p, _stream
for (M, N, K) in [(128, 128, 32), (128, 128, 128), (128, 128, 512), (128, 128, 2048), (7315, 128, 32), (7315, 128, 128), (7315, 128, 512)]:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); C = torch.empty(M, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    for _ in range(20):
        _lib.call("cal_gemm", 0, 0, _p(A), _p(B), _p(C), _p(bias), 1, None, M, N, K, _stream())
    torch.cuda.synchronize()
