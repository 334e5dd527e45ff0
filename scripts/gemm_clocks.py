"""Phase timestamps inside k_gemm (build with CAL_HIPCC_EXTRA=-DCAL_GEMM_CLOCKS)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cal_amd import _lib
from cal_amd.plan import _p, _stream
f = _lib.lib().cal_debug_gemm_clocks
f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int
for (M, N, K, ta, tb) in [(7315, 128, 128, 0, 0), (7315, 128, 128, 0, 1), (7315, 128, 32, 0, 0), (128, 128, 7315, 1, 0)]:
    A = torch.randn((K, M) if ta else (M, K), device="cuda")
    B = torch.randn((N, K) if tb else (K, N), device="cuda")
    C = torch.empty(M, N, device="cuda")
    ws = torch.empty(max(_lib.query("cal_gemm_ws", M, N, K), 4), device="cuda")
    junk = torch.randn(64 << 20, device="cuda")
    for rep in range(3):
        junk.add_(1.0)                       # evict the operands from L2 / MALL
        A.mul_(1.0); B.mul_(1.0)             # fresh operands written by another kernel, as in the step
        _lib.call("cal_gemm", ta, tb, _p(A), _p(B), _p(C), None, 0, _p(ws), M, N, K, _stream())
        torch.cuda.synchronize()
        out = (ctypes.c_longlong * 16)()
        assert f(out) == 0
        v = list(out)
        print("M=%d N=%d K=%d ta=%d tb=%d:" % (M, N, K, ta, tb), " ".join("%.2f" % ((v[k + 1] - v[k]) / 100.0) for k in range(5)),
              "total %.2f us  [issue+tables | commit | sync | mfma | epilogue]" % ((v[5] - v[0]) / 100.0),
              " epilogue: pre %.2f stores %.2f post %.2f" % ((v[6] - v[4]) / 100.0, (v[7] - v[6]) / 100.0, (v[5] - v[7]) / 100.0))

    import numpy as np
    fb = _lib.lib().cal_debug_gemm_blocks
    fb.argtypes = [ctypes.c_void_p]; fb.restype = ctypes.c_int
    blk = (ctypes.c_longlong * 4096)()
    assert fb(blk) == 0
    nb = ((M + 63) // 64) * ((N + 63) // 64)
    t = np.array(list(blk)[:2 * nb], dtype=np.int64).reshape(nb, 2) / 100.0
    t0 = t[:, 0].min()
    print("   %d workgroups: starts spread %.2f us, durations min/med/max %.2f/%.2f/%.2f us, last end %.2f us after first start"
          % (nb, t[:, 0].max() - t0, (t[:, 1] - t[:, 0]).min(), np.median(t[:, 1] - t[:, 0]), (t[:, 1] - t[:, 0]).max(), t[:, 1].max() - t0))
