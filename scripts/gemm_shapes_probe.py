"""Shape sweep of the fp32 MFMA GEMM (cal_gemm -> k_gemm_big from 16k rows) next to torch.mm (rocBLAS / hipBLASLt fp32)
on the same device.  usage on the GPU box: PYTHONPATH=. python scripts/gemm_shapes_probe.py
Round 1, MI355X: [160000,256]x[256,256] 244 us (86 TF) stand-alone, 221 us inside the step with the BatchNorm prologue;
torch.mm 213 us (99 TF); [32768,1024]x[1024,1024] 99 TF vs torch.mm 122 TF."""
import torch
from cal_amd import _lib
from cal_amd.plan import _p, _stream
def t(M,N,K,tb=0,it=10):
    x = torch.randn(M, K, device="cuda"); w = torch.randn((N,K) if tb else (K, N), device="cuda") * 0.05; y = torch.empty(M, N, device="cuda")
    def run(): _lib.call("cal_gemm", 0, tb, _p(x), _p(w), _p(y), None, 0, None, M, N, K, _stream())
    run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): run()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / it * 1e3
    print("M %7d N %4d K %5d %s: %8.1f us  %6.1f TF" % (M,N,K,"NT" if tb else "NN", us, 2*M*N*K/us/1e6))
for (M,N,K) in [(160000,256,256),(160000,256,512),(160000,256,1024),(160000,128,256),(160000,512,256),(40000,256,1024),(65536,256,256),(8192*4,1024,1024)]:
    t(M,N,K)
torch.cuda.synchronize()
a=torch.randn(160000,256,device="cuda"); b=torch.randn(256,256,device="cuda")
torch.mm(a,b); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): torch.mm(a,b)
e.record(); torch.cuda.synchronize()
us=s.elapsed_time(e)/10*1e3
print("torch.mm (hipBLASLt/rocBLAS fp32) 160000x256x256: %.1f us %.1f TF" % (us, 2*160000*256*256/us/1e6))
a=torch.randn(32768,1024,device="cuda"); b=torch.randn(1024,1024,device="cuda")
torch.mm(a,b); torch.cuda.synchronize()
s.record()
for _ in range(10): torch.mm(a,b)
e.record(); torch.cuda.synchronize()
us=s.elapsed_time(e)/10*1e3
print("torch.mm 32768x1024x1024: %.1f us %.1f TF" % (us, 2*32768*1024*1024/us/1e6))
