import torch, time
x = torch.zeros(64, device="cuda")
big = torch.zeros(7315 * 128, device="cuda")
def run(n, t):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): t.add_(1)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): t.add_(1)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e6
for n in (1, 50, 200, 800):
    print("graph of %4d dependent tiny kernels: %.1f us total -> %.2f us/kernel ; 3.7MB elementwise: %.2f us/kernel" % (n, run(n, x), run(n, x) / n, run(n, big) / n))
