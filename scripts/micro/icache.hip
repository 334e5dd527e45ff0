// Microbenchmark: cost of a cold instruction cache.  k_long executes N straight-line (unrolled, all
// distinct) FMA instructions once; its per-workgroup duration is measured right after a different large
// kernel has run (cold) and again immediately after itself (warm).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N, int SALT>
__global__ void __launch_bounds__(256) k_long(float* p, long long* out) {
    long long w0 = wall_clock64();
    float x = p[threadIdx.x], y = x * 0.5f, z = x + 1.f, u = x - 1.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        // distinct constants -> distinct instruction words (nothing for the compiler to roll up)
        x = fmaf(x, 1.0f + (i + SALT) * 1e-6f, y);
        y = fmaf(y, 0.999f + (i + SALT) * 1e-6f, z);
        z = fmaf(z, 1.001f - (i + SALT) * 1e-6f, u);
        u = fmaf(u, 0.998f + (i + SALT) * 2e-6f, x);
    }
    long long w1 = wall_clock64();
    p[threadIdx.x + 256 * blockIdx.x] = x + y + z + u;
    if (threadIdx.x == 0) out[blockIdx.x] = w1 - w0;
}
template <int N> void run(float* p, long long* out) {
    long long h[256];
    auto stat = [&](const char* tag) {
        hipMemcpy(h, out, 8 * 256, hipMemcpyDeviceToHost);
        double mx = 0, sum = 0; for (int i = 0; i < 256; ++i) { sum += h[i]; if (h[i] > mx) mx = h[i]; }
        printf("  %-34s mean %.2f us  max %.2f us\n", tag, sum / 256 / 100.0, mx / 100.0);
    };
    printf("%d FMAs x4 (~%d KB of code):\n", N, N * 4 * 8 / 1024);
    for (int rep = 0; rep < 2; ++rep) {
        k_long<N, 1><<<256, 256>>>(p, out);            // a different kernel of the same size: evicts
        k_long<N, 0><<<256, 256>>>(p, out);
        stat("after another large kernel (cold)");
        k_long<N, 0><<<256, 256>>>(p, out);
        k_long<N, 0><<<256, 256>>>(p, out);
        stat("third launch in a row (warm?)");
    }
}
int main() {
    float* p; long long* out;
    hipMalloc(&p, 256 * 256 * 4 + 1024); hipMalloc(&out, 8 * 256);
    hipMemset(p, 0, 256 * 256 * 4 + 1024);
    run<64>(p, out);
    run<512>(p, out);
    run<2048>(p, out);
    return 0;
}
