// Microbenchmark: shader clock vs wall clock, dependent global-load latency (fresh data written by
// another kernel vs warm), LDS read latency.  hipcc --offload-arch=gfx950 -O3 latency.hip -o latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_fill(int* p, int n, int stride) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (i + stride) % n;
}
__global__ void k_chase(const int* p, int steps, long long* out, int start) {
    long long w0 = wall_clock64(), c0 = clock64();
    int j = start;
    for (int s = 0; s < steps; ++s) j = __builtin_nontemporal_load(p + j);
    long long w1 = wall_clock64(), c1 = clock64();
    out[0] = w1 - w0; out[1] = c1 - c0; out[2] = j;
}
__global__ void k_chase_plain(const int* p, int steps, long long* out, int start) {
    long long w0 = wall_clock64(), c0 = clock64();
    int j = start;
    for (int s = 0; s < steps; ++s) j = p[j];
    long long w1 = wall_clock64(), c1 = clock64();
    out[0] = w1 - w0; out[1] = c1 - c0; out[2] = j;
}
__global__ void k_spin(long long* out, int iters) {
    long long w0 = wall_clock64(), c0 = clock64();
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = fmaf(x, 1.0001f, 0.5f);
    long long w1 = wall_clock64(), c1 = clock64();
    out[0] = w1 - w0; out[1] = c1 - c0; out[2] = (long long)x;
}
int main() {
    const int n = 1 << 22;   // 16 MB
    int* p; long long* out; long long h[3];
    hipMalloc(&p, n * 4); hipMalloc(&out, 64);
    for (int rep = 0; rep < 3; ++rep) {
        k_spin<<<1, 64>>>(out, 100000);
        hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
        printf("spin: 100000 dependent fma: wall %.1f us, clock64 ticks %lld  -> %.2f ns/fma\n", h[0] / 100.0, h[1], h[0] * 10.0 / 100000);
    }
    for (int stride : {1, 33, 4099, 65537}) {
        k_fill<<<n / 256, 256>>>(p, n, stride);
        k_chase_plain<<<1, 1>>>(p, 256, out, 0);
        hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
        printf("chase stride %6d (fresh, written by other CUs): %.1f ns/load\n", stride, h[0] * 10.0 / 256);
        k_chase_plain<<<1, 1>>>(p, 256, out, 0);
        hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
        printf("chase stride %6d (second pass):                 %.1f ns/load\n", stride, h[0] * 10.0 / 256);
    }
    // back-to-back in a busy loop (like a graph): 2000 tiny launches then measure
    k_fill<<<n / 256, 256>>>(p, n, 4099);
    for (int i = 0; i < 3000; ++i) k_spin<<<256, 256>>>(out + 4, 2000);
    k_chase_plain<<<1, 1>>>(p, 256, out, 0);
    hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    printf("chase after 3000 busy launches: %.1f ns/load\n", h[0] * 10.0 / 256);
    k_spin<<<1, 64>>>(out, 100000);
    hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    printf("spin after busy: %.2f ns/fma\n", h[0] * 10.0 / 100000);
    return 0;
}
