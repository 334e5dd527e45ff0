// Microbenchmark: wall time per v_mfma_f32_32x32x2_f32 with 1/2/4 independent accumulator chains,
// one wave per SIMD on one CU, measured (a) cold after an idle gap, (b) right after ~0.5 s of
// sustained work on all CUs -- i.e. what engine clock a short kernel actually sees.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ void __launch_bounds__(256) k_mfma(int iters, long long* out, float* sink) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f;
    long long w0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    long long w1 = wall_clock64(), c1 = clock64();
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; }
    if (s == 12345.f) sink[0] = s;
}
__global__ void k_busy(float* p, int iters) {
    float x = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) x = fmaf(x, 1.0001f, 0.5f);
    p[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <int CH> void run(const char* tag, long long* out, float* sink, int iters) {
    long long h[2];
    k_mfma<CH><<<1, 256>>>(iters, out, sink);
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("%-22s chains %d: %.1f ns / MFMA   (clock64 ticks per MFMA %.1f)\n", tag, CH, h[0] * 10.0 / (iters * CH), (double)h[1] / (iters * CH));
}
int main() {
    long long* out; float* sink; float* buf;
    hipMalloc(&out, 64); hipMalloc(&sink, 64); hipMalloc(&buf, 1024 * 256 * 4);
    hipMemset(buf, 0, 1024 * 256 * 4);
    for (int iters : {64, 2000}) {
        sleep(1);
        run<1>(iters == 64 ? "cold, 64 iters" : "cold, 2000 iters", out, sink, iters);
        sleep(1);
        run<2>(iters == 64 ? "cold, 64 iters" : "cold, 2000 iters", out, sink, iters);
        sleep(1);
        run<4>(iters == 64 ? "cold, 64 iters" : "cold, 2000 iters", out, sink, iters);
    }
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 200; ++i) k_busy<<<1024, 256>>>(buf, 20000);     // sustained load
        run<2>("after sustained load", out, sink, 64);
    }
    // many tiny kernels back to back (like a graph of small kernels), then measure
    for (int i = 0; i < 20000; ++i) k_busy<<<256, 256>>>(buf, 50);
    run<2>("after 20000 tiny kernels", out, sink, 64);
    for (int i = 0; i < 20000; ++i) k_busy<<<256, 256>>>(buf, 50);
    run<4>("after 20000 tiny kernels", out, sink, 64);
    return 0;
}
