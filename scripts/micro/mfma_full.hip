// Sustained v_mfma_f32_32x32x2_f32 rate with the whole chip busy (the ceiling a large GEMM can reach on this box):
// G workgroups of 256 threads, 4 independent accumulator chains per wave, timed with events over several hundred us.
// MODE 0: constant operands (the usual peak microbenchmark); MODE 1: operands rotate through 16 registers of random
// data (what a GEMM feeds the matrix cores: data-dependent power); MODE 2: random operands re-read from LDS every step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f; }
template <int MODE>
__global__ void __launch_bounds__(256) k_mfma(int iters, float* sink) {
    __shared__ float lds[16 * 256];
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x;
    float av[16], bv[16];
    for (int i = 0; i < 16; ++i) { av[i] = MODE ? rnd(seed) : threadIdx.x * 0.001f; bv[i] = MODE ? rnd(seed) : 1.0f; lds[i * 256 + threadIdx.x] = av[i]; }
    __syncthreads();
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float a0, a1, b0, b1;
            if (MODE == 2) {
                a0 = lds[(4 * u + 0) * 256 + threadIdx.x]; a1 = lds[(4 * u + 1) * 256 + threadIdx.x];
                b0 = lds[(4 * u + 2) * 256 + threadIdx.x]; b1 = lds[(4 * u + 3) * 256 + threadIdx.x];
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
            } else { a0 = av[4 * u]; a1 = av[4 * u + 1]; b0 = bv[4 * u + 2]; b1 = bv[4 * u + 3]; }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    if (s == 12345.f) sink[0] = s;
}
template <int MODE> void run(float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 512}) {
        const int iters = 8000;
        k_mfma<MODE><<<grid, 256>>>(iters, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) k_mfma<MODE><<<grid, 256>>>(iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = 5.0 * grid * 4 * (double)iters * 4 * 4096.0;
        printf("mode %d grid %5d: %8.1f us / launch, %6.1f TF\n", MODE, grid, ms * 1e3 / 5, flops / (ms * 1e-3) / 1e12);
    }
}
int main() {
    float* sink; (void)hipMalloc(&sink, 64);
    run<0>(sink); run<1>(sink); run<2>(sink);
    return 0;
}
