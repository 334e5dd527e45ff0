// Microbenchmark: cost of the BatchNorm-statistics reduction when 256 workgroups each add one row of `cols` fp64
// values, (a) atomically into R accumulator rows (row = workgroup % R) whose columns are `stride` doubles apart,
// (b) as plain partial rows followed by a finishing kernel.  Each variant is a dependent chain of `chain` pairs
// [producer -> consumer] captured into one hipGraph, so the per-pair time includes the kernel boundaries like the
// engine step does.  hipcc --offload-arch=gfx950 -O3 atomics64.hip -o atomics64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) k_prod_atomic(double* acc, int cols, int R, int stride, const float* x) {
    // ~2 us of dependent work stands in for the producer's body
    float v = x[threadIdx.x];
    for (int i = 0; i < 300; ++i) v = fmaf(v, 1.0001f, 0.5f);
    double* row = acc + (size_t)(blockIdx.x % R) * cols * stride;
    for (int c = threadIdx.x; c < cols; c += 256) atomicAdd(row + (size_t)c * stride, (double)v);
}
__global__ void __launch_bounds__(256) k_prod_rows(double* parts, int cols, const float* x) {
    float v = x[threadIdx.x];
    for (int i = 0; i < 300; ++i) v = fmaf(v, 1.0001f, 0.5f);
    for (int c = threadIdx.x; c < cols; c += 256) parts[(size_t)blockIdx.x * cols + c] = (double)v;
}
__global__ void __launch_bounds__(256) k_final(const double* parts, int P, int cols, double* dst) {
    // 8 columns x 32 part-lanes per block, like the engine's k_stats_final
    __shared__ double red[256];
    const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3, c = blockIdx.x * 8 + cl;
    double s = 0.0;
    if (c < cols) for (int p = pl; p < P; p += 32) s += parts[(size_t)p * cols + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0 && c < cols) { double t = 0.0; for (int k = 0; k < 32; ++k) t += red[k * 8 + cl]; dst[c] = t; }
}
// consumer: every workgroup needs all columns: reads R rows x cols (atomic variant) or the final row
__global__ void __launch_bounds__(256) k_cons(const double* acc, int cols, int R, int stride, float* out) {
    double s = 0.0;
    if (R <= 8 && stride == 1) {           // (round 5) the R reads of a column issued together, as a consumer prologue would
        for (int c = threadIdx.x; c < cols; c += 256) {
            double v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = acc[(size_t)(r < R ? r : 0) * cols + c];
#pragma unroll
            for (int r = 0; r < 8; ++r) s += r < R ? v[r] : 0.0;
        }
    } else
    for (int c = threadIdx.x; c < cols; c += 256)
        for (int r = 0; r < R; ++r) s += acc[((size_t)r * cols + c) * stride];
    float v = (float)s;
    for (int i = 0; i < 300; ++i) v = fmaf(v, 1.0001f, 0.5f);
    if (v == 12345.f) out[blockIdx.x] = v;
}
__global__ void k_zero(double* p, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = 0.0; }
template <typename F> float run_graph(F body, int chain) {
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < chain; ++i) body(st);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / 20 / chain;
}
int main() {
    const int WG = 256, chain = 40;
    double *acc, *parts; float *x, *out;
    hipMalloc(&acc, (size_t)64 * 512 * 16 * 8); hipMalloc(&parts, (size_t)WG * 512 * 8); hipMalloc(&x, 1024); hipMalloc(&out, 4096);
    hipMemset(x, 0, 1024); hipMemset(acc, 0, (size_t)64 * 512 * 16 * 8);
    for (int cols : {128, 256, 512}) {
        float base = run_graph([&](hipStream_t st) {
            k_prod_rows<<<WG, 256, 0, st>>>(parts, 0, x);
            k_cons<<<WG, 256, 0, st>>>(acc, 0, 1, 1, out); }, chain);
        printf("cols %d: producer+consumer without any reduction: %.2f us per pair\n", cols, base);
        float rows = run_graph([&](hipStream_t st) {
            k_prod_rows<<<WG, 256, 0, st>>>(parts, cols, x);
            k_final<<<(cols + 7) / 8, 256, 0, st>>>(parts, WG, cols, acc);
            k_cons<<<WG, 256, 0, st>>>(acc, cols, 1, 1, out); }, chain);
        printf("cols %d: partial rows + k_final + consumer:        %.2f us per pair (+%.2f)\n", cols, rows, rows - base);
        for (int R : {1, 2, 4, 8, 16}) for (int stride : {1}) {
            float t = run_graph([&](hipStream_t st) {
                k_prod_atomic<<<WG, 256, 0, st>>>(acc, cols, R, stride, x);
                k_cons<<<WG, 256, 0, st>>>(acc, cols, R, stride, out); }, chain);
            printf("cols %d: fp64 atomics into R=%2d rows, column stride %2d doubles: %.2f us per pair (+%.2f)\n", cols, R, stride, t, t - base);
        }
    }
    return 0;
}
