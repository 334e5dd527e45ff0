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
// (round 5) plane = the XCD the workgroup runs on (HW_REG_XCC_ID), workgroup-scope atomics: every add to a plane comes from ONE
// XCD, so its L2 can perform it (the L2s are not coherent with each other, which is why agent-scope atomics go to memory); the
// kernel boundary writes the planes back like any other store
__global__ void __launch_bounds__(256) k_prod_atomic_xcc(double* acc, int cols, const float* x, int* xcc_seen) {
    float v = x[threadIdx.x];
    for (int i = 0; i < 300; ++i) v = fmaf(v, 1.0001f, 0.5f);
    const int xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15;          // HW_REG_XCC_ID[3:0]
    if (threadIdx.x == 0 && xcc_seen) xcc_seen[blockIdx.x] = xcc;
    double* row = acc + (size_t)(xcc & 7) * cols;
    for (int c = threadIdx.x; c < cols; c += 256) __hip_atomic_fetch_add(row + c, (double)v + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
        {
            float t = run_graph([&](hipStream_t st) {
                k_prod_atomic_xcc<<<WG, 256, 0, st>>>(acc, cols, x, nullptr);
                k_cons<<<WG, 256, 0, st>>>(acc, cols, 8, 1, out); }, chain);
            printf("cols %d: fp64 atomics, workgroup scope, plane = XCC_ID (8 rows): %.2f us per pair (+%.2f)\n", cols, t, t - base);
            // check: one launch on zeroed planes -> every column sums to WG * (v + 1) = WG (x = 0 -> v converges to a constant; compare planes' total across columns)
            int* seen; hipMalloc(&seen, WG * 4);
            hipMemset(acc, 0, (size_t)8 * cols * 8);
            k_prod_atomic_xcc<<<WG, 256>>>(acc, cols, x, seen);
            hipDeviceSynchronize();
            std::vector<double> h(8 * cols); std::vector<int> hs(WG);
            hipMemcpy(h.data(), acc, h.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(hs.data(), seen, WG * 4, hipMemcpyDeviceToHost);
            double tot0 = 0, totl = 0; int cnt[16] = {0};
            for (int r = 0; r < 8; ++r) { tot0 += h[(size_t)r * cols]; totl += h[(size_t)r * cols + cols - 1]; }
            for (int i = 0; i < WG; ++i) cnt[hs[i] & 15]++;
            printf("   check: column 0 total %.6f, last column total %.6f over 8 planes (expect equal, = %d contributions); workgroups per XCC:", tot0, totl, WG);
            for (int i = 0; i < 16; ++i) if (cnt[i]) printf(" %d:%d", i, cnt[i]);
            printf("\n");
            hipMemset(acc, 0, (size_t)64 * 512 * 16 * 8);
        }
    }
    return 0;
}
