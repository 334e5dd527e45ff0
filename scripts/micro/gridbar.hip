// Microbenchmark behind the round-3 decision on the BatchNorm seams of the headline step:
// what does one "per-graph phase -> cross-graph column statistics -> next per-graph phase" seam cost
//   V0  as three launches   phase | k_final (sums the partial rows) | phase          (the round-2 step)
//   V1  as two launches     phase | phase, the CONSUMER sums the partial rows itself  (round 3)
//   V2  inside one persistent launch, single-counter grid barrier, consumer-side sums
//   V3  inside one persistent launch, XCD-hierarchical grid barrier, consumer-side sums
// Geometry of k_gconv_fwd/bwd at config 2: 256 workgroups (128 graphs x 2 column slices) of 512 threads, ~90 KB of
// LDS (one per CU); a phase = ~WORK dependent FMAs (stand-in for staging + MFMA), a 64x64 fp32 output tile (16 KB), one
// partial row of 2 x 64 fp64 column sums; the next phase reads its own and its sibling's tile and needs the totals of all
// 128 columns.   hipcc --offload-arch=gfx950 -O3 gridbar.hip -o gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NWG = 256, NT = 512, COLS = 128, NG = 128, LDS_BYTES = 90 * 1024;

struct Bufs {
    float* tiles;      // [2 parity][NWG][4096]
    double* parts;     // [2 parity][2 stats][NG][COLS]
    double* totals;    // [2][COLS]
    float* sink;
    int* ctr;          // [64] barrier words
    unsigned long long* gran;   // [chain][2 * COLS] data-tagged granules {float value, uint tag}
};

__device__ __forceinline__ float body(float v, int work) {
    for (int i = 0; i < work; ++i) v = fmaf(v, 1.0001f, 0.5f);
    return v;
}

// producer half of a phase: the tile and the partial row of this (graph, slice)
__device__ __forceinline__ void produce(const Bufs& b, int par, int wg, float v) {
    const int g = wg >> 1, s = wg & 1, t = threadIdx.x;
    float4* tp = reinterpret_cast<float4*>(b.tiles + ((size_t)par * NWG + wg) * 4096);
    tp[t] = make_float4(v, v, v, v);
    tp[t + NT] = make_float4(v, v, v, v);
    if (t < 128) b.parts[(((size_t)par * 2 + (t >> 6)) * NG + g) * COLS + s * 64 + (t & 63)] = (double)v;
}
// consumer half: own + sibling tile (32 KB), and the statistics: totals (final == true) or the partial rows
template <int MODE>
__device__ __forceinline__ float consume(const Bufs& b, int par, int wg, bool final_, float* lds) {
    const int t = threadIdx.x;
    const float4* t0 = reinterpret_cast<const float4*>(b.tiles + ((size_t)par * NWG + (wg & ~1)) * 4096);
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = t0[t + u * NT];
    double s = 0.0;
    if (final_) {
        if (t < 2 * COLS) s = b.totals[t];
    } else {
        // 512 threads: (stat, column) = t & 255, half of the 128 rows each, 16 loads in flight
        const int sc = t & 255, hf = t >> 8;
        const double* p = b.parts + (((size_t)par * 2 + (sc >> 7)) * NG + hf * 64) * COLS + (sc & 127);
        if (MODE == 1) {
            for (int r0 = 0; r0 < 64; r0 += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = p[(size_t)(r0 + u) * COLS];
#pragma unroll
                for (int u = 0; u < 16; ++u) s += v[u];
            }
        } else if (MODE == 2) {                            // all 64 loads of the lane in flight at once
            double v[64];
#pragma unroll
            for (int u = 0; u < 64; ++u) v[u] = p[(size_t)u * COLS];
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
            for (int u = 0; u < 64; ++u) s += v[u];
        } else {                                           // fp32 partial rows (same buffer read as floats: half the bytes)
            const float* pf = reinterpret_cast<const float*>(b.parts) + (((size_t)par * 2 + (sc >> 7)) * NG + hf * 64) * COLS + (sc & 127);
            float v[64];
#pragma unroll
            for (int u = 0; u < 64; ++u) v[u] = pf[(size_t)u * COLS];
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
            for (int u = 0; u < 64; ++u) s += (double)v[u];
        }
        reinterpret_cast<double*>(lds)[t] = s;
        __syncthreads();
        if (t < 256) s = reinterpret_cast<double*>(lds)[t] + reinterpret_cast<double*>(lds)[t + 256];
        __syncthreads();
    }
    float v = (float)s * 1e-30f;
#pragma unroll
    for (int u = 0; u < 4; ++u) v += (a[u].x + a[u].y) * 1e-30f;
    return v;
}

template <int MODE>
__global__ void __launch_bounds__(NT) k_phase(Bufs b, int par, int work, int final_, int first) {
    extern __shared__ float lds[];
    float v = 1.f;
    if (!first) v += consume<MODE>(b, par ^ 1, blockIdx.x, final_ != 0, lds);
    v = body(v, work);
    produce(b, par, blockIdx.x, v);
    if (v == 12345.f) b.sink[0] = v;
}
__global__ void __launch_bounds__(256) k_final(Bufs b, int par) {
    // 8 columns x 32 part-lanes per block like k_stats_final; grid (COLS / 8, 2 stats)
    __shared__ double red[256];
    const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3, c = blockIdx.x * 8 + cl, st = blockIdx.y;
    const double* p = b.parts + ((size_t)par * 2 + st) * NG * COLS;
    double s = 0.0;
    for (int q = pl; q < NG; q += 32) s += p[(size_t)q * COLS + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0) { double tt = 0.0; for (int k = 0; k < 32; ++k) tt += red[k * 8 + cl]; b.totals[st * COLS + c] = tt; }
}


// ---- V4: the finishing blocks ride at the FRONT of the consumer's grid -----------------------------------------------
// R reducer blocks (256 threads) sum the partial rows and publish every total as one 8-byte {float, tag} granule with a
// single agent-scope store; the consumers issue everything that does not need the totals first (their tiles), then poll
// their granule (one lane per (statistic, column)).  No fence anywhere: the payload and its tag arrive together.
constexpr int RB = 8;
__global__ void __launch_bounds__(NT) k_phase_r(Bufs b, int par, int work, int first, int slot) {
    extern __shared__ float lds[];
    unsigned long long* gran = b.gran + (size_t)slot * 2 * COLS;
    const int t = threadIdx.x;
    if (blockIdx.x < RB) {
        if (first || t >= 256) return;
        // 32 (stat, column) pairs per block x 8 row lanes, 16 rows each in flight
        const int pr = blockIdx.x * 32 + (t & 31), rl = t >> 5;
        const double* p = b.parts + (((size_t)(par ^ 1) * 2 + (pr >> 7)) * NG + rl * 16) * COLS + (pr & 127);
        double v[16], s = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(size_t)u * COLS];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
        double* red = reinterpret_cast<double*>(lds);
        red[t] = s;
        __syncthreads();
        if (rl == 0) {
            double tot = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) tot += red[k * 32 + (t & 31)];
            b.totals[pr] = tot;
            const unsigned long long g = ((unsigned long long)1u << 32) | __float_as_uint((float)tot);
            __hip_atomic_store(gran + pr, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const int wg = blockIdx.x - RB;
    float v = 1.f;
    if (!first) {
        const float4* t0 = reinterpret_cast<const float4*>(b.tiles + ((size_t)(par ^ 1) * NWG + (wg & ~1)) * 4096);
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = t0[t + u * NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(a[u].x), "+v"(a[u].y));
        float tv = 0.f;
        if (t < 2 * COLS) {
            unsigned long long g;
            int spins = 0;
            do {
                g = __hip_atomic_load(gran + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(g >> 32) == 1u) break;
                __builtin_amdgcn_s_sleep(1);
            } while (++spins < (1 << 22));
            tv = __uint_as_float((unsigned)g);
        }
        lds[t] = tv;
        __syncthreads();
        v += lds[t & 255] * 1e-30f;
#pragma unroll
        for (int u = 0; u < 4; ++u) v += (a[u].x + a[u].y) * 1e-30f;
    }
    v = body(v, work);
    produce(b, par, wg, v);
    if (v == 12345.f) b.sink[0] = v;
}
__global__ void k_zero_gran(unsigned long long* g, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) g[i] = 0ull; }

// ---- grid barriers ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ld_relaxed(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// FENCE_ALL: every wave issues the release / acquire fences (what k_ro_step does); else lane 0 only, around the
// workgroup barriers (the L2 write-back / invalidate are cache-wide operations: one wave's covers the workgroup)
template <bool FENCE_ALL>
__device__ __forceinline__ void bar_counter(int* ctr, int target) {
    if (FENCE_ALL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (ld_relaxed(ctr) < target) __builtin_amdgcn_s_sleep(1);
        if (!FENCE_ALL) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (FENCE_ALL) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// ctr[0..7]: arrivals per XCC, ctr[8]: XCC leaders arrived, ctr[16..23]: generation per XCC; nx = XCCs in use,
// nper = workgroups on this XCC (both learnt in the prologue); gen counts from 1
__device__ __forceinline__ void bar_xcd(int* ctr, int xcc, int nper, int nx, int gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int old = __hip_atomic_fetch_add(ctr + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == nper * gen - 1) {                        // last arriver of this XCC: the leader
            __hip_atomic_fetch_add(ctr + 8, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_relaxed(ctr + 8) < nx * gen) __builtin_amdgcn_s_sleep(1);
            __hip_atomic_store(ctr + 16 + xcc, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_relaxed(ctr + 16 + xcc) < gen) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__global__ void __launch_bounds__(NT) k_persist(Bufs b, int phases, int work, int xcd, long long* clk) {
    extern __shared__ float lds[];
    __shared__ int info[3];
    int* ctr = b.ctr;
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    // prologue: census of workgroups per XCC (words 32..39), then one plain barrier (word 40)
    if (threadIdx.x == 0) atomicAdd(ctr + 32 + xcc, 1);
    bar_counter<true>(ctr + 40, NWG);
    if (threadIdx.x == 0) {
        int nx = 0;
        for (int i = 0; i < 8; ++i) nx += ld_relaxed(ctr + 32 + i) > 0;
        info[0] = ld_relaxed(ctr + 32 + xcc); info[1] = nx;
    }
    __syncthreads();
    const int nper = info[0], nx = info[1];
    const long long t0 = wall_clock64();
    float v = 1.f;
    for (int ph = 0; ph < phases; ++ph) {
        if (ph) v += consume<2>(b, (ph & 1) ^ 1, blockIdx.x, false, lds);
        v = body(v, work);
        produce(b, ph & 1, blockIdx.x, v);
        if (xcd == 2) bar_xcd(ctr, xcc, nper, nx, ph + 1);
        else if (xcd == 1) bar_counter<false>(ctr + 41, NWG * (ph + 1));
        else bar_counter<true>(ctr + 41, NWG * (ph + 1));
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = wall_clock64() - t0;
    if (v == 12345.f) b.sink[0] = v;
}

template <typename F> float run_graph(F body_, int chain) {
    hipStream_t st; (void)hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < chain; ++i) body_(st, i);
    (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < 20; ++i) (void)hipGraphLaunch(ge, st);
    (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / 20 / chain;
}

int main() {
    Bufs b;
    (void)hipMalloc(&b.tiles, (size_t)2 * NWG * 4096 * 4); (void)hipMalloc(&b.parts, (size_t)2 * 2 * NG * COLS * 8);
    (void)hipMalloc(&b.totals, 2 * COLS * 8); (void)hipMalloc(&b.sink, 64); (void)hipMalloc(&b.ctr, 64 * 4);
    (void)hipMemset(b.tiles, 0, (size_t)2 * NWG * 4096 * 4); (void)hipMemset(b.parts, 0, (size_t)2 * 2 * NG * COLS * 8);
    long long* clk; (void)hipMalloc(&clk, 8);
    (void)hipMalloc(&b.gran, (size_t)64 * 2 * COLS * 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_phase_r), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_phase<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_phase<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_phase<3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_persist), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int chain = 40;
    for (int work : {0, 600, 2400}) {
        const float solo = run_graph([&](hipStream_t st, int i) {
            hipLaunchKernelGGL(k_phase<1>, dim3(NWG), dim3(NT), LDS_BYTES, st, b, i & 1, work, 1, 1); }, chain);
        const float v0 = run_graph([&](hipStream_t st, int i) {
            hipLaunchKernelGGL(k_phase<1>, dim3(NWG), dim3(NT), LDS_BYTES, st, b, i & 1, work, 1, i == 0);
            hipLaunchKernelGGL(k_final, dim3(COLS / 8, 2), dim3(256), 0, st, b, i & 1); }, chain);
        const float v1 = run_graph([&](hipStream_t st, int i) {
            hipLaunchKernelGGL(k_phase<1>, dim3(NWG), dim3(NT), LDS_BYTES, st, b, i & 1, work, 0, i == 0); }, chain);
        const float v1b = run_graph([&](hipStream_t st, int i) {
            hipLaunchKernelGGL(k_phase<2>, dim3(NWG), dim3(NT), LDS_BYTES, st, b, i & 1, work, 0, i == 0); }, chain);
        const float v1c = run_graph([&](hipStream_t st, int i) {
            hipLaunchKernelGGL(k_phase<3>, dim3(NWG), dim3(NT), LDS_BYTES, st, b, i & 1, work, 0, i == 0); }, chain);
        const float v4 = run_graph([&](hipStream_t st, int i) {
            if (i == 0) hipLaunchKernelGGL(k_zero_gran, dim3(64 * 2 * COLS / 256), dim3(256), 0, st, b.gran, 64 * 2 * COLS);
            hipLaunchKernelGGL(k_phase_r, dim3(NWG + RB), dim3(NT), LDS_BYTES, st, b, i & 1, work, i == 0, i); }, chain);
        float vp[3];
        for (int xcd = 0; xcd < 3; ++xcd) {
            hipStream_t st; (void)hipStreamCreate(&st);
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                (void)hipMemsetAsync(b.ctr, 0, 64 * 4, st);
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0, st);
                hipLaunchKernelGGL(k_persist, dim3(NWG), dim3(NT), LDS_BYTES, st, b, chain * 4, work, xcd, clk);
                (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                long long c; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
                const float us = (float)c / 100.f / (chain * 4);     // wall_clock64: 100 MHz
                if (us < best) best = us;
                (void)ms;
            }
            vp[xcd] = best;
        }
        printf("work %4d: phase alone %.2f | V0 phase+k_final+phase %.2f | V1 consumer-side sums 4x16 loads %.2f, 64 loads %.2f, fp32 rows %.2f | "
               "V4 finishing blocks inside the consumer grid + granules %.2f | persistent (64-load sums): counter barrier, all waves fence %.2f; lane-0 fences %.2f; XCD barrier %.2f   (us per phase)\n",
               work, solo, v0, v1, v1b, v1c, v4, vp[0], vp[1], vp[2]);
    }
    return 0;
}
