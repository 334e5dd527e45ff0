// Microbenchmark (round 4, last session): what would the config-5 GATConv forward (PyG GATConv as used by model.py:340,390; H = 256,
// 4 heads of 64) gain if a_src / a_dst of every node came out of the producing GEMM's epilogue instead of being recomputed from
// each gathered z row?  32 BA(m = 2) graphs of 5000 nodes, no dropout.
//   F0  the product kernel's structure (gat.hip k_gat_fwd_w): wave per row, slots in lanes, a_src = <z_j, att_src> per gathered row
//       (fused DPP row sums batched over four slots), online softmax in base 2
//   F1  a_src [N, 4] / a_dst [N, 4] given: rows of <= 16 slots in the (head, slot) lane layout -- lane (k, l) owns the logit of
//       (slot l, head k): ONE 4-byte score gather, one exp per lane, max / sum as 16-lane DPP reductions -- and every gathered z row
//       is four v_fmac_f32_dpp with a row_newbcast coefficient; longer rows take F0's path
//   S   the score pass alone (z -> a_src, a_dst): what a GEMM epilogue would have to absorb
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/gat_fwd_scores.hip -o scripts/micro/gat_fwd_scores
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <utility>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int H = 256, K = 4, D = 64;
constexpr float LOG2E = 1.4426950408889634f, SLOPE = 0.2f;

struct V4 {
    float4 v;
    __device__ static V4 ld(const float* p) { V4 r; r.v = *reinterpret_cast<const float4*>(p); return r; }
    __device__ void st(float* p) const { *reinterpret_cast<float4*>(p) = v; }
    __device__ void fma(float a, const V4& x) { v.x = fmaf(a, x.v.x, v.x); v.y = fmaf(a, x.v.y, v.y); v.z = fmaf(a, x.v.z, v.z); v.w = fmaf(a, x.v.w, v.w); }
    __device__ void scale(float a) { v.x *= a; v.y *= a; v.z *= a; v.w *= a; }
    __device__ float dot(const V4& x) const { return fmaf(v.w, x.v.w, fmaf(v.z, x.v.z, fmaf(v.y, x.v.y, v.x * x.v.x))); }
    __device__ void pin() { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
};
__device__ __forceinline__ float lrelu(float x) { return x > 0.f ? x : SLOPE * x; }
__device__ __forceinline__ float rdl(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }

#define DPP(OP, r, ctl) OP " " r ", " r ", " r " " ctl " row_mask:0xf bank_mask:0xf\n"
template <int NB>
__device__ __forceinline__ void row16_sum_n(float (&v)[NB]) {
    if constexpr (NB == 8) {
        float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
        row16_sum_n<4>(a); row16_sum_n<4>(b);
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = a[u]; v[4 + u] = b[u]; }
    } else if constexpr (NB == 4) {
        asm volatile("s_nop 1\n"
                     DPP("v_add_f32_dpp", "%0", "row_ror:8") DPP("v_add_f32_dpp", "%1", "row_ror:8") DPP("v_add_f32_dpp", "%2", "row_ror:8") DPP("v_add_f32_dpp", "%3", "row_ror:8")
                     DPP("v_add_f32_dpp", "%0", "row_ror:4") DPP("v_add_f32_dpp", "%1", "row_ror:4") DPP("v_add_f32_dpp", "%2", "row_ror:4") DPP("v_add_f32_dpp", "%3", "row_ror:4")
                     DPP("v_add_f32_dpp", "%0", "row_ror:2") DPP("v_add_f32_dpp", "%1", "row_ror:2") DPP("v_add_f32_dpp", "%2", "row_ror:2") DPP("v_add_f32_dpp", "%3", "row_ror:2")
                     DPP("v_add_f32_dpp", "%0", "row_ror:1") DPP("v_add_f32_dpp", "%1", "row_ror:1") DPP("v_add_f32_dpp", "%2", "row_ror:1") DPP("v_add_f32_dpp", "%3", "row_ror:1")
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    } else {
#pragma unroll
        for (int u = 0; u < NB; ++u)
            asm volatile("s_nop 1\n" DPP("v_add_f32_dpp", "%0", "row_ror:8") "s_nop 1\n" DPP("v_add_f32_dpp", "%0", "row_ror:4") "s_nop 1\n"
                         DPP("v_add_f32_dpp", "%0", "row_ror:2") "s_nop 1\n" DPP("v_add_f32_dpp", "%0", "row_ror:1") : "+v"(v[u]));
    }
}
__device__ __forceinline__ float row16_sum1(float v) { float a[1] = {v}; row16_sum_n<1>(a); return a[0]; }
__device__ __forceinline__ float row16_max1(float v) {
    asm volatile("s_nop 1\n" DPP("v_max_f32_dpp", "%0", "row_ror:8") "s_nop 1\n" DPP("v_max_f32_dpp", "%0", "row_ror:4") "s_nop 1\n"
                 DPP("v_max_f32_dpp", "%0", "row_ror:2") "s_nop 1\n" DPP("v_max_f32_dpp", "%0", "row_ror:1") : "+v"(v));
    return v;
}
template <int U>
__device__ __forceinline__ void fma_rowbcast(V4& acc, float at, const V4& g) {
    asm("v_fmac_f32_dpp %0, %4, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %4, %6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %4, %7 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %4, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
        : "+v"(acc.v.x), "+v"(acc.v.y), "+v"(acc.v.z), "+v"(acc.v.w)
        : "v"(at), "v"(g.v.x), "v"(g.v.y), "v"(g.v.z), "v"(g.v.w), "n"(U));
}
template <int Q, int NB, int... U>
__device__ __forceinline__ void fma_rowbcast_n(V4& acc, float at, const V4 (&g)[NB], std::integer_sequence<int, U...>) {
    (fma_rowbcast<Q + U>(acc, at, g[U]), ...);
}

// ---------------------------------------------------------------- F0: the product kernel's forward (no dropout)
template <int NB>
__device__ __forceinline__ void f0_batch(V4& acc, float& m, float& lsum, const float* __restrict__ z, const V4& att_s, float ad, int jl, int q, int c) {
    V4 zv[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) zv[u] = V4::ld(z + (size_t)__builtin_amdgcn_readlane(jl, q + u) * H + c);
#pragma unroll
    for (int u = 0; u < NB; ++u) zv[u].pin();
    float e[NB], mn = m;
#pragma unroll
    for (int u = 0; u < NB; ++u) e[u] = zv[u].dot(att_s);
    row16_sum_n<NB>(e);
#pragma unroll
    for (int u = 0; u < NB; ++u) { e[u] = lrelu(ad + e[u]); mn = fmaxf(mn, e[u]); }
    const float sc = __builtin_amdgcn_exp2f(m - mn);
    lsum *= sc;
    acc.scale(sc);
#pragma unroll
    for (int u = 0; u < NB; ++u) { const float pe = __builtin_amdgcn_exp2f(e[u] - mn); lsum += pe; acc.fma(pe, zv[u]); }
    m = mn;
}
template <int NB0>
__device__ __forceinline__ void f0_row(int i, int lane, const int* __restrict__ ptr, const int* __restrict__ nbr, const float* __restrict__ z,
                                       const float* __restrict__ att, float* __restrict__ out) {
    const int c = lane * 4, k = lane >> 4, d = c & (D - 1);
    const int s0 = ptr[i], s1 = ptr[i + 1];
    const V4 zi = V4::ld(z + (size_t)i * H + c);
    const V4 att_d = V4::ld(att + k * 2 * D + d);
    V4 att_s = V4::ld(att + k * 2 * D + D + d);
    const float ad = row16_sum1(zi.dot(att_d)), as_i = row16_sum1(zi.dot(att_s));
    att_s.scale(LOG2E);
    const float ad2 = ad * LOG2E;
    float m = lrelu(ad + as_i) * LOG2E, lsum = 1.f;
    V4 acc = zi;
    for (int base = s0; base < s1; base += 64) {
        int jl = nbr[min(base + lane, s1 - 1)];
        asm volatile("" : "+v"(jl));
        const int cnt = min(64, s1 - base);
        int q = 0;
        if constexpr (NB0 == 8) {      // up to eight rows in flight: 5..7 remaining slots are ONE batch
            for (; q + 8 <= cnt; q += 8) f0_batch<8>(acc, m, lsum, z, att_s, ad2, jl, q, c);
            switch (cnt - q) { case 7: f0_batch<7>(acc, m, lsum, z, att_s, ad2, jl, q, c); q += 7; break; case 6: f0_batch<6>(acc, m, lsum, z, att_s, ad2, jl, q, c); q += 6; break;
                               case 5: f0_batch<5>(acc, m, lsum, z, att_s, ad2, jl, q, c); q += 5; break; default: break; }
        }
        for (; q + 4 <= cnt; q += 4) f0_batch<4>(acc, m, lsum, z, att_s, ad2, jl, q, c);
        switch (cnt - q) { case 3: f0_batch<3>(acc, m, lsum, z, att_s, ad2, jl, q, c); break; case 2: f0_batch<2>(acc, m, lsum, z, att_s, ad2, jl, q, c); break;
                           case 1: f0_batch<1>(acc, m, lsum, z, att_s, ad2, jl, q, c); break; default: break; }
    }
    acc.scale(1.f / (lsum + 1e-16f));
    acc.st(out + (size_t)i * H + c);
}
template <int NB0, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) k_f0(const int* __restrict__ ptr, const int* __restrict__ nbr, const float* __restrict__ z,
                                                                                    const float* __restrict__ att, float* __restrict__ out, int N) {
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    if (i >= N) return;
    f0_row<NB0>(i, lane, ptr, nbr, z, att, out);
}

// ---------------------------------------------------------------- S: scores alone
__global__ void __launch_bounds__(256) k_scores(const float* __restrict__ z, const float* __restrict__ att, float* __restrict__ adst, float* __restrict__ asrc, int N) {
    const int lane = threadIdx.x & 63, c = lane * 4, k = lane >> 4, d = c & (D - 1);
    const V4 att_d = V4::ld(att + k * 2 * D + d), att_s = V4::ld(att + k * 2 * D + D + d);
    for (int i = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); i < N; i += (int)gridDim.x * 4) {
        const V4 zi = V4::ld(z + (size_t)i * H + c);
        float e[2] = {zi.dot(att_d), zi.dot(att_s)};
        row16_sum_n<2>(e);
        if ((lane & 15) == 0) { adst[(size_t)i * K + k] = e[0]; asrc[(size_t)i * K + k] = e[1]; }
    }
}

// ---------------------------------------------------------------- F1: scores given
template <int NB1>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) k_f1(const int* __restrict__ ptr, const int* __restrict__ nbr, const float* __restrict__ z,
                                                                                    const float* __restrict__ att, const float* __restrict__ adst,
                                                                                    const float* __restrict__ asrc, float* __restrict__ out, int N) {
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    if (i >= N) return;
    const int s0 = ptr[i], s1 = ptr[i + 1], deg = s1 - s0, nsl = deg + 1;
    if (nsl > 16) { f0_row<4>(i, lane, ptr, nbr, z, att, out); return; }
    const int c = lane * 4, k = lane >> 4, l = lane & 15;
    const bool valid = l < nsl;
    int jl = i;
    if (deg > 0) { jl = nbr[s0 + min(l, deg - 1)]; asm volatile("" : "+v"(jl)); if (l >= deg) jl = i; }      // slot `deg`: the node's own loop
    const float as = asrc[(size_t)jl * K + k], ad = adst[(size_t)i * K + k];
    const float e = valid ? lrelu(ad + as) * LOG2E : -3.0e38f;
    const float m = row16_max1(e);
    const float pe = valid ? __builtin_amdgcn_exp2f(e - m) : 0.f;
    const float den = row16_sum1(pe);
    float at = pe * __builtin_amdgcn_rcpf(den + 1e-16f);
    asm volatile("s_nop 1" : "+v"(at));
    V4 acc; acc.v = make_float4(0.f, 0.f, 0.f, 0.f);
    auto batch = [&](auto qc, auto nbc) {
        constexpr int Q = decltype(qc)::value, NB = decltype(nbc)::value;
        V4 gv[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) gv[u] = V4::ld(z + (size_t)__builtin_amdgcn_readlane(jl, Q + u) * H + c);
#pragma unroll
        for (int u = 0; u < NB; ++u) gv[u].pin();
        fma_rowbcast_n<Q, NB>(acc, at, gv, std::make_integer_sequence<int, NB>());
    };
#define GS_B(Q, NB) batch(std::integral_constant<int, Q>(), std::integral_constant<int, NB>())
#define GS_REM(Q, R) switch (R) { case 3: GS_B(Q, 3); break; case 2: GS_B(Q, 2); break; case 1: GS_B(Q, 1); break; default: break; }
    if constexpr (NB1 == 8) {      // up to eight rows in flight: a row of <= 8 slots is ONE batch
        if (nsl > 8) { GS_B(0, 8); if (nsl >= 12) { GS_B(8, 4); if (nsl == 16) GS_B(12, 4); else GS_REM(12, nsl - 12) } else GS_REM(8, nsl - 8) }
        else if (nsl == 8) GS_B(0, 8);
        else if (nsl == 7) GS_B(0, 7);
        else if (nsl == 6) GS_B(0, 6);
        else if (nsl == 5) GS_B(0, 5);
        else if (nsl == 4) GS_B(0, 4);
        else GS_REM(0, nsl)
    } else
    if (nsl >= 12) { GS_B(0, 4); GS_B(4, 4); GS_B(8, 4); if (nsl == 16) GS_B(12, 4); else GS_REM(12, nsl - 12) }
    else if (nsl >= 8) { GS_B(0, 4); GS_B(4, 4); GS_REM(8, nsl - 8) }
    else if (nsl >= 4) { GS_B(0, 4); GS_REM(4, nsl - 4) }
    else GS_REM(0, nsl)
#undef GS_B
#undef GS_REM
    acc.st(out + (size_t)i * H + c);
}

// ---------------------------------------------------------------- host
struct Graphs { int N; int64_t E; std::vector<int> ptr, nbr; };
static Graphs make(int B, int n, unsigned seed) {
    std::mt19937 rng(seed);
    Graphs G; G.N = B * n;
    std::vector<std::pair<int, int>> edges;
    for (int g = 0; g < B; ++g) {
        std::vector<int> rep;
        const int m = 2, base = g * n;
        std::vector<int> targets = {0, 1};
        for (int v = m; v < n; ++v) {
            for (int t : targets) { edges.push_back({base + v, base + t}); edges.push_back({base + t, base + v}); rep.push_back(t); rep.push_back(v); }
            targets.clear();
            while ((int)targets.size() < m) {
                const int t = rep[rng() % rep.size()];
                if (std::find(targets.begin(), targets.end(), t) == targets.end()) targets.push_back(t);
            }
        }
    }
    std::shuffle(edges.begin(), edges.end(), rng);
    G.E = (int64_t)edges.size();
    G.ptr.assign(G.N + 1, 0);
    for (auto& e : edges) G.ptr[e.first + 1]++;
    for (int i = 0; i < G.N; ++i) G.ptr[i + 1] += G.ptr[i];
    G.nbr.resize(G.E);
    std::vector<int> fill(G.ptr.begin(), G.ptr.end() - 1);
    for (int64_t e = 0; e < G.E; ++e) G.nbr[fill[edges[e].first]++] = edges[e].second;
    return G;
}
template <class T> T* dev(const std::vector<T>& v) { T* p; CK(hipMalloc(&p, v.size() * sizeof(T))); CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }
static float* g_flush = nullptr;
__global__ void k_flush(float* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
static void flush_caches() {      // 1 GB of unrelated traffic between two timed launches (see gather_lds.hip)
    const size_t n = (size_t)256 << 20;
    if (!g_flush) { CK(hipMalloc(&g_flush, n * 4)); CK(hipMemset(g_flush, 0, n * 4)); }
    hipLaunchKernelGGL(k_flush, dim3((unsigned)(n / 256)), dim3(256), 0, 0, g_flush, n);
}
template <class F> float timeit(F f, int reps = 12) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) f();
    CK(hipDeviceSynchronize());
    float best = 1e9f, tot = 0.f;
    for (int i = 0; i < reps; ++i) { flush_caches(); CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms); tot += ms; }
    printf("  [%7.1f us best, %7.1f us mean]", best * 1e3f, tot / reps * 1e3f);
    return best * 1e3f;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, n = argc > 2 ? atoi(argv[2]) : 5000;
    Graphs G = make(B, n, 7);
    const int N = G.N;
    int big = 0;
    for (int i = 0; i < N; ++i) big += G.ptr[i + 1] - G.ptr[i] + 1 > 16;
    const double alg = 2.0 * N * H * 4 + (double)(G.E + N) * (8 + 12.0 * K) + (N + 1) * 4.0;        // bench.py's GAT-forward figure
    printf("B %d  n %d  N %d  E %lld  rows above 16 slots: %d   algorithmic bytes %.1f MB\n", B, n, N, (long long)G.E, big, alg * 1e-6);
    std::vector<float> hz((size_t)N * H), hatt(K * 2 * D);
    std::mt19937 rng(3);
    for (auto& v : hz) v = (int)(rng() % 2001 - 1000) * 0.5e-3f;
    for (auto& v : hatt) v = (int)(rng() % 2001 - 1000) * 0.3e-3f;
    int *d_ptr = dev(G.ptr), *d_nbr = dev(G.nbr);
    float *d_z = dev(hz), *d_att = dev(hatt), *d_o0, *d_o1, *d_ad, *d_as;
    CK(hipMalloc(&d_o0, (size_t)N * H * 4)); CK(hipMalloc(&d_o1, (size_t)N * H * 4)); CK(hipMalloc(&d_ad, (size_t)N * K * 4)); CK(hipMalloc(&d_as, (size_t)N * K * 4));
    auto report = [&](const char* name, float us) { printf("  %-58s %6.1f us  %4.1f %% of 8 TB/s on the forward's algorithmic bytes\n", name, us, alg / us * 1e-3 / 80.0); };
    float us;
    us = timeit([&] { hipLaunchKernelGGL((k_f0<8, 6>), dim3((N + 3) / 4), dim3(256), 0, 0, d_ptr, d_nbr, d_z, d_att, d_o0, N); });
    report("F0 with eight rows in flight at six waves per SIMD", us);
    us = timeit([&] { hipLaunchKernelGGL((k_f0<4, 8>), dim3((N + 3) / 4), dim3(256), 0, 0, d_ptr, d_nbr, d_z, d_att, d_o0, N); });
    report("F0 product structure (a_src from the gathered rows)", us);
    us = timeit([&] { hipLaunchKernelGGL(k_scores, dim3(2048), dim3(256), 0, 0, d_z, d_att, d_ad, d_as, N); });
    report("S  score pass alone (164 MB read)", us);
    us = timeit([&] { hipLaunchKernelGGL((k_f1<8>), dim3((N + 3) / 4), dim3(256), 0, 0, d_ptr, d_nbr, d_z, d_att, d_ad, d_as, d_o1, N); });
    report("F1 with up to eight rows in flight", us);
    us = timeit([&] { hipLaunchKernelGGL((k_f1<4>), dim3((N + 3) / 4), dim3(256), 0, 0, d_ptr, d_nbr, d_z, d_att, d_ad, d_as, d_o1, N); });
    report("F1 scores given, (head, slot) lanes + row_newbcast", us);
    CK(hipDeviceSynchronize());
    std::vector<float> a((size_t)N * H), b((size_t)N * H);
    CK(hipMemcpy(a.data(), d_o0, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_o1, b.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0, ref = 0.0;
    for (size_t t = 0; t < a.size(); ++t) { worst = std::max(worst, (double)std::fabs(a[t] - b[t])); ref = std::max(ref, (double)std::fabs(a[t])); }
    // CPU reference of a few rows against F0
    double cw = 0.0;
    for (int i = 0; i < N; i += 4001) {
        for (int k = 0; k < K; ++k) {
            auto sc = [&](int j, int off) { double s = 0; for (int d = 0; d < D; ++d) s += (double)hz[(size_t)j * H + k * D + d] * hatt[k * 2 * D + off + d]; return s; };
            std::vector<int> js(G.nbr.begin() + G.ptr[i], G.nbr.begin() + G.ptr[i + 1]); js.push_back(i);
            std::vector<double> e;
            double mx = -1e300;
            for (int j : js) { double x = sc(i, 0) + sc(j, D); x = x > 0 ? x : SLOPE * x; e.push_back(x); mx = std::max(mx, x); }
            double den = 0; for (auto& x : e) { x = std::exp(x - mx); den += x; }
            for (int d = 0; d < D; ++d) {
                double o = 0; for (size_t t = 0; t < js.size(); ++t) o += e[t] / den * hz[(size_t)js[t] * H + k * D + d];
                cw = std::max(cw, std::fabs(o - (double)a[(size_t)i * H + k * D + d]));
            }
        }
    }
    printf("  max |F1 - F0| %.2e (max |out| %.2f)   max |F0 - fp64 reference| on sampled rows %.2e   %s\n", worst, ref, cw, worst < 1e-4 && cw < 1e-4 ? "ok" : "MISMATCH");
    return 0;
}
