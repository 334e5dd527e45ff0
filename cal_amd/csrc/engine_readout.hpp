// Fused readout path of the step engine (model.py:125-164 + train_causal.py:176-183 and their
// backward) for mini-batches whose pooled matrix fits in LDS (see use_ro() in engine.hip:
// roundup16(B)*(H+4) <= RO_LDS floats, B*H <= 16384, H % 16 == 0, B, H <= 256, B*C <= 2048, C <= 64):
// four kernels instead of ten launches.
//
// BatchNorm over the B pooled rows is column-local, and so is everything downstream of it until the
// next matrix product mixes columns; so a workgroup that owns a 16-column chunk (and sees all B
// rows) can do BN statistics, normalisation and their backward without any cross-workgroup
// reduction:
//   k_ro_fwd_a  grid (3 heads, H/16): x_co = xc[perm] + xo, BN1 (all columns, recomputed per
//               workgroup from the raw pooled rows in LDS), y1[:, chunk] = relu(BN1(x) W1[chunk]^T + b1),
//               BN2 batch statistics of the chunk (final, no atomics)
//   k_ro_fwd_b  grid (3, B/16): z = BN2(y1) W2^T + b2, log_softmax, per-graph loss, dz (row-local)
//   k_ro_bwd_a  grid (3, H/16): d(BN2 out) = dz W2[:, chunk], BN2 backward, ReLU mask -> dy1 chunk,
//               d b1, d gamma2/beta2, d W2[:, chunk]; chunk 0 also sums the losses and d b2
//   k_ro_bwd_b  grid (3, H/16) over INPUT columns: d(BN1 out)[:, chunk] = dy1 W1[:, chunk], BN1
//               backward -> dxin chunk, d gamma1/beta1, d W1[:, chunk]
// What shapes the code (all measured with the in-kernel clocks below, scripts/ro_clocks.py):
//  * one workgroup (4 waves) per CU, nothing else resident: every dependent round of global loads
//    costs ~1.5-2 us (first touch of a buffer from that CU), so each kernel issues ALL its tile loads
//    up front (RoBatch: up to 16 per lane per tile) and only then commits them to LDS;
//  * hipcc sinks a load guarded by `if (idx < total)` next to its guarded use, which serialises the
//    loads (13.8 us for a 64 KB tile vs 1.8 us): loads are unconditional on clamped indices and
//    pinned by an empty asm (scripts/micro/stage64k.hip); per-item divisions are avoided;
//  * LDS-operand FMA loops are LDS-bandwidth bound (a broadcast ds_read_b128 still costs the wave
//    its full LDS cycles: 5 us per 128x16x128 product), so the three products with >= 16 output
//    columns run on v_mfma_f32_16x16x4_f32 (exact f32), branch-free, operands of the next 4 k-steps
//    in flight while the current 4 are multiplied.
#pragma once
#include "engine_kernels.hpp"

namespace cal {

#ifdef CAL_RO_CLOCKS                      // profiling aid: phase timestamps (100 MHz) of one workgroup per kernel
__device__ long long g_ro_clk[64];
#define RO_CLK(k) do { if (threadIdx.x == 0 && blockIdx.x == 2 && blockIdx.y == 0) g_ro_clk[k] = wall_clock64(); } while (0)
#else
#define RO_CLK(k) do {} while (0)
#endif

constexpr int RO_CW = 16;                 // columns per workgroup
constexpr int RO_LDS = 24576;             // floats of the big LDS tile: roundup16(B) * (K + 4) (96 KB)
constexpr int RO_WLD = 260;               // row stride of a [16][K] weight chunk in LDS (K <= 256)
constexpr int RO_RB = 16;                 // graphs per workgroup in k_ro_fwd_b

struct RoHead {
    const float* W1; const float* b1; const float* W2; const float* b2;   // fc1 [H,H], fc2 [C,H]
    BNRef bn1, bn2;                       // fc1_bn (input), fc2_bn (hidden): gamma/beta/running/eps/inv_n
    double* st2_sum; double* st2_sq;      // BN2 batch statistics (final values, doubles, arena)
    double* d1_sum; double* d1_prod; double* d2_sum; double* d2_prod;     // BN backward sums -> d beta / d gamma
    double* db1; double* db2;             // [H], [C] bias gradients (arena)
    float* gW1; float* gW2;               // weight gradients (flat gradient buffer)
};
struct RoArgs {
    RoHead h[3];
    const float* pooled;                  // [2,B,H]: xc_pool, xo_pool
    const int64_t* perm; int* iperm;
    float* xco;                           // [B,H]
    float* y1;                            // [3,B,H] relu(fc1(...)) (raw, pre-BN2)
    float* zl; float* logp; float* dzl;   // [3,B,C]
    float* dy1;                           // [3,B,H]
    float* dxin;                          // [3,B,H] gradient w.r.t. the three readout inputs
    const int64_t* y;
    float* rowloss;                       // [6,B]: per-graph loss of the three heads, then their hit flags (argmax == y)
    float* stats;                         // [8]: [1+h] = loss of head h, [4] = correct_o, [5] = correct_c, [6] = correct_co (eval_acc_causal, train_causal.py:214-218)
    int B, H, C;
    float wc, wo, wco;
    int training, want_grad;
};

// scale/shift/mean/rstd of BN `bn` for column c from the column's sum / sum of squares over the B rows
__device__ __forceinline__ void ro_bn_from_sums(const BNRef& bn, int c, double s, double q, float& sc, float& sh,
                                                float& mean, float& rstd) {
    float var;
    if (bn.use_running) { mean = bn.run_mean[c]; var = bn.run_var[c]; }
    else {
        const double m = s * (double)bn.inv_n, v = q * (double)bn.inv_n - m * m;
        mean = (float)m; var = (float)(v > 0.0 ? v : 0.0);
    }
    rstd = 1.0f / sqrtf(var + bn.eps);
    sc = (bn.gamma ? bn.gamma[c] : 1.f) * rstd;
    sh = (bn.beta ? bn.beta[c] : 0.f) - mean * sc;
}

__device__ __forceinline__ void ro_pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void ro_pin(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// One batch of a [rows x cols] grid of items (rows*cols <= 256 U): lane t holds items t, t+256, ..;
// item (row, col) = flat / cols, flat % cols, walked incrementally (one division per call).
// ro_issue starts the loads (unconditional, out-of-range lanes re-read item (0,0)); ro_commit pins
// the values and hands the in-range ones to `st`.
template <class T, int U> struct RoBatch { T v[U]; };
template <int NT = 256, class T, int U, class LoadF>
__device__ __forceinline__ void ro_issue(RoBatch<T, U>& bt, int rows, int cols, LoadF ld) {      // NT = threads per workgroup
    const int total = rows * cols, q = NT / cols, r = NT % cols;
    int row = (int)threadIdx.x / cols, col = (int)threadIdx.x % cols;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool ok = (int)threadIdx.x + u * NT < total;
        bt.v[u] = ld(ok ? row : 0, ok ? col : 0);
        row += q; col += r;
        if (col >= cols) { col -= cols; ++row; }
    }
}
template <int NT = 256, class T, int U, class StoreF>
__device__ __forceinline__ void ro_commit(RoBatch<T, U>& bt, int rows, int cols, StoreF st) {
    const int total = rows * cols, q = NT / cols, r = NT % cols;
    int row = (int)threadIdx.x / cols, col = (int)threadIdx.x % cols;
#pragma unroll
    for (int u = 0; u < U; ++u) ro_pin(bt.v[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if ((int)threadIdx.x + u * NT < total) st(row, col, bt.v[u]);
        row += q; col += r;
        if (col >= cols) { col -= cols; ++row; }
    }
}

// two batches over the same item grid, committed together: st(row, col, v1, v2)
template <int NT = 256, class T, int U, class StoreF>
__device__ __forceinline__ void ro_commit2(RoBatch<T, U>& b1, RoBatch<T, U>& b2, int rows, int cols, StoreF st) {
    const int total = rows * cols, q = NT / cols, r = NT % cols;
    int row = (int)threadIdx.x / cols, col = (int)threadIdx.x % cols;
#pragma unroll
    for (int u = 0; u < U; ++u) { ro_pin(b1.v[u]); ro_pin(b2.v[u]); }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if ((int)threadIdx.x + u * NT < total) st(row, col, b1.v[u], b2.v[u]);
        row += q; col += r;
        if (col >= cols) { col -= cols; ++row; }
    }
}

__device__ __forceinline__ float dot4(const float4 a, const float4 b, float acc) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, fmaf(a.x, b.x, acc))));
}

typedef float ro_f32x4 __attribute__((ext_vector_type(4)));

// 16x16 output tiles on v_mfma_f32_16x16x4_f32 with both operands read from LDS: wave w owns tiles
// w, w+4, .. w+4(NT-1) (output rows tile*16 ..+15).  a_at(row, k) / b_at(k, col) return one operand
// element and must be safe for every row < 64 NT; lane l feeds A[row = l&15][k = l>>4],
// B[k = l>>4][col = l&15] and receives acc[t][r] = D[row = 4 (l>>4) + r][col = l&15].  Branch-free
// (a wave-uniform `if (tile < ntiles)` around the MFMA compiled to exec-mask branches with the
// accumulators bounced through VGPRs); kred % 16 == 0.
template <int NT, class AF, class BF>
__device__ __forceinline__ void ro_mfma_nt(int kred, AF a_at, BF b_at, ro_f32x4 (&acc)[4]) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
    float bv[4], av[4][NT];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bv[j] = b_at(4 * j + lk, lr);
#pragma unroll
        for (int t = 0; t < NT; ++t) av[j][t] = a_at((w + 4 * t) * 16 + lr, 4 * j + lk);
    }
    for (int k0 = 16; k0 < kred; k0 += 16) {          // 4 MFMA steps per block: one LDS wait per 4 NT MFMAs
        float bn[4], an[4][NT];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bn[j] = b_at(k0 + 4 * j + lk, lr);
#pragma unroll
            for (int t = 0; t < NT; ++t) an[j][t] = a_at((w + 4 * t) * 16 + lr, k0 + 4 * j + lk);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][t], bv[j], acc[t], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bv[j] = bn[j];
#pragma unroll
            for (int t = 0; t < NT; ++t) av[j][t] = an[j][t];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][t], bv[j], acc[t], 0, 0, 0);
}
template <class AF, class BF>
__device__ __forceinline__ void ro_mfma_tiles(int ntiles, int kred, AF a_at, BF b_at, ro_f32x4 (&acc)[4]) {
    switch ((ntiles + 3) / 4) {
        case 1: ro_mfma_nt<1>(kred, a_at, b_at, acc); break;
        case 2: ro_mfma_nt<2>(kred, a_at, b_at, acc); break;
        case 3: ro_mfma_nt<3>(kred, a_at, b_at, acc); break;
        default: ro_mfma_nt<4>(kred, a_at, b_at, acc); break;
    }
}

__global__ void __launch_bounds__(256) k_ro_fwd_a(const RoArgs a) {
    __shared__ __attribute__((aligned(16))) float Xs[RO_LDS];
    __shared__ __attribute__((aligned(16))) float Ws[RO_CW * RO_WLD];
    __shared__ float sc_s[256], sh_s[256];
    __shared__ int perm_s[256];
    __shared__ double red[8][256];
    const int hd = blockIdx.x, ch = blockIdx.y, j0 = ch * RO_CW;
    const int B = a.B, K = a.H, ld = K + 4, K4 = K / 4;
    const RoHead& h = a.h[hd];
    RO_CLK(0);
    // 1. W1 chunk and the raw input rows into LDS (co head: xo first, then + xc[perm]); while the rows
    //    pass through registers each lane sums its fixed 4-column group when 256 % K4 == 0
    const bool colfix = 256 % K4 == 0;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    auto add_stats = [&](const float4 v) {
        s[0] += (double)v.x; q[0] += (double)v.x * (double)v.x; s[1] += (double)v.y; q[1] += (double)v.y * (double)v.y;
        s[2] += (double)v.z; q[2] += (double)v.z * (double)v.z; s[3] += (double)v.w; q[3] += (double)v.w * (double)v.w;
    };
    {
        RoBatch<float4, 4> bw;
        RoBatch<float4, 16> bx;
        const float* src = a.pooled + (hd == 0 ? (size_t)0 : (size_t)B * K);
        ro_issue(bw, RO_CW, K4, [&](int j, int c) { return *reinterpret_cast<const float4*>(h.W1 + (size_t)(j0 + j) * K + c * 4); });
        ro_issue(bx, B, K4, [&](int b, int c) { return *reinterpret_cast<const float4*>(src + (size_t)b * K + c * 4); });
        const int pv = hd == 2 ? (int)a.perm[min((int)threadIdx.x, B - 1)] : 0;
        ro_commit(bw, RO_CW, K4, [&](int j, int c, const float4 v) { *reinterpret_cast<float4*>(Ws + j * RO_WLD + c * 4) = v; });
        ro_commit(bx, B, K4, [&](int b, int c, const float4 v) {
            *reinterpret_cast<float4*>(Xs + b * ld + c * 4) = v;
            if (hd < 2 && colfix) add_stats(v);
        });
        perm_s[threadIdx.x] = pv;
        __syncthreads();
        if (hd == 2) {
            ro_issue(bx, B, K4, [&](int b, int c) { return *reinterpret_cast<const float4*>(a.pooled + (size_t)perm_s[b] * K + c * 4); });
            ro_commit(bx, B, K4, [&](int b, int c, const float4 p) {
                float4* d = reinterpret_cast<float4*>(Xs + b * ld + c * 4);       // this lane wrote it above
                const float4 o = *d;
                const float4 v = make_float4(p.x + o.x, p.y + o.y, p.z + o.z, p.w + o.w);
                *d = v;
                if (colfix) add_stats(v);
                if (ch == 0) *reinterpret_cast<float4*>(a.xco + (size_t)b * K + c * 4) = v;
            });
            if (ch == 0 && (int)threadIdx.x < B) a.iperm[pv] = threadIdx.x;
            __syncthreads();
        }
    }
    RO_CLK(1);
    // 2. BN1 statistics of all K columns: lane = (4-column group cg, row part), then one lane per column
    {
        const int np = 256 / K4, cg = threadIdx.x % K4, part = threadIdx.x / K4;
        if (!colfix && part < np) {
#pragma unroll 4
            for (int b = part; b < B; b += np) add_stats(*reinterpret_cast<const float4*>(Xs + b * ld + 4 * cg));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { red[c][threadIdx.x] = s[c]; red[4 + c][threadIdx.x] = q[c]; }
        __syncthreads();
        if ((int)threadIdx.x < K) {
            const int k = threadIdx.x, g = k >> 2, c = k & 3;
            double S = 0.0, Q = 0.0;
            for (int p = 0; p < np; ++p) { S += red[c][p * K4 + g]; Q += red[4 + c][p * K4 + g]; }
            float sc, sh, mean, rstd;
            ro_bn_from_sums(h.bn1, k, S, Q, sc, sh, mean, rstd);
            sc_s[k] = sc; sh_s[k] = sh;
            if (a.training && ch == 0) {
                const double m = S * (double)h.bn1.inv_n;
                double v = Q * (double)h.bn1.inv_n - m * m;
                if (v < 0.0) v = 0.0;
                h.bn1.run_mean[k] = 0.9f * h.bn1.run_mean[k] + 0.1f * (float)m;
                h.bn1.run_var[k] = 0.9f * h.bn1.run_var[k] + 0.1f * (float)(v * (double)h.bn1.unbias);
                if (k == 0 && h.bn1.nbt) *h.bn1.nbt += 1;
            }
        }
        __syncthreads();
    }
    RO_CLK(3);
    // 3. y1[:, chunk] = relu(BN1(x) W1[chunk]^T + b1) on MFMA: rows = graphs, cols = chunk, reduction over K;
    //    BN1 is applied to the A operand as it is read (x_hat = fma(x, scale_k, shift_k))
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, j = l & 15, lk = l >> 4;
    ro_f32x4 acc[4] = {};
    const int ntiles = (B + 15) / 16;
    ro_mfma_tiles(ntiles, K, [&](int row, int k) { return fmaf(Xs[min(row, B - 1) * ld + k], sc_s[k], sh_s[k]); },
                  [&](int k, int col) { return Ws[col * RO_WLD + k]; }, acc);
    const float bias = h.b1[j0 + j];
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = (w + 4 * t) * 16 + lk * 4 + r;
            if (b < B) {
                const float v = fmaxf(acc[t][r] + bias, 0.f);
                a.y1[((size_t)hd * B + b) * K + j0 + j] = v;
                s1 += (double)v; s2 += (double)v * (double)v;
            }
        }
    RO_CLK(4);
    // 4. BN2 batch statistics of the chunk: final values, no atomics (lane t holds column t % 16)
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < RO_CW && a.training) {
        double S = 0.0, Q = 0.0;
        for (int p = 0; p < 16; ++p) { S += red[0][p * RO_CW + threadIdx.x]; Q += red[1][p * RO_CW + threadIdx.x]; }
        h.st2_sum[j0 + threadIdx.x] = S;
        h.st2_sq[j0 + threadIdx.x] = Q;
    }
    RO_CLK(5);
}

// grid (3 heads, ceil(B/16)): 16 graphs per workgroup; nothing here crosses rows (the loss sums and
// d b2 are finished by k_ro_bwd_a, or by k_ro_loss when no backward follows).
__global__ void __launch_bounds__(256) k_ro_fwd_b(const RoArgs a) {
    __shared__ __attribute__((aligned(16))) float Ys[RO_RB * RO_WLD];
    __shared__ __attribute__((aligned(16))) float W2s[64 * RO_WLD];
    __shared__ float zs[RO_RB * 64];
    __shared__ float sc_s[256], sh_s[256];
    const int hd = blockIdx.x, b0 = blockIdx.y * RO_RB;
    const int B = a.B, K = a.H, C = a.C, K4 = K / 4, nb = min(RO_RB, B - b0);
    const RoHead& h = a.h[hd];
    RO_CLK(6);
    {
        RoBatch<float4, 4> by, bws;
        RoBatch<float4, 16> bwl;
        const bool small = C * K4 <= 1024;
        auto ldw = [&](int c, int k) { return *reinterpret_cast<const float4*>(h.W2 + (size_t)c * K + k * 4); };
        auto stw = [&](int c, int k, const float4 v) { *reinterpret_cast<float4*>(W2s + c * RO_WLD + k * 4) = v; };
        ro_issue(by, nb, K4, [&](int b, int c) { return *reinterpret_cast<const float4*>(a.y1 + ((size_t)hd * B + b0 + b) * K + c * 4); });
        if (small) ro_issue(bws, C, K4, ldw); else ro_issue(bwl, C, K4, ldw);
        {   // BN2 constants (K <= 256: one column per lane), loaded unconditionally with the tiles (BNRaw, engine.hpp)
            const int k = threadIdx.x;
            BNRaw raw = bn_raw_load(h.bn2, min(k, K - 1));
            bn_raw_pin(raw);
            if (k < K) {
                bn_raw_scale_shift(h.bn2, raw, sc_s[k], sh_s[k]);
                if (a.training && blockIdx.y == 0) bn_raw_update_running(h.bn2, raw, k);
            }
        }
        if (small) ro_commit(bws, C, K4, stw); else ro_commit(bwl, C, K4, stw);
        __syncthreads();
        ro_commit(by, nb, K4, [&](int b, int c, float4 v) {
            const int c4 = c * 4;
            v.x = fmaf(v.x, sc_s[c4], sh_s[c4]); v.y = fmaf(v.y, sc_s[c4 + 1], sh_s[c4 + 1]);
            v.z = fmaf(v.z, sc_s[c4 + 2], sh_s[c4 + 2]); v.w = fmaf(v.w, sc_s[c4 + 3], sh_s[c4 + 3]);
            *reinterpret_cast<float4*>(Ys + b * RO_WLD + c4) = v;
        });
        __syncthreads();
    }
    RO_CLK(8);
    // scores: nb*C outputs; `parts` adjacent lanes share one output (a quarter / half of K each)
    {
        const int nout = RO_RB * C;
        // K % 16 == 0; the k loop below takes float4 pairs, so K4 / parts must stay even
        const int parts = nout <= 64 && K % 32 == 0 ? 4 : (nout <= 128 ? 2 : 1), kq = K4 / parts;
        const int part = threadIdx.x % parts;
        for (int o = threadIdx.x / parts; o - (int)threadIdx.x / parts < nout; o += 256 / parts) {
            const bool ok = o < nout;
            const int oc = ok ? o : 0, b = oc / C, c = oc % C;
            const float4* yr = reinterpret_cast<const float4*>(Ys + b * RO_WLD) + part * kq;
            const float4* wr = reinterpret_cast<const float4*>(W2s + c * RO_WLD) + part * kq;
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 4
            for (int k4 = 0; k4 < kq; k4 += 2) { acc0 = dot4(yr[k4], wr[k4], acc0); acc1 = dot4(yr[k4 + 1], wr[k4 + 1], acc1); }
            float z = acc0 + acc1;
            if (parts >= 2) z += __shfl_xor(z, 1);
            if (parts >= 4) z += __shfl_xor(z, 2);
            if (ok && part == 0 && b < nb) {
                z += h.b2[c];
                zs[b * C + c] = z;
                a.zl[((size_t)hd * B + b0 + b) * C + c] = z;
            }
        }
    }
    __syncthreads();
    RO_CLK(9);
    // log_softmax / per-graph loss / dz: one lane per graph
    if ((int)threadIdx.x < nb) {
        const int b = b0 + threadIdx.x;
        const float u = 1.0f / (float)C, invB = 1.0f / (float)B, logu = logf(u);
        const float wgt = hd == 0 ? a.wc : (hd == 1 ? a.wo : a.wco);
        const float* zr = zs + threadIdx.x * C;
        float m = -INFINITY;
        for (int k = 0; k < C; ++k) m = fmaxf(m, zr[k]);
        float se = 0.f;
        for (int k = 0; k < C; ++k) se += expf(zr[k] - m);
        const float lse = m + logf(se);
        const int yy = (int)a.y[b];
        int arg = 0; float best = -INFINITY;
        double lrow = hd != 0 ? (double)(-(zr[yy] - lse)) : 0.0;
        for (int k = 0; k < C; ++k) {
            const float lp = zr[k] - lse;
            a.logp[((size_t)hd * B + b) * C + k] = lp;
            if (lp > best) { best = lp; arg = k; }
            if (hd == 0) lrow += (double)(u * (logu - lp));
            if (a.want_grad) {
                const float p = expf(lp);
                a.dzl[((size_t)hd * B + b) * C + k] = wgt * invB * (hd == 0 ? (p - u) : (p - (k == yy ? 1.f : 0.f)));
            }
        }
        a.rowloss[(size_t)hd * B + b] = (float)lrow;
        a.rowloss[(size_t)(3 + hd) * B + b] = arg == yy ? 1.f : 0.f;
    }
    RO_CLK(10);
}

// per-head loss = mean of the per-graph losses, correct_o = number of hits (doubles, fixed order);
// lv / cv = this lane's graph (B <= 256), red = 512 doubles
__device__ __forceinline__ void ro_loss_sums(const RoArgs& a, int hd, float lv, float cv, double* red) {
    red[threadIdx.x] = (double)lv; red[256 + threadIdx.x] = (double)cv;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[threadIdx.x] += red[threadIdx.x + o]; red[256 + threadIdx.x] += red[256 + threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.stats[1 + hd] = (float)(red[0] / (double)a.B);
        a.stats[hd == 1 ? 4 : (hd == 0 ? 5 : 6)] = (float)red[256];
    }
    __syncthreads();
}
__device__ __forceinline__ void ro_loss_load(const RoArgs& a, int hd, float& lv, float& cv) {
    const int b = min((int)threadIdx.x, a.B - 1);
    const bool ok = (int)threadIdx.x < a.B;
    const float l = a.rowloss[(size_t)hd * a.B + b], c = a.rowloss[(size_t)(3 + hd) * a.B + b];
    lv = ok ? l : 0.f; cv = ok ? c : 0.f;
}
// forward-only steps (no k_ro_bwd_a): grid (3)
__global__ void __launch_bounds__(256) k_ro_loss(const RoArgs a) {
    __shared__ double red[512];
    float lv, cv;
    ro_loss_load(a, blockIdx.x, lv, cv);
    ro_loss_sums(a, blockIdx.x, lv, cv, red);
}

// grid (3, H/16): backward through fc2 + BN2 + ReLU for a chunk of hidden columns.
__global__ void __launch_bounds__(256) k_ro_bwd_a(const RoArgs a) {
    __shared__ float dzs[2048];                // B*C floats
    __shared__ float W2c[64 * RO_CW];
    __shared__ float dyh[256 * RO_CW];         // [B][16]
    __shared__ float yn[256 * RO_CW];
    __shared__ double red[2][256];
    __shared__ float m1s[RO_CW], m2s[RO_CW];
    const int hd = blockIdx.x, ch = blockIdx.y, j0 = ch * RO_CW;
    const int B = a.B, K = a.H, C = a.C;
    const RoHead& h = a.h[hd];
    const int j = threadIdx.x % RO_CW, rl = threadIdx.x / RO_CW;
    RO_CLK(24);
    // every global read of the kernel is issued here, before the first wait
    RoBatch<float, 8> bd;
    RoBatch<float, 4> bw;
    ro_issue(bd, 1, B * C, [&](int, int i) { return a.dzl[(size_t)hd * B * C + i]; });
    ro_issue(bw, C, RO_CW, [&](int c, int jj) { return h.W2[(size_t)c * K + j0 + jj]; });
    float yv[16];                              // this lane's y1 values (rows rl + 16 q)
#pragma unroll
    for (int q = 0; q < 16; ++q) yv[q] = a.y1[((size_t)hd * B + min(rl + 16 * q, B - 1)) * K + j0 + j];
    float mean, rstd, lv = 0.f, cv = 0.f;
    bn_mean_rstd(h.bn2, j0 + j, mean, rstd);
    const float gam = h.bn2.gamma ? h.bn2.gamma[j0 + j] : 1.f, bet = h.bn2.beta ? h.bn2.beta[j0 + j] : 0.f;
    if (ch == 0) ro_loss_load(a, hd, lv, cv);
#pragma unroll
    for (int q = 0; q < 16; ++q) ro_pin(yv[q]);
    ro_commit(bd, 1, B * C, [&](int, int i, float v) { dzs[i] = v; });
    ro_commit(bw, C, RO_CW, [&](int c, int jj, float v) { W2c[c * RO_CW + jj] = v; });
    if (ch == 0) ro_loss_sums(a, hd, lv, cv, &red[0][0]);
    __syncthreads();
    RO_CLK(25);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int b = rl + 16 * q;
        if (b < B) {
            float d = 0.f;
            for (int c = 0; c < C; ++c) d = fmaf(dzs[b * C + c], W2c[c * RO_CW + j], d);
            const float n = (yv[q] - mean) * rstd;
            dyh[b * RO_CW + j] = d;
            yn[b * RO_CW + j] = n;
            s1 += (double)d; s2 += (double)d * (double)n;
        }
    }
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    if (rl == 0) {
        for (int p = 1; p < 16; ++p) { s1 += red[0][p * RO_CW + j]; s2 += red[1][p * RO_CW + j]; }
        h.d2_sum[j0 + j] = s1; h.d2_prod[j0 + j] = s2;
        m1s[j] = (float)(s1 * (double)h.bn2.inv_n); m2s[j] = (float)(s2 * (double)h.bn2.inv_n);
    }
    __syncthreads();
    const float m1 = m1s[j], m2 = m2s[j], gs = gam * rstd;
    RO_CLK(26);
    double sb = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int b = rl + 16 * q;
        if (b < B) {
            const float n = yn[b * RO_CW + j];
            const float dy = yv[q] > 0.f ? gs * (dyh[b * RO_CW + j] - m1 - n * m2) : 0.f;     // ReLU mask
            a.dy1[((size_t)hd * B + b) * K + j0 + j] = dy;
            sb += (double)dy;
            yn[b * RO_CW + j] = fmaf(n, gam, bet);              // BN2 output (fc2 input) for d W2
        }
    }
    red[0][threadIdx.x] = sb;
    __syncthreads();
    if (rl == 0) {
        for (int p = 1; p < 16; ++p) sb += red[0][p * RO_CW + j];
        h.db1[j0 + j] = sb;
    }
    RO_CLK(27);
    if (ch == 0) {                             // d b2[c] = sum_b dz[b, c]: np row groups per class, then combine
        __syncthreads();
        const int np = 256 / C, c = threadIdx.x % C, part = threadIdx.x / C;
        double sdz = 0.0;
        if (part < np) for (int b = part; b < B; b += np) sdz += (double)dzs[b * C + c];
        red[0][threadIdx.x] = sdz;
        __syncthreads();
        if (part == 0) {
            for (int p2 = 1; p2 < np; ++p2) sdz += red[0][p2 * C + c];
            h.db2[c] = sdz;
        }
    }
    // d W2[c, chunk] = sum_b dz[b,c] * BN2out[b, chunk]: np row groups per output, combined through LDS
    const int nout = C * RO_CW;
    float* fred = dyh;                         // dyh is dead from here on (all reads precede the barrier above)
    for (int o0 = 0; o0 < nout; o0 += 256) {
        const int no = min(256, nout - o0), np = 256 / no;
        const int o = o0 + threadIdx.x % no, part = threadIdx.x / no;
        const int c = o / RO_CW, jj = o % RO_CW;
        float acc = 0.f;
        if (part < np) {
#pragma unroll 4
            for (int b = part; b < B; b += np) acc = fmaf(dzs[b * C + c], yn[b * RO_CW + jj], acc);
        }
        __syncthreads();
        fred[threadIdx.x] = acc;
        __syncthreads();
        if (part == 0) {
            for (int p = 1; p < np; ++p) acc += fred[p * no + threadIdx.x];
            h.gW2[(size_t)c * K + j0 + jj] = acc;
        }
    }
    RO_CLK(28);
}

// grid (3, H/16) over INPUT columns: backward through fc1 + BN1.
__global__ void __launch_bounds__(256) k_ro_bwd_b(const RoArgs a) {
    __shared__ __attribute__((aligned(16))) float Ds[RO_LDS];               // dy1 [roundup16(B)][K+4]
    __shared__ __attribute__((aligned(16))) float W1t[RO_CW * RO_WLD];      // W1[j][i0 + i] stored [i][j]
    __shared__ float xn[256 * RO_CW];          // input chunk [roundup16(B)][16]: raw -> normalised -> BN1 output
    __shared__ double red[2][256];
    __shared__ float m1s[RO_CW], m2s[RO_CW], mean_s[RO_CW], rstd_s[RO_CW];
    const int hd = blockIdx.x, ch = blockIdx.y, i0 = ch * RO_CW;
    const int B = a.B, K = a.H, ld = K + 4, K4 = K / 4;
    const RoHead& h = a.h[hd];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, i = l & 15, lk = l >> 4;     // i == threadIdx.x % 16
    RO_CLK(16);
    // raw input chunk (lane t always gets column t % 16) with its BN1 sums, W1[:, chunk], dy1: all loads first
    const float* xsrc = hd == 0 ? a.pooled : (hd == 1 ? a.pooled + (size_t)B * K : a.xco);
    double s = 0.0, q = 0.0;
    {
        RoBatch<float, 16> bx, bw;
        RoBatch<float4, 16> bd;
        ro_issue(bx, B, RO_CW, [&](int b, int ii) { return xsrc[(size_t)b * K + i0 + ii]; });
        ro_issue(bw, K, RO_CW, [&](int jj, int ii) { return h.W1[(size_t)jj * K + i0 + ii]; });
        ro_issue(bd, B, K4, [&](int b, int c) { return *reinterpret_cast<const float4*>(a.dy1 + ((size_t)hd * B + b) * K + c * 4); });
        ro_commit(bx, B, RO_CW, [&](int b, int ii, float v) { xn[b * RO_CW + ii] = v; s += (double)v; q += (double)v * (double)v; });
        ro_commit(bw, K, RO_CW, [&](int jj, int ii, float v) { W1t[ii * RO_WLD + jj] = v; });
        ro_commit(bd, B, K4, [&](int b, int c, const float4 v) { *reinterpret_cast<float4*>(Ds + b * ld + c * 4) = v; });
    }
    // zero rows B .. roundup16(B) of dy1 and of the input chunk (the d W1 product reduces over 16 graphs per block)
    const int B16 = (B + 15) & ~15;
    for (int idx = threadIdx.x; idx < (B16 - B) * ld; idx += 256) Ds[B * ld + idx] = 0.f;
    for (int idx = threadIdx.x; idx < (B16 - B) * RO_CW; idx += 256) xn[B * RO_CW + idx] = 0.f;
    RO_CLK(17);
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < RO_CW) {
        double S = 0.0, Q = 0.0;
        for (int p = 0; p < 16; ++p) { S += red[0][p * RO_CW + threadIdx.x]; Q += red[1][p * RO_CW + threadIdx.x]; }
        float sc, sh, mean, rstd;
        ro_bn_from_sums(h.bn1, i0 + threadIdx.x, S, Q, sc, sh, mean, rstd);
        mean_s[threadIdx.x] = mean; rstd_s[threadIdx.x] = rstd;
    }
    __syncthreads();
    const float mean = mean_s[i], rstd = rstd_s[i];
    const float gam = h.bn1.gamma ? h.bn1.gamma[i0 + i] : 1.f, bet = h.bn1.beta ? h.bn1.beta[i0 + i] : 0.f;
    RO_CLK(18);
    // d(BN1 out)[b, i] = sum_j dy1[b, j] W1[j, i] on MFMA (rows = graphs, cols = chunk, reduction over j)
    ro_f32x4 acc[4] = {};
    const int ntiles = (B + 15) / 16;
    ro_mfma_tiles(ntiles, K, [&](int row, int k) { return Ds[min(row, B - 1) * ld + k]; },
                  [&](int k, int col) { return W1t[col * RO_WLD + k]; }, acc);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = (w + 4 * t) * 16 + lk * 4 + r;
            if (b < B) {
                const float n = (xn[b * RO_CW + i] - mean) * rstd;
                xn[b * RO_CW + i] = n;                        // element (b, i) belongs to this lane in this phase
                s1 += (double)acc[t][r]; s2 += (double)acc[t][r] * (double)n;
            }
        }
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < RO_CW) {
        double S = 0.0, Q = 0.0;
        for (int p = 0; p < 16; ++p) { S += red[0][p * RO_CW + threadIdx.x]; Q += red[1][p * RO_CW + threadIdx.x]; }
        h.d1_sum[i0 + threadIdx.x] = S; h.d1_prod[i0 + threadIdx.x] = Q;
        m1s[threadIdx.x] = (float)(S * (double)h.bn1.inv_n); m2s[threadIdx.x] = (float)(Q * (double)h.bn1.inv_n);
    }
    __syncthreads();
    const float m1 = m1s[i], m2 = m2s[i], gs = gam * rstd;
    RO_CLK(19);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = (w + 4 * t) * 16 + lk * 4 + r;
            if (b < B) {
                const float n = xn[b * RO_CW + i];
                a.dxin[((size_t)hd * B + b) * K + i0 + i] = gs * (acc[t][r] - m1 - n * m2);
                xn[b * RO_CW + i] = fmaf(n, gam, bet);        // BN1 output (fc1 input) for d W1
            }
        }
    __syncthreads();
    RO_CLK(20);
    // d W1[j, chunk] = sum_b dy1[b, j] * BN1out[b, chunk] on MFMA (rows = j, cols = chunk, reduction over graphs)
    ro_f32x4 wacc[4] = {};
    ro_mfma_tiles(K / 16, B16, [&](int row, int k) { return Ds[k * ld + min(row, K - 1)]; },
                  [&](int k, int col) { return xn[k * RO_CW + col]; }, wacc);
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (w + 4 * t < K / 16) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                h.gW1[(size_t)((w + 4 * t) * 16 + lk * 4 + r) * K + i0 + i] = wacc[t][r];
        }
    RO_CLK(21);
}

}  // namespace cal
