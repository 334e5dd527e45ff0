// Per-graph fused GATConv layer, forward (CausalGAT backbone, model.py:340,390 behind model.py:388-390):
//
//     out = relu(softmax-attention aggregation of z + b),   z = BN(x) W,   for ONE graph x 64 output columns
//
// Same skeleton as k_gconv_fwd (engine_gconv.hpp): every global load issued up front, x staged k-major with the
// BatchNorm applied, z = x'W on the 32x32x2 f32 MFMA, z kept in LDS.  Instead of the normalised adjacency block the
// workgroup then builds, for each attention head whose columns lie in its 64-column slice (64 / D heads, D = 32 or
// 64), the dense attention block alpha_h[j][i] of the graph: per (node, head) scores a_dst / a_src from the z tile,
// the edge softmax over the node's incoming CSR slots and its self loop (max and denominator saved for the
// backward), attention dropout from the same counter-based mask as k_gat_fwd -- and aggregates with a second MFMA
// product out[:, head h] = alpha_h z[:, head h].  z, a_dst, a_src, max, denominator go to HBM for the (unfused)
// backward; neither the gather of neighbour rows nor k_gat_scores / k_gat_fwd / k_colstats launches remain.
// Needs the per-graph bounds from the host like k_gconv_fwd: <= 64 nodes and <= 1024 stored edges per graph.
#pragma once
#include "engine_gconv.hpp"
#include "gat_common.hpp"

namespace cal {

constexpr int GG_T = 64;                       // nodes per graph
constexpr int GG_E = 1024;                     // stored edges per graph

struct GgatArgs {
    const float* x;          // [N,K] layer input (raw)
    const float* W;          // [K,H]
    const float* bias;       // [H]
    const float* att;        // [heads, 2 D]: target half, source half
    BNRef bn;                // BatchNorm applied to x
    float* out;              // [N,H]
    float* z;                // [N,H] BN(x) W (kept for the backward)
    float* adst;             // [N,heads] each, kept for the backward
    float* asrc;
    float* mx;
    float* den;
    Acc st_sum, st_sq;       // column statistics of out (one partial row per graph), or off
    int heads, D;
    float slope, p;
    uint64_t seed;
    const uint64_t* ctr;     // device step counter folded into the seed, or null
    int64_t E;               // input edges of the batch (self-loop mask ids start at E)
};

__device__ __forceinline__ float gg_lrelu(float v, float slope) { return v > 0.f ? v : slope * v; }

__global__ void __launch_bounds__(256) k_ggat_fwd(const CSR g, const int* __restrict__ gptr, const int* __restrict__ eptr,
                                                  const GgatArgs a, int H, int K, int* __restrict__ status) {
    constexpr int T = GG_T, LDA = T + 1;
    __shared__ __attribute__((aligned(16))) float As[GC_K * LDA];          // x stage [k][row]; later two attention blocks [j][i]
    __shared__ __attribute__((aligned(16))) float Bs[GC_K * GC_LDB];       // W slice [k][col]; later the z tile [row][col]
    __shared__ float sc_s[GC_K], sh_s[GC_K];
    __shared__ int ptr_s[T + 4];
    __shared__ signed char en[GG_E];             // source node of a slot, local to the graph (-1: edge leaves the graph)
    __shared__ int ee[GG_E];
    __shared__ float att_s[2 * GC_N];          // [head in slice][2 D]
    __shared__ float ad_s[2][T], as_s[2][T], idn_s[2][T];
    __shared__ double red[4][2][32];
    const int b = blockIdx.x, n0 = blockIdx.y * GC_N, t = threadIdx.x;
    const int g0 = gptr[b], rows = gptr[b + 1] - g0, e0 = eptr[b], ne = eptr[b + 1] - e0;
    const bool want = a.st_sum.on();
    if (rows <= 0) {
        if (a.bn.update && blockIdx.x == 0 && blockIdx.y == 0 && t < K) bn_update_running(a.bn, t);
        if (t < GC_N && want) { a.st_sum.add(n0 + t, 0.0); a.st_sq.add(n0 + t, 0.0); }
        return;
    }
    if (rows > T || ne > GG_E || ne < 0) {
        if (t == 0) atomicOr(status, 8);
        return;
    }
    const int D = a.D, hs = GC_N / D, h0 = n0 / D;        // heads of this slice: h0 .. h0 + hs - 1
    const int rowsP = (rows + 31) & ~31, R = rowsP >> 5, nkc = K >> 5;
    // ---- every global load of the kernel, issued before the first wait -------------------------------------
    constexpr int UA = T / 8;
    float4 va[UA];
    {
        int kc = 0, rr = 0;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const bool ok = kc < nkc;
            const int r = min(((ok ? rr : 0) << 5) + (t >> 3), rows - 1), k = ((ok ? kc : 0) << 5) + ((t & 7) << 2);
            va[u] = *reinterpret_cast<const float4*>(a.x + (size_t)(g0 + r) * K + k);
            if (++rr == R) { rr = 0; ++kc; }
        }
    }
    float4 vb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx = t + u * 256, k = min(idx >> 4, K - 1), j4 = idx & 15;
        vb[u] = *reinterpret_cast<const float4*>(a.W + (size_t)k * H + n0 + 4 * j4);
    }
    const int pv = g.ptr[g0 + min(t, rows)];
    int nv[4], ev[4];
    if (ne > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = e0 + min(t + u * 256, ne - 1);
            nv[u] = g.nbr[s];
            ev[u] = g.eid[s];
        }
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) { nv[u] = g0; ev[u] = 0; }
    }
    const float attv = t < 2 * GC_N ? a.att[(size_t)h0 * 2 * D + t] : 0.f;       // hs heads x 2 D = 128 floats, contiguous
    const int lane = t & 63, li = lane & 31, lk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ct = w & 1, r0 = w >> 1;
    const float bias = a.bias ? a.bias[n0 + ct * 32 + li] : 0.f;
    if (t < K) {
        bn_scale_shift(a.bn, t, sc_s[t], sh_s[t]);
        if (a.bn.update && blockIdx.x == 0 && blockIdx.y == 0) bn_update_running(a.bn, t);
    }
#pragma unroll
    for (int u = 0; u < UA; ++u) ro_pin(va[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) ro_pin(vb[u]);
    // ---- stage everything in LDS ---------------------------------------------------------------------------
    if (t <= rows) ptr_s[t] = pv - e0;
    if (t < 2 * GC_N) att_s[t] = attv;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int s = t + u * 256;
        if (s < ne) {
            const int loc = nv[u] - g0;
            const bool inb = loc >= 0 && loc < rows;
            en[s] = (signed char)(inb ? loc : -1); ee[s] = ev[u];
            if (!inb) atomicOr(status, 16);
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx = t + u * 256, k = idx >> 4, j4 = idx & 15;
        if (k < K) *reinterpret_cast<float4*>(Bs + k * GC_LDB + 4 * j4) = vb[u];
    }
    __syncthreads();                                     // BN tables
    {
        int kc = 0, rr = 0;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            if (kc < nkc) {
                const int r = (rr << 5) + (t >> 3), k = (kc << 5) + ((t & 7) << 2);
                float* d = As + k * LDA + r;
                d[0] = fmaf(va[u].x, sc_s[k], sh_s[k]);
                d[LDA] = fmaf(va[u].y, sc_s[k + 1], sh_s[k + 1]);
                d[2 * LDA] = fmaf(va[u].z, sc_s[k + 2], sh_s[k + 2]);
                d[3 * LDA] = fmaf(va[u].w, sc_s[k + 3], sh_s[k + 3]);
            }
            if (++rr == R) { rr = 0; ++kc; }
        }
    }
    __syncthreads();
    // ---- z tile = BN(x) W on the matrix cores: wave w owns column tile w & 1 and row tile w >> 1 -----------------
    gc_f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    if (r0 < R) gconv_mma<false, LDA, GC_LDB>(As, Bs, K, r0, ct, li, lk, acc0, acc1);
    __syncthreads();                                     // every wave is done reading both stages
    float* Zs = Bs;
    float* At = As;                                      // At[(h * T + j) * LDA + i] = alpha of edge j -> i, head h0 + h
    if (r0 < R) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            Zs[row * GC_LDZ + ct * 32 + li] = acc0[r];
            if (row < rows) a.z[(size_t)(g0 + row) * H + n0 + ct * 32 + li] = acc0[r];
        }
    }
    {
        float4* z4 = reinterpret_cast<float4*>(At);
        for (int idx = t; idx < (GC_K * LDA) / 4; idx += 256) z4[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // ---- scores: one lane per (node, head of the slice) ----------------------------------------------------------
    const int pi = t & (T - 1), ph = t >> 6;             // this lane's (node, head-in-slice) pair
    const bool pair = ph < hs && pi < rows;
    float my_ad = 0.f, my_as = 0.f;
    if (pair) {
        const float* zr = Zs + pi * GC_LDZ + ph * D;
        const float* av = att_s + ph * 2 * D;
        for (int d = 0; d < D; ++d) { my_ad = fmaf(zr[d], av[d], my_ad); my_as = fmaf(zr[d], av[D + d], my_as); }
        ad_s[ph][pi] = my_ad; as_s[ph][pi] = my_as;
        a.adst[(size_t)(g0 + pi) * a.heads + h0 + ph] = my_ad;
        a.asrc[(size_t)(g0 + pi) * a.heads + h0 + ph] = my_as;
    }
    __syncthreads();
    // ---- edge softmax of row pi for head ph, attention dropout, dense block (duplicate edges accumulate) ---------
    if (pair) {
        const uint64_t seed = step_seed(a.seed, a.ctr);
        const float inv_keep = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
        const int hg = h0 + ph;
        const float eself = gg_lrelu(my_ad + my_as, a.slope);
        // Two passes over the row's slots, four slots per round of (dependent) LDS reads: the maximum, then the
        // UNNORMALISED weights exp(e - m) * keep into the block and their sum -- the division by the denominator is a
        // per-row scale of the aggregated tile in the epilogue, so no third pass rewrites the block.
        float m = eself;
        const int s0 = ptr_s[pi], s1 = ptr_s[pi + 1];
        for (int s = s0; s < s1; s += 4) {
            int j[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) j[q] = en[min(s + q, s1 - 1)];
            float e4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) e4[q] = as_s[ph][max(j[q], 0)];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (j[q] >= 0) m = fmaxf(m, gg_lrelu(my_ad + e4[q], a.slope));      // a repeated last slot does not change a max
        }
        const float pself = expf(eself - m);
        float lsum = pself;
        float* Ah = At + (size_t)ph * T * LDA;
        for (int s = s0; s < s1; s += 4) {
            int j[4], id[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int sq = min(s + q, s1 - 1); j[q] = en[sq]; id[q] = ee[sq]; }
            float e4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) e4[q] = as_s[ph][max(j[q], 0)];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (s + q < s1 && j[q] >= 0) {
                    const float pe = expf(gg_lrelu(my_ad + e4[q], a.slope) - m);
                    lsum += pe;
                    Ah[j[q] * LDA + pi] += pe * keep_scale(seed, id[q], hg, a.heads, a.p, inv_keep);
                }
            }
        }
        Ah[pi * LDA + pi] += pself * keep_scale(seed, a.E + g0 + pi, hg, a.heads, a.p, inv_keep);
        const float dn = lsum + 1e-16f;
        idn_s[ph][pi] = 1.f / dn;
        a.mx[(size_t)(g0 + pi) * a.heads + hg] = m;
        a.den[(size_t)(g0 + pi) * a.heads + hg] = dn;
    }
    __syncthreads();
    // ---- out tile = alpha_h z on the matrix cores: the 32-column tile ct lies in head (ct * 32) / D of the slice ----
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = 0.f;
    if (r0 < R) gconv_mma<false, LDA, GC_LDZ>(At + (size_t)((ct * 32) / D) * T * LDA, Zs, rowsP, r0, ct, li, lk, acc0, acc1);
    // ---- epilogue: bias, ReLU, store, column sums of this graph ---------------------------------------------------
    double s1 = 0.0, s2 = 0.0;
    const int col = n0 + ct * 32 + li;
    asm volatile("" :: "v"(bias));
    if (r0 < R) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            // softmax denominator of (row, head of this column tile), then bias; the backbone always applies ReLU (model.py:390)
            const float v = fmaxf(fmaf(acc0[r], idn_s[(ct * 32) / D][min(row, T - 1)], bias), 0.f);
            if (row < rows) {
                a.out[(size_t)(g0 + row) * H + col] = v;
                s1 += (double)v; s2 += (double)v * (double)v;
            }
        }
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lk == 0) { red[w][0][li] = s1; red[w][1][li] = s2; }
    __syncthreads();
    if (w < 2 && lk == 0 && want) {
        a.st_sum.add(col, red[w][0][li] + red[w + 2][0][li]);
        a.st_sq.add(col, red[w][1][li] + red[w + 2][1][li]);
    }
}

}  // namespace cal
