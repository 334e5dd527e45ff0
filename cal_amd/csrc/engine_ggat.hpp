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
constexpr int GG_E = 512;                      // stored edges per graph (forward: keeps the workgroup under 80 KB of LDS, two per CU)
constexpr int GGB_E = 512;                     // ... for the fused backward (three per-slot arrays in LDS)

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
    warm_kernargs<(sizeof(GgatArgs) + 64 < 1024 ? sizeof(GgatArgs) + 64 : 1024)>();
    constexpr int LDT = T + 4;                           // row stride of the j-major tiles of the second product (4 mod 32)
    __shared__ __attribute__((aligned(16))) float As[2 * T * LDT];          // x' rows [row][k] (stride GC_LDX, as k_gconv_fwd); later two attention blocks [h][i][j]
    __shared__ __attribute__((aligned(16))) float Bs[GC_K * GC_LDB];       // W slice [k][col]; later the z tile [row][col]
    __shared__ float sc_s[GC_K], sh_s[GC_K];
    __shared__ int ptr_s[T + 4];
    __shared__ signed char en[GG_E], er[GG_E];   // source / destination node of a slot, local to the graph (-1: edge leaves the graph)
    __shared__ int ee[GG_E];
    __shared__ float le_s[2][GG_E + T], m_s[2][T];  // per (head, slot): logit, then exp(logit - max); per (head, node): max
    __shared__ float att_s[2 * GC_N];          // [head in slice][2 D]
    __shared__ float ad_s[2][T], as_s[2][T], idn_s[2][T];
    __shared__ double red[4][2][32];
    const int b = blockIdx.x, n0 = blockIdx.y * GC_N, t = threadIdx.x;
    const int g0 = gptr[b], rows = gptr[b + 1] - g0, e0 = eptr[b], ne = eptr[b + 1] - e0;
    const bool want = a.st_sum.on();
    if (rows <= 0) {
        if (a.bn.update && blockIdx.x == 0 && blockIdx.y == 0 && t < K) { const BNRaw r0 = bn_raw_load_st(a.bn, t); bn_raw_update_running(a.bn, r0, t); }
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
    // small operands: unconditional loads on clamped indices / substituted pointers, pinned with the tiles (branch-free
    // prologue, BNRaw in engine.hpp)
    int pv = g.ptr[g0 + min(t, rows)];
    int nv[4], ev[4];
    const int slot_hi = max(g.nnz - 1, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int s = min(e0 + max(min(t + u * 256, ne - 1), 0), slot_hi);
        nv[u] = g.nbr[s];
        ev[u] = g.eid[s];
    }
    float attv = a.att[(size_t)h0 * 2 * D + min(t, 2 * GC_N - 1)];       // hs heads x 2 D = 128 floats, contiguous
    const int lane = t & 63, li = lane & 31, lk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ct = w & 1, r0 = w >> 1;
    const float* biasp = a.bias ? a.bias : a.W;          // W: any valid [>= H] float array; the value is masked below
    float bias = biasp[n0 + ct * 32 + li];
    BNRawS braws = bn_raws_load(a.bn, min(t, K - 1));     // (striped reader, engine.hpp)
#pragma unroll
    for (int u = 0; u < UA; ++u) ro_pin(va[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) ro_pin(vb[u]);
    bn_raws_pin(braws);
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(nv[u]), "+v"(ev[u]));
    asm volatile("" : "+v"(pv), "+v"(attv), "+v"(bias));
    const BNRaw braw = bn_raws_sum(a.bn, braws);
    if (!a.bias) bias = 0.f;
    if (t >= 2 * GC_N) attv = 0.f;
    if (ne <= 0) {                                       // no slot of this graph exists: the clamped loads fetched no index
#pragma unroll
        for (int u = 0; u < 4; ++u) { nv[u] = g0; ev[u] = 0; }
    }
    if (t < K) {
        bn_raw_scale_shift(a.bn, braw, sc_s[t], sh_s[t]);
        if (a.bn.update && blockIdx.x == 0 && blockIdx.y == 0) bn_raw_update_running(a.bn, braw, t);
    }
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
    if (t < rows) {                                      // destination row of every CSR slot (stores only)
        const int s1 = ptr_s[t + 1];
        for (int s = ptr_s[t]; s < s1; ++s) er[s] = (signed char)t;
    }
    {
        int kc = 0, rr = 0;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            if (kc < nkc) {
                const int r = (rr << 5) + (t >> 3), k = (kc << 5) + ((t & 7) << 2);
                *reinterpret_cast<float4*>(As + r * GC_LDX + k) =
                    make_float4(fmaf(va[u].x, sc_s[k], sh_s[k]), fmaf(va[u].y, sc_s[k + 1], sh_s[k + 1]),
                                fmaf(va[u].z, sc_s[k + 2], sh_s[k + 2]), fmaf(va[u].w, sc_s[k + 3], sh_s[k + 3]));
            }
            if (++rr == R) { rr = 0; ++kc; }
        }
    }
    __syncthreads();
    // ---- z tile = BN(x) W on the matrix cores: wave w owns column tile w & 1 and row tile w >> 1 -----------------
    gc_f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    if (r0 < R) gconv_mma_arow<false, GC_LDB>(As, Bs, K, r0, ct, li, lk, acc0, acc1);
    __syncthreads();                                     // every wave is done reading both stages
    float* Zt = Bs;                                      // Zt[col * LDT + j] = z[j][col]: the z tile transposed (as k_gconv_fwd)
    float* At = As;                                      // At[(h * T + i) * LDT + j] = alpha of edge j -> i, head h0 + h
    if (r0 < R) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int row = r0 * 32 + 8 * gq + 4 * lk;
            *reinterpret_cast<float4*>(Zt + (ct * 32 + li) * LDT + row) = make_float4(acc0[4 * gq], acc0[4 * gq + 1], acc0[4 * gq + 2], acc0[4 * gq + 3]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < rows) a.z[(size_t)(g0 + row) * H + n0 + ct * 32 + li] = acc0[r];
        }
    }
    {
        float4* z4 = reinterpret_cast<float4*>(At);
        for (int idx = t; idx < (2 * T * LDT) / 4; idx += 256) z4[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // ---- scores: one lane per (node, head of the slice) ----------------------------------------------------------
    const int pi = t & (T - 1), ph = t >> 6;             // this lane's (node, head-in-slice) pair
    const bool pair = ph < hs && pi < rows;
    float my_ad = 0.f, my_as = 0.f;
    if (pair) {
        const float* zr = Zt + (ph * D) * LDT + pi;          // column ph * D + d of node pi: lanes = consecutive nodes
        const float* av = att_s + ph * 2 * D;
        for (int d0 = 0; d0 < D; d0 += 8) {                  // 24 independent LDS reads per round (D is 32 or 64)
            float zz[8], a1[8], a2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { zz[u] = zr[(d0 + u) * LDT]; a1[u] = av[d0 + u]; a2[u] = av[D + d0 + u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { my_ad = fmaf(zz[u], a1[u], my_ad); my_as = fmaf(zz[u], a2[u], my_as); }
        }
        ad_s[ph][pi] = my_ad; as_s[ph][pi] = my_as;
        a.adst[(size_t)(g0 + pi) * a.heads + h0 + ph] = my_ad;
        a.asrc[(size_t)(g0 + pi) * a.heads + h0 + ph] = my_as;
    }
    __syncthreads();
    // ---- edge softmax, attention dropout, dense block.  Per (node, head) only the cheap parts (maximum, sum) walk the row's
    // slots, in batches of independent LDS reads; everything with an expf / mask hash runs one lane per (slot, head) -- the
    // hub rows of a BA graph otherwise set the kernel's critical path (3.7 us of 14.6 for an average graph, more for the
    // worst).  The block holds UNNORMALISED weights exp(e - m) * keep; the denominator is a per-row scale in the epilogue.
    const uint64_t seed = step_seed(a.seed, a.ctr);
    const float inv_keep = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
    const int nit = (ne + rows) * hs;                    // items: slot-major, head-minor; slots ne .. ne + rows - 1 are the self loops
    for (int it = t; it < nit; it += 256) {
        const int s = it / hs, h = it - s * hs;
        const bool self = s >= ne;
        const int i = self ? s - ne : er[s], j = self ? s - ne : en[s];
        le_s[h][s] = j >= 0 ? gg_lrelu(ad_s[h][i] + as_s[h][j], a.slope) : -3.0e38f;
    }
    __syncthreads();
    if (pair) {
        float m = le_s[ph][ne + pi];
        const int s1 = ptr_s[pi + 1];
        for (int s = ptr_s[pi]; s < s1; s += 8) {
            float x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = le_s[ph][min(s + q, s1 - 1)];
#pragma unroll
            for (int q = 0; q < 8; ++q) m = fmaxf(m, x[q]);
        }
        m_s[ph][pi] = m;
    }
    __syncthreads();
    for (int it = t; it < nit; it += 256) {
        const int s = it / hs, h = it - s * hs;
        const bool self = s >= ne;
        const int i = self ? s - ne : er[s], j = self ? s - ne : en[s];
        float pe = 0.f;
        if (j >= 0) {
            pe = expf(le_s[h][s] - m_s[h][i]);
            const float kp = keep_scale(seed, self ? a.E + g0 + i : (int64_t)ee[s], h0 + h, a.heads, a.p, inv_keep);
            atomicAdd(&At[((size_t)h * T + i) * LDT + j], pe * kp);
        }
        le_s[h][s] = pe;
    }
    __syncthreads();
    if (pair) {
        float lsum = le_s[ph][ne + pi];
        const int s1 = ptr_s[pi + 1];
        for (int s = ptr_s[pi]; s < s1; s += 8) {
            float x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = le_s[ph][min(s + q, s1 - 1)];
#pragma unroll
            for (int q = 0; q < 8; ++q) lsum += s + q < s1 ? x[q] : 0.f;
        }
        const float dn = lsum + 1e-16f;
        idn_s[ph][pi] = 1.f / dn;
        a.mx[(size_t)(g0 + pi) * a.heads + h0 + ph] = m_s[ph][pi];
        a.den[(size_t)(g0 + pi) * a.heads + h0 + ph] = dn;
    }
    __syncthreads();
    // ---- out tile = alpha_h z on the matrix cores: the 32-column tile ct lies in head (ct * 32) / D of the slice ----
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = 0.f;
    if (r0 < R) gconv_mma_rowk<false, LDT>(At + (size_t)((ct * 32) / D) * T * LDT, Zt, rowsP, r0, ct, li, lk, acc0, acc1);
    // ---- epilogue: bias, ReLU, store, column sums of this graph ---------------------------------------------------
    // (a lane's 16 terms of the column sums in fp32, masked and unguarded, as in k_gconv_fwd; the denominators in one batch)
    float f1[4] = {0.f, 0.f, 0.f, 0.f}, f2[4] = {0.f, 0.f, 0.f, 0.f};
    const int col = n0 + ct * 32 + li;
    asm volatile("" :: "v"(bias));
    if (r0 < R) {
        float idn[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) idn[r] = idn_s[(ct * 32) / D][min(r0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, T - 1)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            // softmax denominator of (row, head of this column tile), then bias; the backbone always applies ReLU (model.py:390)
            const float v = fmaxf(fmaf(acc0[r], idn[r], bias), 0.f);
            if (row < rows) a.out[(size_t)(g0 + row) * H + col] = v;
            const float vm = row < rows ? v : 0.f;
            f1[r & 3] += vm; f2[r & 3] = fmaf(vm, vm, f2[r & 3]);
        }
    }
    double s1 = ((double)f1[0] + (double)f1[1]) + ((double)f1[2] + (double)f1[3]);
    double s2 = ((double)f2[0] + (double)f2[1]) + ((double)f2[2] + (double)f2[3]);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lk == 0) { red[w][0][li] = s1; red[w][1][li] = s2; }
    __syncthreads();
    if (w < 2 && lk == 0 && want) {
        a.st_sum.add(col, red[w][0][li] + red[w + 2][0][li]);
        a.st_sq.add(col, red[w][1][li] + red[w + 2][1][li]);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Per-graph fused GATConv layer, backward.  With g = dOut (gradient at the layer output, ReLU mask applied), per head:
//     dalpha_ij = keep_ij <g_i, z_j>,   S_i = sum_j alpha_ij dalpha_ij,   de_ij = alpha_ij (dalpha_ij - S_i),
//     dr_ij = de_ij lrelu'(a_dst_i + a_src_j),   d a_dst_i = sum_j dr_ij,   d a_src_j = sum_i dr_ij,
//     dz_j = sum_i alpha_ij keep_ij g_i + d a_dst_j att_dst + d a_src_j att_src,   d att = (sum_j d a_dst_j z_j, sum_j d a_src_j z_j)
// then, exactly as k_gconv_bwd (engine_gconv_bwd.hpp):  dX' = dz W^T (partial per 64-column slice, + BatchNorm-backward
// sums),  dW = x'^T dz (per-graph slab).  One workgroup per (graph, 64-column slice = 64 / D heads), 512 threads; the
// heads of the slice go one after the other through two dense 64 x 64 blocks in LDS: dalpha = G_h Z_h^T on MFMA, the two
// passes over the CSR slots of every destination row, alpha~^T, dz_h = alpha~^T G_h on MFMA.  Replaces k_gat_bwd_dst /
// _src / k_gat_datt_part + the dual GEMM; z and the saved scores / max / denominator come from the forward.
// ------------------------------------------------------------------------------------------------------------------
struct GgatBwdArgs {
    const float* dout;       // [N,H]
    const float* x;          // [N,K] raw layer input
    const float* W;          // [K,H]
    const float* att;        // [heads, 2 D]
    const float* z;          // [N,H] BN(x) W of the forward
    const float* adst; const float* asrc; const float* mx; const float* den;     // [N,heads] of the forward
    BNRef bn;
    float* dxp0; float* dxp1; // [N,K] partial dX' of output-column slice 0 / 1
    float* slab;             // [B][K,H] per-graph dW
    float* att_slab;         // [B][heads * 2 D] per-graph d att
    double* dot_parts;       // [B * H/64][2K]
    double* dacc_sum; double* dacc_prod; int dacc_ss;    // or (non-null): into the workgroup's accumulator plane (engine.hpp: stripe_sum)
    int heads, D;
    float slope, p;
    uint64_t seed;
    const uint64_t* ctr;
    int64_t E;
    // UP variant (as k_gconv_bwd's): dOut is not materialised but built while it is staged from the layer ABOVE's two
    // partial dX' (its BatchNorm-backward + this layer's ReLU mask); the per-graph column sums of dOut (this layer's
    // bias gradient) go to bias_parts [B][H].
    const float* dy0; const float* dy1;
    const float* y;          // [N,H] this layer's output after ReLU = the upper BatchNorm's input
    BNRef ubn;
    const double* udot_sum; const double* udot_prod;
    double* bias_parts;
};

template <bool UP>
__global__ void __launch_bounds__(GB_NT) k_ggat_bwd(const CSR g, const int* __restrict__ gptr, const int* __restrict__ eptr,
                                                  const GgatBwdArgs a, int N, int H, int K, int* __restrict__ status) {
    constexpr int T = GB_T;
    __shared__ float um_s[GC_N], ur_s[GC_N], ug_s[GC_N], u1_s[GC_N], u2_s[GC_N];     // upper BatchNorm, this slice's columns
    __shared__ float bs_s[GB_NT / 64][16][4];
    __shared__ __attribute__((aligned(16))) float Bk[T * GB_LDJ];          // alpha~ of the slice's first head: Bk[i][j] (edge j -> i); the second head's block and dalpha live in Ws until W is committed
    __shared__ __attribute__((aligned(16))) float Ds[T * GB_LDD];          // dOut slice [j][n]; later dz [i][n]
    __shared__ __attribute__((aligned(16))) float Zt[GC_N * GB_LDJ];       // dalpha of the first head [i][j]; later dz^T [n][i]
    __shared__ __attribute__((aligned(16))) float Ws[GC_K * GB_LDD];       // W[:, ns] as loaded: Ws[k_in][n] (16 B operand reads, see k_gconv_bwd)
    __shared__ __attribute__((aligned(16))) float Xs[T * GB_LDX];          // x_hat rows [i][k_in]
    __shared__ __attribute__((aligned(16))) float Zr[T * GB_LDD];          // z slice [j][n]
    __shared__ float mean_s[GC_K], rstd_s[GC_K], gam_s[GC_K], bet_s[GC_K];
    __shared__ int ptr_s[T + 4];
    __shared__ signed char en[GGB_E], er[GGB_E];        // source / destination node of a CSR slot, local to the graph
    __shared__ int ee[GGB_E];
    __shared__ float al_s[2][GGB_E + T], dk_s[2][GGB_E + T], dr_s[2][GGB_E + T];     // per (head of the slice, slot): edges, then self loops
    __shared__ float att_s[2 * GC_N];
    __shared__ float ad_s[2][T], as_s[2][T], mx_s[2][T], dn_s[2][T], dad_s[2][T], das_s[2][T];
    BLK_CLK(0);
    warm_kernargs<(sizeof(GgatBwdArgs) + 64 < 1024 ? sizeof(GgatBwdArgs) + 64 : 1024)>();
    const int b = blockIdx.x, sl = blockIdx.y, ns0 = sl * GC_N, t = threadIdx.x;
    const int g0 = gptr[b], rows = gptr[b + 1] - g0, e0 = eptr[b], ne = eptr[b + 1] - e0;
    const int lane = t & 63, li = lane & 31, lk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int D = a.D, hs = GC_N / D, h0 = ns0 / D;
    double* parts = a.dot_parts + ((size_t)sl * gridDim.x + b) * (2 * K);
    float* slab = a.slab + (size_t)b * K * H;
    float* aslab = a.att_slab + (size_t)b * a.heads * 2 * D + (size_t)h0 * 2 * D;      // this slice's hs * 2 D = 128 entries
    if (rows <= 0 || rows > T || ne > GGB_E || ne < 0) {
        if (rows > 0 && t == 0) atomicOr(status, 8);
        if (!a.dacc_sum) for (int i = t; i < 2 * K; i += GB_NT) parts[i] = 0.0;
        for (int i = t; i < K * GC_N; i += GB_NT) slab[(size_t)(i / GC_N) * H + ns0 + i % GC_N] = 0.f;
        if (t < 2 * GC_N) aslab[t] = 0.f;
        if (UP && t < GC_N) a.bias_parts[(size_t)b * H + ns0 + t] = 0.0;
        return;
    }
    const int rowsP = (rows + 31) & ~31, R = rowsP >> 5, K4 = K >> 2;
    // ---- every global load of the kernel, issued before the first wait ------------------------------------------
    RoBatch<float4, 2> bd, bd1, by, bz;                  // dOut (UP: dy0, dy1, y) / z [g0 + j][ns0 + 4 n4 ..]: rows x 16 float4
    RoBatch<float4, 4> bx, bw;                           // x[g0 + i][4 k4 ..];  W[k_in][ns0 + 4 n4 ..]
    {
        const float* d0 = UP ? a.dy0 : a.dout;
        ro_issue<GB_NT>(bd, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(d0 + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
        if (UP) {
            const float* d1 = a.dy1 ? a.dy1 : a.dy0;
            ro_issue<GB_NT>(bd1, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(d1 + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
            ro_issue<GB_NT>(by, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(a.y + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
        }
    }
    ro_issue<GB_NT>(bz, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(a.z + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
    ro_issue<GB_NT>(bx, rows, K4, [&](int i, int k4) { return *reinterpret_cast<const float4*>(a.x + (size_t)(g0 + i) * K + 4 * k4); });
    ro_issue<GB_NT>(bw, K, 16, [&](int k, int n4) { return *reinterpret_cast<const float4*>(a.W + (size_t)k * H + ns0 + 4 * n4); });
    // the small operands: unconditional loads on clamped indices, pinned below (branch-free prologue, see BNRaw in engine.hpp:
    // as selects / guarded blocks they were six serial round trips behind the tile loads)
    int pv = g.ptr[g0 + min(t, rows)], pn = g.ptr[g0 + min(t + 1, rows)];
    int nv[2], ev[2];
    const int slot_hi = max(g.nnz - 1, 0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int s = min(e0 + max(min(t + u * GB_NT, ne - 1), 0), slot_hi);
        nv[u] = g.nbr[s];
        ev[u] = g.eid[s];
    }
    // forward scores of this slice's heads: lane (kind, head, node) = (t >> 7, (t >> 6) & 1, t & 63)
    const int sn = t & 63, sh = (t >> 6) & 1, sk = t >> 7;
    const float* ssrc = sk == 0 ? a.adst : (sk == 1 ? a.asrc : (sk == 2 ? a.mx : a.den));
    float scv = ssrc[(size_t)(g0 + min(sn, rows - 1)) * a.heads + h0 + min(sh, hs - 1)];
    float attv = a.att[(size_t)h0 * 2 * D + min(t, 2 * GC_N - 1)];
    // (striped readers, engine.hpp; one register set for the two BatchNorms: lanes 0 .. K-1 this layer's, 256 .. 319 the upper one's)
    const bool ulane = UP && t >= 256;
    BNRawS braws = UP ? bn_raws_load2(a.bn, min(t, K - 1), a.ubn, ns0 + (t & (GC_N - 1)), ulane) : bn_raws_load(a.bn, min(t, K - 1));
    StripeVal ud1s, ud2s;
    if (UP) {
        const int c = ns0 + (t & (GC_N - 1));
        ud1s = stripe_load(a.udot_sum, c, a.ubn.ss); ud2s = stripe_load(a.udot_prod, c, a.ubn.ss);
    }
    bn_raws_pin(braws);
    if (UP) { stripe_pin(ud1s); stripe_pin(ud2s); }
    asm volatile("" : "+v"(pv), "+v"(pn), "+v"(nv[0]), "+v"(nv[1]), "+v"(ev[0]), "+v"(ev[1]), "+v"(scv), "+v"(attv));
    if (ne <= 0) { nv[0] = g0; nv[1] = g0; ev[0] = 0; ev[1] = 0; }   // no slot of this graph exists: the clamped loads fetched no index
    if (sh >= hs) scv = 0.f;
    if (t >= 2 * GC_N) attv = 0.f;
    if (t < K) {
        float m1, r1;
        const BNRaw braw = bn_raws_sum(a.bn, braws);
        bn_raw_mean_rstd(a.bn, braw, m1, r1);
        mean_s[t] = m1; rstd_s[t] = r1;
        gam_s[t] = braw.g;
        bet_s[t] = braw.b;
    }
    // ---- stage everything in LDS -----------------------------------------------------------------------------------
    if (t <= rows) ptr_s[t] = pv - e0;
    if (t < rows) for (int s = pv - e0; s < pn - e0; ++s) er[s] = (signed char)t;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int s = t + u * GB_NT;
        if (s < ne) {
            const int loc = nv[u] - g0;
            const bool inb = loc >= 0 && loc < rows;
            en[s] = (signed char)(inb ? loc : -1); ee[s] = ev[u];
            if (!inb) atomicOr(status, 16);
        }
    }
    if (sh < hs) {
        float* dst = sk == 0 ? &ad_s[sh][sn] : (sk == 1 ? &as_s[sh][sn] : (sk == 2 ? &mx_s[sh][sn] : &dn_s[sh][sn]));
        *dst = sn < rows ? scv : (sk == 3 ? 1.f : 0.f);
    }
    if (t < 2 * GC_N) att_s[t] = attv;
    if (t < 2 * T) { dad_s[t >> 6][t & 63] = 0.f; das_s[t >> 6][t & 63] = 0.f; }
    if (UP && t >= 256 && t < 256 + GC_N) {              // upper BatchNorm constants of this slice's 64 columns
        float m1, r1;
        const BNRaw uraw = bn_raws_sum(a.ubn, braws);
        const double ud1 = stripe_total(ud1s, a.ubn.ss), ud2 = stripe_total(ud2s, a.ubn.ss);
        bn_raw_mean_rstd(a.ubn, uraw, m1, r1);
        um_s[t - 256] = m1; ur_s[t - 256] = r1;
        ug_s[t - 256] = uraw.g * r1;
        u1_s[t - 256] = (float)(ud1 * (double)a.ubn.inv_n);
        u2_s[t - 256] = (float)(ud2 * (double)a.ubn.inv_n);
    }
    if (!UP) ro_commit<GB_NT>(bd, rows, 16, [&](int j, int n4, const float4 v) { *reinterpret_cast<float4*>(Ds + j * GB_LDD + 4 * n4) = v; });
    ro_commit<GB_NT>(bz, rows, 16, [&](int j, int n4, const float4 v) { *reinterpret_cast<float4*>(Zr + j * GB_LDD + 4 * n4) = v; });
    __syncthreads();                                     // BN constants (W stays in registers: its LDS tile holds the second head's blocks first)
    ro_commit<GB_NT>(bx, rows, K4, [&](int i, int k4, float4 v) {
        const int k = 4 * k4;
        v.x = (v.x - mean_s[k]) * rstd_s[k]; v.y = (v.y - mean_s[k + 1]) * rstd_s[k + 1];
        v.z = (v.z - mean_s[k + 2]) * rstd_s[k + 2]; v.w = (v.w - mean_s[k + 3]) * rstd_s[k + 3];
        *reinterpret_cast<float4*>(Xs + i * GB_LDX + k) = v;
    });
    if (UP) {
        // dOut slice from the upper layer's partials: lane t always holds column group t % 16, so its column sums stay
        // in registers until the cross-lane reduction below
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
        const bool two = a.dy1 != nullptr;
#pragma unroll
        for (int u = 0; u < 2; ++u) { ro_pin(bd.v[u]); ro_pin(bd1.v[u]); ro_pin(by.v[u]); }
        const int c = 4 * (t & 15);
#pragma unroll
        for (int u = 0; u < 2; ++u) {                   // item (u, t) = row t / 16 + 32 u, column group t % 16
            const int j = (t >> 4) + u * (GB_NT / 16);
            if (j < rows) {
                const float4 v0 = bd.v[u], v1 = bd1.v[u], yv = by.v[u];
                const float d[4] = {v0.x + (two ? v1.x : 0.f), v0.y + (two ? v1.y : 0.f), v0.z + (two ? v1.z : 0.f), v0.w + (two ? v1.w : 0.f)};
                const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float yn = (yy[q] - um_s[c + q]) * ur_s[c + q];
                    const float g1 = ug_s[c + q] * (d[q] - u1_s[c + q] - yn * u2_s[c + q]);
                    o[q] = yy[q] > 0.f ? g1 : 0.f;
                    cs[q] += o[q];
                }
                *reinterpret_cast<float4*>(Ds + j * GB_LDD + c) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cs[q] += __shfl_xor(cs[q], 16, 64);
            cs[q] += __shfl_xor(cs[q], 32, 64);
        }
        if (lane < 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bs_s[t >> 6][lane][q] = cs[q];
        }
    }
    for (int i = t; i < (rowsP - rows) * GB_LDD; i += GB_NT) { Ds[rows * GB_LDD + i] = 0.f; Zr[rows * GB_LDD + i] = 0.f; }
    for (int i = t; i < (rowsP - rows) * GB_LDX; i += GB_NT) Xs[rows * GB_LDX + i] = 0.f;
    __syncthreads();
    if (UP && t < GC_N) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < GB_NT / 64; ++k) tot += (double)bs_s[k][t >> 2][t & 3];
        a.bias_parts[(size_t)b * H + ns0 + t] = tot;
    }
    auto ident = [](float v) { return v; };
    gc_f32x16 acc[2], dzacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dzacc[0][i] = 0.f; dzacc[1][i] = 0.f; }
    const uint64_t seed = step_seed(a.seed, a.ctr);
    const float inv_keep = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
    const int rt = w >> 1, ct = w & 1;                   // waves 0-3: the 32 x 32 tile (rt, ct) of a 64 x 64 product
    BLK_CLK(2);
    // ---- the heads of the slice, side by side (D = 32: two heads, 256 lanes each in the per-slot passes; D = 64: one) ----------
    // One head after the other, this section was 4 barriers and ~7 us per head with a quarter of the lanes busy.
    static_assert(GC_K * GB_LDD >= 2 * GC_N * GB_LDJ, "the W tile holds the second head's dalpha and alpha~ blocks");
    float* const Zt1 = Ws;                               // dalpha of the second head
    float* const Bk1 = Ws + GC_N * GB_LDJ;               // alpha~ of the second head
    const int nsl = ne + rows;                           // slots: the graph's edges, then one self loop per node
    const int LPH = GB_NT / hs;                          // lanes per head in the per-slot passes
    const int hq = t / LPH, tl = t - hq * LPH;
    {
        // (a) dalpha blocks: Zt_h[i][j] = <g_i, z_j> over the head's D columns -- waves 0-3 the first head, 4-7 the second
        const int ha = w >> 2, rta = (w & 3) >> 1, cta = w & 1;
        if (ha < hs && rta < R && cta < R) {
            float* Zh = ha ? Zt1 : Zt;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
            gb_mma_rowk(Ds + (rta * 32 + li) * GB_LDD + ha * D, Zr + (cta * 32 + li) * GB_LDD + ha * D, D, lk, acc[0]);      // (16 B reads: the 4 B form was a 4-way bank conflict)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = rta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                Zh[i * GB_LDJ + cta * 32 + li] = acc[0][r];
            }
        }
        for (int i = t; i < (T * GB_LDJ) / 4; i += GB_NT) {
            reinterpret_cast<float4*>(Bk)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hs > 1) reinterpret_cast<float4*>(Bk1)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    // (b1) one lane per (head, slot): alpha, alpha * keep (stays in a register for b2), keep * dalpha.
    //      (One lane per destination ROW made the hub rows of a BA graph the critical path: 2 passes x 30 slots of
    //      expf + mask hash in one lane, 7 us per head.)
    float akr[3] = {0.f, 0.f, 0.f};
    {
        const float* Zh = hq ? Zt1 : Zt;
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int s = tl + it * LPH;
            if (s < nsl) {
                const bool self = s >= ne;
                const int i = self ? s - ne : er[s], j = self ? s - ne : en[s];
                float al = 0.f, kp = 0.f, dk = 0.f;
                if (j >= 0) {
                    al = expf(gg_lrelu(ad_s[hq][i] + as_s[hq][j], a.slope) - mx_s[hq][i]) / dn_s[hq][i];
                    kp = keep_scale(seed, self ? a.E + g0 + i : (int64_t)ee[s], h0 + hq, a.heads, a.p, inv_keep);
                    dk = Zh[i * GB_LDJ + j] * kp;
                }
                al_s[hq][s] = al; akr[it] = al * kp; dk_s[hq][s] = dk;
            }
        }
    }
    __syncthreads();
    // (b2 + b3) one lane per (head, slot): S_i = sum over the slots of its destination row of alpha * dalpha, computed by the lane
    //      itself in slot order (8 independent LDS reads per round; the row sums used to be a phase of their own -- one
    //      lane per row, a barrier, the other 450 lanes idle), then d(raw logit) into dr_s and alpha~ into the block
    //      (atomic: duplicate edges share an entry)
    {
        float* Bh = hq ? Bk1 : Bk;
        const float* alh = al_s[hq];
        const float* dkh = dk_s[hq];
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int s = tl + it * LPH;
            if (s < nsl) {
                const bool self = s >= ne;
                const int i = self ? s - ne : er[s], j = self ? s - ne : en[s];
                float dr = 0.f;
                if (j >= 0) {
                    float S = alh[ne + i] * dkh[ne + i];
                    const int s1 = ptr_s[i + 1];
                    for (int q0 = ptr_s[i]; q0 < s1; q0 += 8) {
                        float x[8], y[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) { const int sq = min(q0 + q, s1 - 1); x[q] = alh[sq]; y[q] = dkh[sq]; }
#pragma unroll
                        for (int q = 0; q < 8; ++q) S = fmaf(q0 + q < s1 ? x[q] : 0.f, y[q], S);
                    }
                    const float raw = ad_s[hq][i] + as_s[hq][j];
                    dr = alh[s] * (dkh[s] - S) * (raw > 0.f ? 1.f : a.slope);
                    atomicAdd(&Bh[i * GB_LDJ + j], akr[it]);
                }
                dr_s[hq][s] = dr;
            }
        }
    }
    __syncthreads();
    // (c) waves 0-3: dz tile (rows j, 32 columns: the head those columns belong to) = alpha~^T G_h; waves 4-7: d a_src_j (the
    //     slots whose source is j, scanned in slot order by two lanes per (head, node): deterministic), then d a_dst_i (row sums of dr)
    if (w < 4) {
        if (rt < R) {
            const float* Bh = (ct * 32) / D ? Bk1 : Bk;
            gb_mma<1, 1, GB_LDJ, GB_LDD>(Bh + rt * 32 + li, nullptr, Ds + ct * 32 + li, nullptr, rowsP, lk, ident, dzacc);
        }
    } else {
        const int q = t - 256;
        {
            const int hh = q >> 7, j = (q & 127) >> 1, half = q & 1;
            const int qb = half ? (ne + 1) / 2 : 0, qe = half ? ne : (ne + 1) / 2;
            const float* drh = dr_s[min(hh, hs - 1)];
            float sj = 0.f;
            if (hh < hs && j < rows) {
                for (int q0 = qb; q0 < qe; q0 += 8) {
                    int e8[8];
                    float y[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int sq = min(q0 + u, qe - 1); e8[u] = en[sq]; y[u] = drh[sq]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) sj += (q0 + u < qe && e8[u] == j) ? y[u] : 0.f;
                }
            }
            const float other = __shfl_xor(sj, 1, 64);
            if (hh < hs && j < rows && half == 0) das_s[hh][j] = drh[ne + j] + sj + other;
        }
        if (q < 2 * T) {
            const int hh = q >> 6, i = q & 63;
            if (hh < hs && i < rows) {
                const float* drh = dr_s[hh];
                float sd = drh[ne + i];
                const int s1 = ptr_s[i + 1];
                for (int s0 = ptr_s[i]; s0 < s1; s0 += 8) {
                    float y[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) y[u] = drh[min(s0 + u, s1 - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sd += s0 + u < s1 ? y[u] : 0.f;
                }
                dad_s[hh][i] = sd;
            }
        }
    }
    __syncthreads();
    ro_commit<GB_NT>(bw, K, 16, [&](int k, int n4, const float4 v) { *reinterpret_cast<float4*>(Ws + k * GB_LDD + 4 * n4) = v; });
    BLK_CLK(3);
    // ---- dz = aggregated part + the two rank-1 terms; d att of this slice's heads -----------------------------------------
    if (w < 4 && rt < R) {
        const int n = ct * 32 + li, hh = n / D, d = n % D;
        const float a_dst = att_s[hh * 2 * D + d], a_src = att_s[hh * 2 * D + D + d];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const float v = dzacc[0][r] + dad_s[hh][j] * a_dst + das_s[hh][j] * a_src;
            Ds[j * GB_LDD + n] = v;
        }
    } else if (w >= 4 && t - 256 < 2 * GC_N) {
        const int q = t - 256, hh = q / (2 * D), r2 = q % (2 * D);
        if (hh < hs) {
            const float* wv = r2 < D ? dad_s[hh] : das_s[hh];
            const int c = hh * D + (r2 < D ? r2 : r2 - D);
            float s = 0.f;
            for (int j0 = 0; j0 < rows; j0 += 8) {            // 16 independent LDS reads per round
                float x[8], y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = min(j0 + u, rows - 1); x[u] = wv[j]; y[u] = Zr[j * GB_LDD + c]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) s = fmaf(j0 + u < rows ? x[u] : 0.f, y[u], s);
            }
            aslab[q] = s;
        }
    }
    __syncthreads();
    // ---- P2: partial dX'[:, :] = dz[:, ns] W[:, ns]^T (+ BatchNorm-backward sums), as k_gconv_bwd ---------------------------
    if (w < 4 && w * 32 < K) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
        if (R == 2) gb_mma_rowk2<true>(Ds + li * GB_LDD, Ds + (32 + li) * GB_LDD, Ws + (w * 32 + li) * GB_LDD, GC_N, lk, acc[0], acc[1]);
        else gb_mma_rowk2<false>(Ds + li * GB_LDD, nullptr, Ws + (w * 32 + li) * GB_LDD, GC_N, lk, acc[0], acc[1]);
        const int k = w * 32 + li;
        float* dxp = sl ? a.dxp1 : a.dxp0;
        // x_hat of all rows as ONE batch of unconditional LDS reads, masked afterwards, and a lane's 32 terms summed in
        // fp32 (see k_gconv_bwd: the `q < R ? Xs[..] : 0` form is 32 branches with a ds_read + wait each)
        float xh[2][16];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) xh[q][r] = Xs[(q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * GB_LDX + k];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) { asm volatile("" : "+v"(xh[q][r])); if (q >= R) xh[q][r] = 0.f; }
        float f1[4] = {0.f, 0.f, 0.f, 0.f}, f2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float v = acc[q][r];                   // (row tile 1 of a one-tile graph: zero accumulators)
                if (i < rows) dxp[(size_t)(g0 + i) * K + k] = v;
                f1[r & 3] += v;
                f2[r & 3] = fmaf(v, xh[q][r], f2[r & 3]);
            }
        }
        double s1 = ((double)f1[0] + (double)f1[1]) + ((double)f1[2] + (double)f1[3]);
        double s2 = ((double)f2[0] + (double)f2[1]) + ((double)f2[2] + (double)f2[3]);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (lk == 0) {
            if (a.dacc_sum) {
                const size_t po = (size_t)stripe_of_block() * a.dacc_ss + k;
                atomicAdd(a.dacc_sum + po, s1); atomicAdd(a.dacc_prod + po, s2);
            } else { parts[k] = s1; parts[K + k] = s2; }
        }
    }
    // ---- P3: dW[:, ns] (this graph) = x'^T dz[:, ns] ------------------------------------------------------------------------
    if (w >= 4 && (w - 4) * 32 < K) {
        const int wq = w - 4, k = wq * 32 + li;
        const float gam = gam_s[k], bet = bet_s[k];
        auto affine = [&](float v) { return fmaf(v, gam, bet); };
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
        gb_mma<1, 2, GB_LDX, GB_LDD>(Xs + wq * 32 + li, nullptr, Ds + li, Ds + 32 + li, rowsP, lk, affine, acc);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = wq * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                slab[(size_t)kk * H + ns0 + q * 32 + li] = acc[q][r];
            }
    }
    BLK_CLK(1);
}

}  // namespace cal
