// Per-graph fused GINConv layer of the step engine (CausalGIN, model.py:188-194, 244-245):
//     GINConv(Sequential(Linear(H,H), BatchNorm1d(H), ReLU(), Linear(H,H), ReLU())), eps = 0:
//     t1 = W1 (h + sum_{j -> v} h_j) + b1,   y = relu(BN(t1)),   h' = relu(W2 y + b2)
// for mini-batches of graphs (or tiles of graphs) of at most 64 nodes, in the shape of engine_gconv.hpp / engine_gconv_bwd.hpp:
// a workgroup owns one graph and a 64-column slice of the OUTPUT features, the graph's unit adjacency block (A + I) sits
// in LDS and the aggregation is a second MFMA product.  The unweighted aggregation commutes with the first Linear,
//     W1 ((A + I) h) = (A + I) (h W1^T),
// so the first half is the GCN kernel's "z = x W, out = A z" with unit coefficients and no BatchNorm in front, and the
// layer's only cross-graph dependency is the BatchNorm between the two Linear layers: two kernels per layer and direction
// (k_ggin_fwd<1> | statistics | k_ggin_fwd<2>;  k_ggin_bwd<2> | BatchNorm-backward sums | k_ggin_bwd<1>) instead of the
// five / nine launches of the node-level chain (k_espmm, two k_gemm, k_gin_rows passes).
// Linear weights are [out][in] (GCNConv's are [in][out]): the W slice is transposed while it is staged, and the weight-
// gradient slab is produced as [out][in] tiles.
#pragma once
#include "engine_gconv_bwd.hpp"

namespace cal {

struct GginFwdArgs {
    const float* x;          // [N,K] input rows: h_{i-1} (PART 1) / t1 (PART 2)
    const float* W;          // Linear weight [H][K]
    const float* bias;       // [H]
    BNRef bn;                // PART 2: the layer's BatchNorm, applied to x before the ReLU
    float* out;              // [N,H]: t1 (PART 1) / h_i (PART 2)
    Acc st_sum, st_sq;       // PART 1: column statistics of out, one partial row per workgroup unit
};

// PART 1: out = (A + I) (x W^T) + b          (+ column statistics for the BatchNorm that follows)
// PART 2: out = relu(relu(BN(x)) W^T + b)
// grid (units, H / 64), 512 threads: the eight waves split the reduction range of the first product in two halves
// (partial z tiles combined through LDS), like k_gconv_fwd<., 64, 512>.
template <int PART>
__global__ void __launch_bounds__(512, 2) k_ggin_fwd(const CSR g, const int* __restrict__ gptr, const int* __restrict__ eptr,
                                                     const GginFwdArgs a, int H, int K, int* __restrict__ status) {
    constexpr int T = 64, NT = 512, ECAP = gc_edge_cap(T), LDT = T + 4;
    constexpr int CU = ECAP / NT, RPP = NT / 8;
    __shared__ __attribute__((aligned(16))) float As[T * GC_LDX];          // x' rows [row][k]; PART 1 later: adjacency block [i][j] (stride LDT)
    __shared__ __attribute__((aligned(16))) float Bs[GC_K * GC_LDB];       // W^T slice [k][col]; later the z tile [col][row] (stride LDT)
    __shared__ float sc_s[GC_K], sh_s[GC_K];
    __shared__ int ptr_s[T + 4];
    __shared__ short en[ECAP];
    __shared__ signed char er[ECAP];
    __shared__ double red[4][2][32];
    warm_kernargs<sizeof(CSR) + 2 * sizeof(void*) + sizeof(GginFwdArgs) + 32>();
    const int b = blockIdx.x, n0 = blockIdx.y * GC_N, t = threadIdx.x;
    // W^T slice: lane -> (output column t % 64, four consecutive k): requested before the graph's extents are known
    float4 vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = t + u * NT, col = idx & 63, k4 = min(idx >> 6, (K >> 2) - 1);
        vb[u] = *reinterpret_cast<const float4*>(a.W + (size_t)(n0 + col) * K + 4 * k4);
    }
    const int g0 = gptr[b], rows = gptr[b + 1] - g0, e0 = eptr[b], ne = eptr[b + 1] - e0;
    const bool want = PART == 1 && a.st_sum.on();
    if (rows <= 0) {
        if (PART == 2 && a.bn.update && blockIdx.x == 0 && blockIdx.y == 0 && t < K) { const BNRaw r0 = bn_raw_load_st(a.bn, t); bn_raw_update_running(a.bn, r0, t); }
        if (want && t < GC_N) { a.st_sum.add(n0 + t, 0.0); a.st_sq.add(n0 + t, 0.0); }
        return;
    }
    if (rows > T || ne > ECAP || ne < 0) { if (t == 0) atomicOr(status, 8); return; }
    const int rowsP = (rows + 31) & ~31, R = rowsP >> 5, nkc = K >> 5, RB = (rowsP + RPP - 1) / RPP;
    constexpr int UA = T * 32 / NT;
    float4 va[UA];
    {
        int kc = 0, rr = 0;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const bool ok = kc < nkc;
            const int r = min((ok ? rr : 0) * RPP + (t >> 3), rows - 1), k = ((ok ? kc : 0) << 5) + ((t & 7) << 2);
            va[u] = *reinterpret_cast<const float4*>(a.x + (size_t)(g0 + r) * K + k);
            if (++rr == RB) { rr = 0; ++kc; }
        }
    }
    int pv = 0, nv[CU];
    const int slot_hi = max(g.nnz - 1, 0);
    if (PART == 1) {
        pv = g.ptr[g0 + min(t, rows)];
#pragma unroll
        for (int u = 0; u < CU; ++u) nv[u] = g.nbr[min(e0 + max(min(t + u * NT, ne - 1), 0), slot_hi)];
    }
    const int lane = t & 63, li = lane & 31, lk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kh = w >> 2, ct = w & 1, r0 = (w & 3) >> 1;
    float bias = a.bias[n0 + ct * 32 + li];
    BNRawS braws;                                        // (striped reader, engine.hpp: PART 1 of the layer may add into the planes)
    if (PART == 2) braws = bn_raws_load(a.bn, min(t, K - 1));
#pragma unroll
    for (int u = 0; u < UA; ++u) ro_pin(va[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) ro_pin(vb[u]);
    if (PART == 2) bn_raws_pin(braws);
    if (PART == 1) {
#pragma unroll
        for (int u = 0; u < CU; ++u) asm volatile("" : "+v"(nv[u]));
        asm volatile("" : "+v"(pv));
        if (ne <= 0) {
#pragma unroll
            for (int u = 0; u < CU; ++u) nv[u] = g0;
        }
    }
    asm volatile("" : "+v"(bias));
    if (PART == 2 && t < K) {
        const BNRaw braw = bn_raws_sum(a.bn, braws);
        bn_raw_scale_shift(a.bn, braw, sc_s[t], sh_s[t]);
        if (a.bn.update && blockIdx.x == 0 && blockIdx.y == 0) bn_raw_update_running(a.bn, braw, t);
    }
    // ---- stage -----------------------------------------------------------------------------------------------------
    if (PART == 1) {
        if (t <= rows) ptr_s[t] = pv - e0;
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int s = t + u * NT;
            if (s < ne) {
                const int loc = nv[u] - g0;
                const bool inb = loc >= 0 && loc < rows;
                en[s] = (short)(inb ? loc : 0);
                if (!inb) atomicOr(status, 16);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = t + u * NT, col = idx & 63, k4 = idx >> 6;
        if (4 * k4 < K) {
            float* bp = Bs + (4 * k4) * GC_LDB + col;
            bp[0] = vb[u].x; bp[GC_LDB] = vb[u].y; bp[2 * GC_LDB] = vb[u].z; bp[3 * GC_LDB] = vb[u].w;
        }
    }
    __syncthreads();                                     // BN tables, CSR pointers
    if (PART == 1 && t < rows) {
        const int s1 = ptr_s[t + 1];
        for (int s = ptr_s[t]; s < s1; ++s) er[s] = (signed char)t;
    }
    {
        int kc = 0, rr = 0;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            if (kc < nkc) {
                const int r = rr * RPP + (t >> 3), k = (kc << 5) + ((t & 7) << 2);
                float4 v = va[u];
                if (PART == 2)
                    v = make_float4(fmaxf(fmaf(v.x, sc_s[k], sh_s[k]), 0.f), fmaxf(fmaf(v.y, sc_s[k + 1], sh_s[k + 1]), 0.f),
                                    fmaxf(fmaf(v.z, sc_s[k + 2], sh_s[k + 2]), 0.f), fmaxf(fmaf(v.w, sc_s[k + 3], sh_s[k + 3]), 0.f));
                *reinterpret_cast<float4*>(As + r * GC_LDX + k) = v;
            }
            if (++rr == RB) { rr = 0; ++kc; }
        }
    }
    __syncthreads();
    // ---- z tile = x' W^T slice: wave (kh, r0, ct) takes half kh of the reduction range of tile (r0, ct) -----------------
    gc_f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    if (r0 < R) {
        const int kofs = kh * (K >> 1);
        gconv_mma_arow<false, GC_LDB>(As + kofs, Bs + kofs * GC_LDB, K >> 1, r0, ct, li, lk, acc0, acc1);
    }
    __syncthreads();                                     // every wave is done reading both stages
    float* Zt = Bs;                                      // Zt[col * LDT + j]
    float* At = As;                                      // At[i * LDT + j]
    if (r0 < R && kh == 1) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<float4*>(Zt + (ct * 32 + li) * LDT + r0 * 32 + 8 * gq + 4 * lk) =
                make_float4(acc0[4 * gq], acc0[4 * gq + 1], acc0[4 * gq + 2], acc0[4 * gq + 3]);
    }
    if (PART == 1) {
        const int nz4 = (rowsP * LDT) >> 2;
        float4* z4 = reinterpret_cast<float4*>(At);
        for (int idx = t; idx < nz4; idx += NT) z4[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (kh == 0 && r0 < R) {                             // z tile = this wave's half + the partner's
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float4* zp = reinterpret_cast<float4*>(Zt + (ct * 32 + li) * LDT + r0 * 32 + 8 * gq + 4 * lk);
            const float4 p = *zp;
            acc0[4 * gq] += p.x; acc0[4 * gq + 1] += p.y; acc0[4 * gq + 2] += p.z; acc0[4 * gq + 3] += p.w;
            if (PART == 1) *zp = make_float4(acc0[4 * gq], acc0[4 * gq + 1], acc0[4 * gq + 2], acc0[4 * gq + 3]);
        }
    }
    const bool own = kh == 0;
    if (PART == 1) {
        // unit adjacency block A + I: one lane per CSR slot (duplicate edges accumulate), one per self loop
        for (int s = t; s < ne; s += NT) atomicAdd(&At[er[s] * LDT + en[s]], 1.f);
        if (t < rows) atomicAdd(&At[t * LDT + t], 1.f);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
        if (own && r0 < R) gconv_mma_rowk<false, LDT>(At, Zt, rowsP, r0, ct, li, lk, acc0, acc1);
    }
    // ---- epilogue: bias (PART 2: ReLU), store, PART 1: column sums of this unit -----------------------------------------
    float f1[4] = {0.f, 0.f, 0.f, 0.f}, f2[4] = {0.f, 0.f, 0.f, 0.f};
    const int col = n0 + ct * 32 + li;
    asm volatile("" :: "v"(bias));
    if (own && r0 < R) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            float v = acc0[r] + bias;
            if (PART == 2) v = fmaxf(v, 0.f);
            if (row < rows) a.out[(size_t)(g0 + row) * H + col] = v;
            const float vm = row < rows ? v : 0.f;
            f1[r & 3] += vm; f2[r & 3] = fmaf(vm, vm, f2[r & 3]);
        }
    }
    if (PART == 1) {
        double s1 = ((double)f1[0] + (double)f1[1]) + ((double)f1[2] + (double)f1[3]);
        double s2 = ((double)f2[0] + (double)f2[1]) + ((double)f2[2] + (double)f2[3]);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (own && lk == 0) { red[w & 3][0][li] = s1; red[w & 3][1][li] = s2; }
        __syncthreads();
        if (w < 2 && lk == 0 && want) {
            a.st_sum.add(col, red[w][0][li] + red[w + 2][0][li]);
            a.st_sq.add(col, red[w][1][li] + red[w + 2][1][li]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward.  PART 2 (second Linear; runs first): with dz = d h_i masked by h_i > 0 (given, or built from the two partial
// d h_i of the layer above's PART 1 while staging, its column sums = d b2), y = relu(BN(t1)):
//     dW2 [out][in] slab = dz^T y,   dy partial = dz[:, ns] W2[ns, :]  written MASKED by y > 0,
//     (s1, s2) partial rows = (sum m dy, sum m dy x_hat)   -- the BatchNorm-backward sums behind the ReLU
// PART 1 (BatchNorm backward, first Linear, aggregation): dt1 = gamma rstd (dyM - s1 / n - x_hat s2 / n) on the slice
// columns (dyM = the two masked partials), its column sums = d b1,
//     dz = (A + I)^T dt1,   dW1 slab = dz^T h_{i-1},   d h_{i-1} partial = dz[:, ns] W1[ns, :]
// grid (units, H / 64), 512 threads: waves 0-3 the input-gradient product, waves 4-7 the weight-gradient slab.
// ------------------------------------------------------------------------------------------------------------------
struct GginBwdArgs {
    const float* dout;       // PART 2: [N,H] d h_i (already masked), or null: built from dy0 (+ dy1) masked by hmask > 0
    const float* dy0; const float* dy1;      // partials [N,H]: PART 2 (dout null): d h_i from the layer above; PART 1: masked dy of PART 2
    const float* hmask;      // PART 2 (dout null): h_i
    double* bias_parts;      // [units][H] column sums of the dOut built here (PART 2 with dout null: d b2; PART 1: d b1), or null
    const float* x;          // [N,K]: t1 (PART 2) / h_{i-1} (PART 1)
    const float* W;          // [H][K]: W2 (PART 2) / W1 (PART 1)
    BNRef bn;                // the layer's BatchNorm (batch statistics of the forward)
    float* dxp0; float* dxp1; // [N,K] partial input gradients of output-column slice 0 / 1
    float* slab;             // [units][H*K] weight-gradient slabs, Linear layout [out][in]
    double* dot_parts;       // PART 2: [units * H/64][2K] partial rows of (s1, s2)
    double* dacc_sum; double* dacc_prod; int dacc_ss;    // ... or (non-null) into the workgroup's accumulator plane (engine.hpp: stripe_sum)
    const float* t1;         // PART 1: [N,H] pre-BatchNorm activations
    const double* dot_sum; const double* dot_prod;   // PART 1: finalised s1, s2 [H]
};

// acc[q] (q = 0, 1: row tiles a0, a1) += A B over kred with k-major LDS operands A[k * LDA + row], B[k * LDB + col];
// bx transforms the B element of column `col` (the lane's)
template <int LDA, int LDB, class BX>
__device__ __forceinline__ void ggin_mma_bx(const float* a0, const float* a1, const float* b0, int kred, int lk, BX bx, gc_f32x16 (&acc)[2]) {
    float av[2][2][16], bv[2][16];
    auto read_ops = [&](int kb, int s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = kb * 32 + 2 * i + lk;
            av[s][0][i] = a0[k * LDA];
            av[s][1][i] = a1[k * LDA];
            bv[s][i] = bx(b0[k * LDB]);
        }
    };
    auto mul = [&](int s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][0][i], bv[s][i], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][1][i], bv[s][i], acc[1], 0, 0, 0);
        }
    };
    const int nkb = kred / 32;
    read_ops(0, 0);
    for (int kb = 0; kb < nkb; kb += 2) {
        if (kb + 1 < nkb) read_ops(kb + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mul(0);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < nkb) {
            if (kb + 2 < nkb) read_ops(kb + 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            mul(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int PART>
__global__ void __launch_bounds__(GB_NT) k_ggin_bwd(const CSR g, const int* __restrict__ gptr, const int* __restrict__ eptr,
                                                  const GginBwdArgs a, int N, int H, int K, int* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) float Ab[PART == 1 ? GB_T * GB_LDJ : 4];    // PART 1: unit adjacency block Ab[j][i]: dz_i += Ab[j][i] dOut_j
    __shared__ __attribute__((aligned(16))) float Ds[GB_T * GB_LDD];       // dOut slice [j][n]; PART 1 later dz [i][n]
    __shared__ __attribute__((aligned(16))) float Ws[GC_K * GB_LDD];       // W[ns, :] transposed: Ws[k_in][n]
    __shared__ __attribute__((aligned(16))) float Xs[GB_T * GB_LDX];       // PART 2: x_hat rows of t1; PART 1: h_{i-1} rows
    __shared__ float gam_s[GC_K], bet_s[GC_K], mean_s[GC_K], rstd_s[GC_K];
    __shared__ float um_s[GC_N], ur_s[GC_N], ug_s[GC_N], u1_s[GC_N], u2_s[GC_N];     // PART 1: the BatchNorm on this slice's columns
    __shared__ int ptr_s[GB_T + 4];
    __shared__ int en[PART == 1 ? GB_E : 1];
    __shared__ short er[PART == 1 ? GB_E : 1];
    __shared__ float bs_s[GB_NT / 64][16][4];
    warm_kernargs<sizeof(CSR) + 2 * sizeof(void*) + sizeof(GginBwdArgs) + 32>();
    const int b = blockIdx.x, sl = blockIdx.y, ns0 = sl * GC_N, t = threadIdx.x;
    const int g0 = gptr[b], rows = gptr[b + 1] - g0, e0 = eptr[b], ne = eptr[b + 1] - e0;
    const int lane = t & 63, li = lane & 31, lk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    double* parts = PART == 2 ? a.dot_parts + ((size_t)sl * gridDim.x + b) * (2 * K) : nullptr;
    float* slab = a.slab + (size_t)b * K * H;
    const bool given = PART == 2 && a.dout != nullptr;
    if (rows <= 0 || rows > GB_T || ne > GB_E || ne < 0) {
        if (rows > 0 && t == 0) atomicOr(status, 8);
        if (PART == 2 && !a.dacc_sum) for (int i = t; i < 2 * K; i += GB_NT) parts[i] = 0.0;
        if (a.bias_parts && t < GC_N) a.bias_parts[(size_t)b * H + ns0 + t] = 0.0;
        for (int i = t; i < K * GC_N; i += GB_NT) slab[(size_t)(ns0 + i / K) * K + i % K] = 0.f;
        return;
    }
    const int rowsP = (rows + 31) & ~31, R = rowsP >> 5, K4 = K >> 2;
    // ---- every global load, issued before the first wait ---------------------------------------------------------------
    RoBatch<float4, 2> bd, bd1, bh;                      // rows x 16 float4 tiles of this slice's columns
    RoBatch<float4, 4> bx;                               // x rows: rows x K/4
    float4 vw[4];                                        // W[ns0 + n][4 k4 ..]: lane -> (n = idx % 64, k4 = idx / 64)
    {
        const float* d0 = given ? a.dout : a.dy0;
        const float* d1 = (!given && a.dy1) ? a.dy1 : d0;
        const float* hm = PART == 1 ? a.t1 : (given ? d0 : a.hmask);
        ro_issue<GB_NT>(bd, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(d0 + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
        ro_issue<GB_NT>(bd1, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(d1 + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
        ro_issue<GB_NT>(bh, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(hm + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
    }
    ro_issue<GB_NT>(bx, rows, K4, [&](int i, int k4) { return *reinterpret_cast<const float4*>(a.x + (size_t)(g0 + i) * K + 4 * k4); });
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = t + u * GB_NT, n = idx & 63, k4 = min(idx >> 6, K4 - 1);
        vw[u] = *reinterpret_cast<const float4*>(a.W + (size_t)(ns0 + n) * K + 4 * k4);
    }
    int pv = 0, pn = 0, nv[2] = {0, 0};
    if (PART == 1) {
        pv = g.ptr[g0 + min(t, rows)];
        pn = g.ptr[g0 + min(t + 1, rows)];
        const int slot_hi = max(g.nnz - 1, 0);
#pragma unroll
        for (int u = 0; u < 2; ++u) nv[u] = g.nbr[min(e0 + max(min(t + u * GB_NT, ne - 1), 0), slot_hi)];
    }
    BNRawS braws = bn_raws_load(a.bn, PART == 2 ? min(t, K - 1) : ns0 + (t & (GC_N - 1)));      // (striped readers, engine.hpp)
    StripeVal ud1s, ud2s;
    if (PART == 1) { ud1s = stripe_load(a.dot_sum, ns0 + (t & (GC_N - 1)), a.bn.ss); ud2s = stripe_load(a.dot_prod, ns0 + (t & (GC_N - 1)), a.bn.ss); }
    bn_raws_pin(braws);
    if (PART == 1) {
        stripe_pin(ud1s); stripe_pin(ud2s);
        asm volatile("" : "+v"(pv), "+v"(pn), "+v"(nv[0]), "+v"(nv[1]));
        if (ne <= 0) { nv[0] = g0; nv[1] = g0; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) ro_pin(vw[u]);
    if (PART == 2 && t < K) {
        float m1, r1;
        const BNRaw braw = bn_raws_sum(a.bn, braws);
        bn_raw_mean_rstd(a.bn, braw, m1, r1);
        mean_s[t] = m1; rstd_s[t] = r1; gam_s[t] = braw.g; bet_s[t] = braw.b;
    }
    if (PART == 1 && t < GC_N) {
        float m1, r1;
        const BNRaw braw = bn_raws_sum(a.bn, braws);
        const double ud1 = stripe_total(ud1s, a.bn.ss), ud2 = stripe_total(ud2s, a.bn.ss);
        bn_raw_mean_rstd(a.bn, braw, m1, r1);
        um_s[t] = m1; ur_s[t] = r1; ug_s[t] = braw.g * r1;
        u1_s[t] = (float)(ud1 * (double)a.bn.inv_n);
        u2_s[t] = (float)(ud2 * (double)a.bn.inv_n);
    }
    if (PART == 1) {
        for (int i = t; i < (rowsP * GB_LDJ + 3) / 4; i += GB_NT) reinterpret_cast<float4*>(Ab)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t <= rows) ptr_s[t] = pv - e0;
        if (t < rows) for (int s = pv - e0; s < pn - e0; ++s) er[s] = (short)t;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = t + u * GB_NT;
            if (s < ne) {
                const int loc = nv[u] - g0;
                const bool inb = loc >= 0 && loc < rows;
                en[s] = inb ? loc : 0;
                if (!inb) atomicOr(status, 16);
            }
        }
    }
    // W[ns, :]^T: Ws[k][n], lanes along n (conflict-free scalar stores)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = t + u * GB_NT, n = idx & 63, k4 = idx >> 6;
        if (k4 < K4) {
            float* wp = Ws + (4 * k4) * GB_LDD + n;
            wp[0] = vw[u].x; wp[GB_LDD] = vw[u].y; wp[2 * GB_LDD] = vw[u].z; wp[3 * GB_LDD] = vw[u].w;
        }
    }
    __syncthreads();                                     // BatchNorm constants, CSR
    ro_commit<GB_NT>(bx, rows, K4, [&](int i, int k4, float4 v) {
        const int k = 4 * k4;
        if (PART == 2) {
            v.x = (v.x - mean_s[k]) * rstd_s[k]; v.y = (v.y - mean_s[k + 1]) * rstd_s[k + 1];
            v.z = (v.z - mean_s[k + 2]) * rstd_s[k + 2]; v.w = (v.w - mean_s[k + 3]) * rstd_s[k + 3];
        }
        *reinterpret_cast<float4*>(Xs + i * GB_LDX + k) = v;
    });
    {
        // dOut slice: lane t holds column group t % 16 of rows t / 16 (+ 32): column sums stay in registers
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) { ro_pin(bd.v[u]); ro_pin(bd1.v[u]); ro_pin(bh.v[u]); }
        const int c = 4 * (t & 15);
        const bool two = !given && a.dy1 != nullptr;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = (t >> 4) + u * (GB_NT / 16);
            if (j < rows) {
                const float4 v0 = bd.v[u], v1 = bd1.v[u], hv = bh.v[u];
                const float d[4] = {v0.x + (two ? v1.x : 0.f), v0.y + (two ? v1.y : 0.f), v0.z + (two ? v1.z : 0.f), v0.w + (two ? v1.w : 0.f)};
                const float hh[4] = {hv.x, hv.y, hv.z, hv.w};
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (PART == 2) o[q] = (given || hh[q] > 0.f) ? d[q] : 0.f;
                    else {
                        const float xn = (hh[q] - um_s[c + q]) * ur_s[c + q];
                        o[q] = ug_s[c + q] * (d[q] - u1_s[c + q] - xn * u2_s[c + q]);
                    }
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
    for (int i = t; i < (rowsP - rows) * GB_LDD; i += GB_NT) Ds[rows * GB_LDD + i] = 0.f;
    for (int i = t; i < (rowsP - rows) * GB_LDX; i += GB_NT) Xs[rows * GB_LDX + i] = 0.f;
    if (PART == 1) {
        for (int s = t; s < ne; s += GB_NT) atomicAdd(&Ab[er[s] * GB_LDJ + en[s]], 1.f);
        if (t < rows) atomicAdd(&Ab[t * GB_LDJ + t], 1.f);
    }
    __syncthreads();
    if (a.bias_parts && t < GC_N) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < GB_NT / 64; ++k) tot += (double)bs_s[k][t >> 2][t & 3];
        a.bias_parts[(size_t)b * H + ns0 + t] = tot;
    }
    gc_f32x16 acc[2];
    if (PART == 1) {
        // ---- P1: dz[:, ns] = (A + I)^T dOut[:, ns] ------------------------------------------------------------------
        auto ident = [](float v) { return v; };
        const int rt = w >> 1, ct = w & 1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
        if (w < 4 && rt < R) gb_mma<1, 1, GB_LDJ, GB_LDD>(Ab + rt * 32 + li, nullptr, Ds + ct * 32 + li, nullptr, rowsP, lk, ident, acc);
        __syncthreads();                                 // every wave is done reading dOut
        if (w < 4 && rt < R) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Ds[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * GB_LDD + ct * 32 + li] = acc[0][r];
        }
        __syncthreads();
    }
    // ---- P2: partial input gradient [:, all K] = dz[:, ns] W[ns, :] --------------------------------------------------------
    if (w < 4 && w * 32 < K) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
        if (R == 2) gb_mma_rowk2<true>(Ds + li * GB_LDD, Ds + (32 + li) * GB_LDD, Ws + (w * 32 + li) * GB_LDD, GC_N, lk, acc[0], acc[1]);
        else gb_mma_rowk2<false>(Ds + li * GB_LDD, nullptr, Ws + (w * 32 + li) * GB_LDD, GC_N, lk, acc[0], acc[1]);
        const int k = w * 32 + li;
        float* dxp = sl ? a.dxp1 : a.dxp0;
        if (PART == 2) {
            float xh[2][16];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) xh[q][r] = Xs[(q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * GB_LDX + k];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) { asm volatile("" : "+v"(xh[q][r])); if (q >= R) xh[q][r] = 0.f; }
            const float gam = gam_s[k], bet = bet_s[k];
            float f1[4] = {0.f, 0.f, 0.f, 0.f}, f2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    const float v = fmaf(xh[q][r], gam, bet) > 0.f ? acc[q][r] : 0.f;        // behind the ReLU of y = relu(BN(t1))
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
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (i < rows) dxp[(size_t)(g0 + i) * K + k] = acc[q][r];
                }
            }
        }
    }
    // ---- P3: weight-gradient slab rows ns (Linear layout [out][in]) = dz[:, ns]^T x'   (reduction over the unit's nodes) -----
    if (w >= 4 && (w - 4) * 32 < K) {
        const int wq = w - 4, k = wq * 32 + li;
        const float gam = PART == 2 ? gam_s[k] : 1.f, bet = PART == 2 ? bet_s[k] : 0.f;
        auto bxf = [&](float v) { return PART == 2 ? fmaxf(fmaf(v, gam, bet), 0.f) : v; };
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
        ggin_mma_bx<GB_LDD, GB_LDX>(Ds + li, Ds + 32 + li, Xs + wq * 32 + li, rowsP, lk, bxf, acc);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                slab[(size_t)(ns0 + n) * K + k] = acc[q][r];
            }
    }
}

}  // namespace cal
