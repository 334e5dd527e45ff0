// Per-graph fused backward of a GCN convolution of the step engine (the transpose of engine_gconv.hpp):
// with dOut = gradient w.r.t. the convolution's pre-activation output, x' = BN(rs * x) its (normalised)
// input and z = x' W,
//     dz  = A_hat^T dOut                       (gcn_conv.py:92-104 backward)
//     dX' = dz W^T  (+ the two BatchNorm-backward column sums  sum dX',  sum dX' * x_hat)
//     dW  = x'^T dz
// as ONE kernel instead of transposed aggregation -> dX GEMM + dW GEMM.  A workgroup owns one graph and a
// 64-column slice `ns` of the OUTPUT features: it aggregates only its slice of dz (dense adjacency block
// on MFMA, like the forward), multiplies it with W[:, ns]^T into a PARTIAL dX' over all K input columns
// (the H/64 slices are summed by the consumer: k_bn_bwd / k_att_bwd read both partials) and with x'^T into
// the columns `ns` of this graph's dW slab (the B slabs are summed by k_finish, as split-K slabs were).
// dz never goes to HBM and neither product re-reads it.
//
//   grid (B graphs, H / 64 slices, branches), 512 threads; graphs of at most 64 nodes / 1024 stored edges
//   (cal_engine_set_graph_bounds), K = H in {64, 128}.  LDS ~115-135 KB: one workgroup per CU, so it brings its
//   own latency hiding: 8 waves -- waves 0-3 run the dX' product and its epilogue while waves 4-7 run the
//   dW product (both only need dz), and twice as many loads are in flight while staging.
#pragma once
#include "engine_gconv.hpp"

namespace cal {

constexpr int GB_NT = 512;                // threads per workgroup
constexpr int GB_T = 64;                  // nodes per graph
constexpr int GB_E = 1024;                // stored edges per graph
constexpr int GB_LDJ = GB_T + 1;          // k-major tiles indexed by a node: adjacency block, dz^T
constexpr int GB_LDD = GC_N + 4;          // row-major [node][64 output columns]: dOut slice, dz
constexpr int GB_LDW = GC_K + 1;          // W slice transposed: Wt[n][k_in]
constexpr int GB_LDX = GC_K + 4;          // row-major [node][K]: normalised input rows

struct GconvBwdBranch {
    const float* dout;       // [N,H]
    const float* x;          // [N,K] raw layer input
    const float* W;          // [K,H]
    const float* ew;         // per-edge weight in edge-id order, or null
    const float* dis;        // [N]
    const float* rs;         // per-row scale of x, or null
    int rs_stride;
    BNRef bn;                // BatchNorm applied to rs * x (batch statistics of the forward)
    float* dxp0; float* dxp1; // [N,K] partial dX' of output-column slice 0 / 1
    float* slab;             // [B][K,H] per-graph dW
    double* dot_parts;       // [B * H/64][2K]: (sum dX', sum dX' * x_hat) partial rows
    double* dacc_sum; double* dacc_prod; int dacc_ss;    // or (non-null): added atomically into the workgroup's plane of the site's
                                                         // NSTRIPE accumulator planes (engine.hpp: stripe_sum; no finishing launch)
    const float* coef_in;    // edge coefficients dis_j * w_e in CSR-slot order, written by the forward kernel, or null
    // UP variant: dOut is not materialised.  It is the BatchNorm-backward (+ ReLU mask) of the layer ABOVE,
    //     dOut = relu'(y) * gamma_u rstd_u (dY - m1_u - y_hat m2_u),   dY = dy0 + dy1,
    // computed while the slice is staged (what k_bn_bwd would have written and this kernel read back), and its
    // per-graph column sums (the bias gradient of this convolution) go to bias_parts [B][H].
    const float* dy0; const float* dy1;      // partials of the upper layer's dX' (dy1 null when H == 64)
    const float* y;                          // [N,H] this convolution's output after ReLU = the upper BatchNorm's input
    BNRef ubn;                               // the upper BatchNorm
    const double* udot_sum; const double* udot_prod;
    double* bias_parts;
    // POOL variant (the two weighted convolutions under global_add_pool, model.py:112-116): dOut is not materialised
    // either -- dOut[v] = relu'(out[v]) * g_b with g_b = gp0[b] (+ gp1[pb], pb = iperm[b] or b) the gradient of graph b's
    // pooled row (y = the convolution's output, bias_parts as above) -- and the kernel also emits this slice's part of
    //     gn[e] = <dOut[col_e], z[row_e]>,   gself[v] = <dOut[v], z[v]>
    // (the SDDMM behind the edge-weight gradients, gcn_conv.py:63-70,97 differentiated) from LDS while P1 runs.
    const float* gp0; const float* gp1; const int* iperm;
    const float* z;          // [N,H] x' W of the forward
    float* gn; float* gself; // this branch's [E] / [N] partials of slice 0; slice 1 is gn_stride / gself_stride further
    size_t gn_stride, gself_stride;
    int gn_slot;             // gn is written in CSR-slot order (gn[eptr[b] + s], for the per-graph attention backward) instead of edge-id order
    // packed batch (TILED instantiation of the POOL variant, cal_engine_set_tiles): the unit is a tile of the consecutive
    // graphs [tile_gptr[b], tile_gptr[b + 1]); gp0 / gp1 / iperm are per GRAPH, batch [N] names the graph of every row
    const int64_t* batch;
    const int64_t* tile_gptr;
};

struct GconvBwdBranch2 { GconvBwdBranch b[2]; };

// acc[0] (+acc[1]) += A B over kred (multiple of 32) with k-major LDS operands A[k*LDA + row], B[k*LDB + col];
// NA == 2: two row tiles (a0, a1) against b0; NB == 2: a0 against two column tiles (b0, b1).
// ax(v, kstep) transforms an A element (identity or the BatchNorm affine of the lane's row).
template <int NA, int NB, int LDA, int LDB, class AX>
__device__ __forceinline__ void gb_mma(const float* a0, const float* a1, const float* b0, const float* b1, int kred, int lk,
                                       AX ax, gc_f32x16 (&acc)[2]) {
    static_assert(NA == 1 || NB == 1, "one of the operands is shared");
    float av[2][2][16], bv[2][2][16];
    auto read_ops = [&](int kb, int s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = kb * 32 + 2 * i + lk;
            av[s][0][i] = ax(a0[k * LDA]);
            if (NA == 2) av[s][1][i] = ax(a1[k * LDA]);
            bv[s][0][i] = b0[k * LDB];
            if (NB == 2) bv[s][1][i] = b1[k * LDB];
        }
    };
    auto mul = [&](int s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][0][i], bv[s][0][i], acc[0], 0, 0, 0);
            if (NA == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][1][i], bv[s][0][i], acc[1], 0, 0, 0);
            if (NB == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][0][i], bv[s][1][i], acc[1], 0, 0, 0);
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

// One 32 x 32 tile from operands stored ROW-MAJOR IN k (A[row * LDA + k], B[col * LDB + k], strides = 4 mod 32 floats: 16 B
// reads of 32 consecutive rows are bank-conflict free, 4 B reads would be 4-way conflicts): lane (li, lk) takes the four
// consecutive k of every eight and feeds them to four MFMA steps -- any bijection of k onto (step, lk) is a valid reduction
// order as long as A and B share it.  a_row / b_row point at this lane's row; kred % 32 == 0.
__device__ __forceinline__ void gb_mma_rowk(const float* a_row, const float* b_row, int kred, int lk, gc_f32x16& acc) {
    for (int k0 = 0; k0 < kred; k0 += 32) {
        float4 av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + 8 * i + 4 * lk;
            av[i] = *reinterpret_cast<const float4*>(a_row + k);
            bv[i] = *reinterpret_cast<const float4*>(b_row + k);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[i].w, acc, 0, 0, 0);
        }
    }
}

// the same with two row tiles of A against one tile of B (acc0: rows of a0_row, acc1: rows of a1_row); TWO = false: only acc0
template <bool TWO>
__device__ __forceinline__ void gb_mma_rowk2(const float* a0_row, const float* a1_row, const float* b_row, int kred, int lk,
                                             gc_f32x16& acc0, gc_f32x16& acc1) {
    for (int k0 = 0; k0 < kred; k0 += 32) {
        float4 av[4], aw[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + 8 * i + 4 * lk;
            av[i] = *reinterpret_cast<const float4*>(a0_row + k);
            if (TWO) aw[i] = *reinterpret_cast<const float4*>(a1_row + k);
            bv[i] = *reinterpret_cast<const float4*>(b_row + k);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a[4] = {av[i].x, av[i].y, av[i].z, av[i].w}, b[4] = {bv[i].x, bv[i].y, bv[i].z, bv[i].w};
            const float a2[4] = {TWO ? aw[i].x : 0.f, TWO ? aw[i].y : 0.f, TWO ? aw[i].z : 0.f, TWO ? aw[i].w : 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc0, 0, 0, 0);
                if (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j], b[j], acc1, 0, 0, 0);
            }
        }
    }
}

// LEAN (MODE 0 / 1, the single-branch launches; round 5): under 80 KB of LDS and 128 registers, so TWO workgroups share a CU
// and a launch of more workgroups than CUs (a packed batch: 240 tiles x 2 slices at NCI1-like batches of 512 graphs) stops
// running as two rounds of latency chains -- the W slice is not staged (waves 0-3 read their rows of it straight from L2 as
// MFMA operands in P2: 16-byte reads of a 64 KB matrix every workgroup shares), the x_hat rows have no padding (no access to
// them is strided by a row), CSR neighbours are 16-bit.  The POOL variant (MODE 2) also holds the z rows for the SDDMM: 140 KB,
// one workgroup per CU.
// Chosen per launch (gconv_bwd in engine.hip): a launch of at most one workgroup per CU keeps the staged W slice -- with one
// workgroup on a CU the two L2 round trips of P2's operand reads are exposed (config 2: 0.2386 -> 0.2418 ms with LEAN everywhere).
template <bool RS, int MODE, bool TILED = false, bool LEAN = false>      // MODE 0: dOut given; 1: UP (from the upper layer's partials); 2: POOL (+ gn / gself)
__global__ void __launch_bounds__(GB_NT, (LEAN ? 4 : 1)) k_gconv_bwd(const CSR g, const int* __restrict__ gptr, const int* __restrict__ eptr,
                                                   const GconvBwdBranch2 bb, float loop_w, int N, int H,
                                                   int K, int* __restrict__ status) {
    constexpr int LDX = LEAN ? GC_K : GB_LDX;
    __shared__ __attribute__((aligned(16))) float Ab[GB_T * GB_LDJ];       // adjacency block Ab[j][i]: dz_i += Ab[j][i] dOut_j
    __shared__ __attribute__((aligned(16))) float Ds[GB_T * GB_LDD];       // dOut slice [j][n]; later dz [i][n]
    __shared__ __attribute__((aligned(16))) float Ws[LEAN ? 4 : GC_K * GB_LDD];   // POOL: W[:, ns] as loaded: Ws[k_in][n] (row-major in n, 16 B operand reads)
    __shared__ __attribute__((aligned(16))) float Xs[GB_T * LDX];          // x_hat rows [i][k_in] (normalised, no affine)
    __shared__ float mean_s[GC_K], rstd_s[GC_K], gam_s[GC_K], bet_s[GC_K];
    __shared__ int ptr_s[GB_T + 4];
    __shared__ float dis_s[GB_T], rs_s[GB_T];
    __shared__ unsigned char en[GB_E];                   // (local node index < 64)
    __shared__ float ec[GB_E];
    __shared__ float um_s[MODE == 1 ? GC_N : 1], ur_s[MODE == 1 ? GC_N : 1], ug_s[MODE == 1 ? GC_N : 1], u1_s[MODE == 1 ? GC_N : 1], u2_s[MODE == 1 ? GC_N : 1];     // UP: upper BatchNorm, this slice's columns
    __shared__ float bs_s[GB_NT / 64][16][4];
    __shared__ __attribute__((aligned(16))) float Zr_own[(MODE == 2 && !LEAN) ? GB_T * GB_LDD : 4];       // POOL: z slice rows [j][n]
    // LEAN POOL: the z rows live in the x_hat stage until P1 and the SDDMM are done with them; x_hat is committed only then
    // (its registers wait through P1) -- the two are never needed at the same time
    float* const Zr = (MODE == 2 && LEAN) ? Xs : Zr_own;
    __shared__ float gv_s[TILED ? GC_TILE_GRAPHS * GC_N : GC_N];   // POOL: gradient of this graph's pooled row (TILED: of every graph of the tile), slice columns
    __shared__ unsigned char bg_s[TILED ? GB_T : 4];     // TILED: graph (inside the tile) of every row
    __shared__ int ee[(MODE == 2 && !LEAN) ? GB_E : 1];  // POOL: edge id of CSR slot s (LEAN: gn goes out in slot order only, the caller vouches for gn_slot)
    __shared__ unsigned char er[GB_E];                   // destination row of CSR slot s (< 64)
    constexpr bool UP = MODE == 1, POOL = MODE == 2;
    static_assert(!TILED || MODE == 2, "only the POOL variant looks at the graphs inside a tile");
    BLK_CLK(0);
    warm_kernargs<sizeof(CSR) + 2 * sizeof(void*) + sizeof(GconvBwdBranch2) + 32>();
    const GconvBwdBranch& br = bb.b[blockIdx.z];         // indexed in the kernel-argument segment (see k_gconv_fwd)
    const int b = blockIdx.x, sl = blockIdx.y, ns0 = sl * GC_N, t = threadIdx.x;
    const int g0 = gptr[b], rows = gptr[b + 1] - g0, e0 = eptr[b], ne = eptr[b + 1] - e0;
    const int pb = (MODE == 2 && !TILED && br.iperm) ? br.iperm[b] : b;         // row of the second pooled-gradient partial (scalar load, with the extents)
    const int tg0 = TILED ? (int)br.tile_gptr[b] : b, ng = TILED ? (int)br.tile_gptr[b + 1] - tg0 : 1;
    const int lane = t & 63, li = lane & 31, lk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    double* parts = br.dot_parts + ((size_t)sl * gridDim.x + b) * (2 * K);
    float* slab = br.slab + (size_t)b * K * H;
    if (rows <= 0 || rows > GB_T || ne > GB_E || ne < 0 || (TILED && (ng < 1 || ng > GC_TILE_GRAPHS))) {
        // empty graph (or a violated bound, flagged): its partial row and its slab slice must still exist
        if (rows > 0 && t == 0) atomicOr(status, 8);
        if (!br.dacc_sum) for (int i = t; i < 2 * K; i += GB_NT) parts[i] = 0.0;
        if ((UP || POOL) && t < GC_N) br.bias_parts[(size_t)b * H + ns0 + t] = 0.0;
        for (int i = t; i < K * GC_N; i += GB_NT) slab[(size_t)(i / GC_N) * H + ns0 + i % GC_N] = 0.f;
        return;
    }
    const bool hasw = br.ew != nullptr;
    const int rowsP = (rows + 31) & ~31, R = rowsP >> 5, K4 = K >> 2;
    // ---- every global load of the kernel, issued before the first wait ------------------------------------------
    RoBatch<float4, 2> bd, bd1, by;                      // dOut[g0 + j][ns0 + 4 n4 ..]: rows x 16 float4 (UP: dy0, dy1, y)
    RoBatch<float4, 4> bx, bw;                           // x[g0 + i][4 k4 ..]: rows x K/4;  W[k_in][ns0 + 4 n4 ..]: K x 16
    RoBatch<float4, 2> bz;                               // POOL: z[g0 + j][ns0 + 4 n4 ..]
    float gv = 0.f, gv1 = 0.f;
    int pbq = 0;
    long long bgv = 0;
    if (POOL) {
        ro_issue<GB_NT>(by, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(br.y + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
        ro_issue<GB_NT>(bz, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(br.z + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
        if (!TILED) {   // gradient of this graph's pooled row, slice columns: both partials unconditionally (gp1 absent: gp0 twice, weight 0)
            const float* gp1 = br.gp1 ? br.gp1 : br.gp0;
            gv = br.gp0[(size_t)b * H + ns0 + (t & (GC_N - 1))];
            gv1 = gp1[(size_t)pb * H + ns0 + (t & (GC_N - 1))];
        } else {        // lane (q = t / 64, column t % 64): graph tg0 + q of the tile; the permuted row's index is a load of its own
            const int gq = tg0 + min(t >> 6, ng - 1);
            gv = br.gp0[(size_t)gq * H + ns0 + (t & (GC_N - 1))];
            pbq = br.iperm ? br.iperm[gq] : gq;
            bgv = br.batch[g0 + min(t, rows - 1)];
        }
    } else {
        const float* d0 = UP ? br.dy0 : br.dout;
        ro_issue<GB_NT>(bd, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(d0 + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
        if (UP) {
            const float* d1 = br.dy1 ? br.dy1 : br.dy0;
            ro_issue<GB_NT>(bd1, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(d1 + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
            ro_issue<GB_NT>(by, rows, 16, [&](int j, int n4) { return *reinterpret_cast<const float4*>(br.y + (size_t)(g0 + j) * H + ns0 + 4 * n4); });
        }
    }
    ro_issue<GB_NT>(bx, rows, K4, [&](int i, int k4) { return *reinterpret_cast<const float4*>(br.x + (size_t)(g0 + i) * K + 4 * k4); });
    if (!LEAN) ro_issue<GB_NT>(bw, K, 16, [&](int k, int n4) { return *reinterpret_cast<const float4*>(br.W + (size_t)k * H + ns0 + 4 * n4); });
    const int pv = g.ptr[g0 + min(t, rows)];
    const int pn = g.ptr[g0 + min(t + 1, rows)];
    const float dv = br.dis[g0 + min(t, rows - 1)];
    const float rv = RS ? br.rs[(size_t)(g0 + min(t, rows - 1)) * br.rs_stride] : 1.f;
    // CSR slots, coefficients and the BatchNorm constants: unconditional loads on clamped indices / substituted pointers,
    // pinned below (BNRaw in engine.hpp: as guarded blocks these were up to ten serial round trips behind the tile loads)
    int nv[2], ev[2];
    const int slot_hi = max(g.nnz - 1, 0);
    const float* coefp = br.coef_in ? br.coef_in : br.dis;
    const int coef_hi = br.coef_in ? slot_hi : 0;
    float cin[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int s = min(e0 + max(min(t + u * GB_NT, ne - 1), 0), slot_hi);
        nv[u] = g.nbr[s];
        ev[u] = g.eid[s];
        cin[u] = coefp[min(s, coef_hi)];
    }
    // (striped readers, engine.hpp: the producers may be per-graph kernels.  Lanes 0 .. K-1 need this layer's BatchNorm, lanes
    //  256 .. 319 the upper one's constants of this slice's 64 columns: ONE register set, the pointers chosen per lane)
    const bool ulane = UP && t >= 256;
    BNRawS braws = UP ? bn_raws_load2(br.bn, min(t, K - 1), br.ubn, ns0 + (t & (GC_N - 1)), ulane) : bn_raws_load(br.bn, min(t, K - 1));
    StripeVal ud1s, ud2s;
    if (UP) {
        const int c = ns0 + (t & (GC_N - 1));
        ud1s = stripe_load(br.udot_sum, c, br.ubn.ss); ud2s = stripe_load(br.udot_prod, c, br.ubn.ss);
    }
    bn_raws_pin(braws);
    if (UP) { stripe_pin(ud1s); stripe_pin(ud2s); }
#pragma unroll
    for (int u = 0; u < 2; ++u) asm volatile("" : "+v"(nv[u]), "+v"(ev[u]), "+v"(cin[u]));
    if (POOL && !TILED) { asm volatile("" : "+v"(gv), "+v"(gv1)); gv += br.gp1 ? gv1 : 0.f; }
    if (TILED) {                                         // second round: the permuted graph's pooled-gradient row
        asm volatile("" : "+v"(gv), "+v"(pbq), "+v"(bgv));
        const float* gp1 = br.gp1 ? br.gp1 : br.gp0;
        gv1 = gp1[(size_t)pbq * H + ns0 + (t & (GC_N - 1))];
    }
    if (ne <= 0) { nv[0] = g0; nv[1] = g0; ev[0] = 0; ev[1] = 0; }   // no slot of this graph exists: the clamped loads fetched no index
    if (UP && t >= 256 && t < 256 + GC_N) {
        float m1, r1;
        const BNRaw uraw = bn_raws_sum(br.ubn, braws);
        const double ud1 = stripe_total(ud1s, br.ubn.ss), ud2 = stripe_total(ud2s, br.ubn.ss);
        bn_raw_mean_rstd(br.ubn, uraw, m1, r1);
        um_s[t - 256] = m1; ur_s[t - 256] = r1;
        ug_s[t - 256] = uraw.g * r1;
        u1_s[t - 256] = (float)(ud1 * (double)br.ubn.inv_n);
        u2_s[t - 256] = (float)(ud2 * (double)br.ubn.inv_n);
    }
    if (t < K) {
        float m1, r1;
        const BNRaw braw = bn_raws_sum(br.bn, braws);
        bn_raw_mean_rstd(br.bn, braw, m1, r1);
        mean_s[t] = m1; rstd_s[t] = r1;
        gam_s[t] = braw.g;
        bet_s[t] = braw.b;
    }
    for (int i = t; i < (rowsP * GB_LDJ + 3) / 4; i += GB_NT) reinterpret_cast<float4*>(Ab)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float cv[2];
    if (br.coef_in) {
#pragma unroll
        for (int u = 0; u < 2; ++u) cv[u] = cin[u];
    } else {
        const float* ewp = hasw ? br.ew : br.dis;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float c = br.dis[nv[u]];
            const float wl = ewp[hasw ? ev[u] : 0];
            cv[u] = hasw ? c * wl : c;
        }
    }
    // ---- stage everything in LDS -----------------------------------------------------------------------------------
    if (t <= rows) ptr_s[t] = pv - e0;
    if (t < rows) {
        dis_s[t] = dv; rs_s[t] = rv;
        for (int s = pv - e0; s < pn - e0; ++s) er[s] = (unsigned char)t;      // destination row of every slot (stores only)
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int s = t + u * GB_NT;
        if (s < ne) {
            const int loc = nv[u] - g0;
            const bool inb = loc >= 0 && loc < rows;
            en[s] = (unsigned char)(inb ? loc : 0); ec[s] = inb ? cv[u] : 0.f;
            if (POOL && !LEAN) ee[s] = ev[u];
            if (!inb) atomicOr(status, 16);
        }
    }
    if (POOL && !TILED && t < GC_N) gv_s[t] = gv;
    if (TILED) {
        if (t < ng * GC_N) gv_s[t] = gv + (br.gp1 ? gv1 : 0.f);
        if (t < rows) bg_s[t] = (unsigned char)min(max((int)(bgv - tg0), 0), ng - 1);
    }
    if (MODE == 0) ro_commit<GB_NT>(bd, rows, 16, [&](int j, int n4, const float4 v) { *reinterpret_cast<float4*>(Ds + j * GB_LDD + 4 * n4) = v; });
    if (!LEAN) ro_commit<GB_NT>(bw, K, 16, [&](int k, int n4, const float4 v) { *reinterpret_cast<float4*>(Ws + k * GB_LDD + 4 * n4) = v; });
    __syncthreads();                                     // per-column BN constants, row scales, zeroed Ab, CSR
    auto commit_x = [&]() {
        ro_commit<GB_NT>(bx, rows, K4, [&](int i, int k4, float4 v) {
            const float s = RS ? rs_s[i] : 1.f;
            const int k = 4 * k4;
            v.x = (v.x * s - mean_s[k]) * rstd_s[k]; v.y = (v.y * s - mean_s[k + 1]) * rstd_s[k + 1];
            v.z = (v.z * s - mean_s[k + 2]) * rstd_s[k + 2]; v.w = (v.w * s - mean_s[k + 3]) * rstd_s[k + 3];
            *reinterpret_cast<float4*>(Xs + i * LDX + k) = v;
        });
    };
    if (!(POOL && LEAN)) commit_x();
    if (POOL) {
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) { ro_pin(by.v[u]); ro_pin(bz.v[u]); }
        const int c = 4 * (t & 15);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = (t >> 4) + u * (GB_NT / 16);
            if (j < rows) {
                const float4 yv = by.v[u];
                const float* gvr = gv_s + (TILED ? bg_s[j] * GC_N : 0);
                const float4 o = make_float4(yv.x > 0.f ? gvr[c] : 0.f, yv.y > 0.f ? gvr[c + 1] : 0.f,
                                             yv.z > 0.f ? gvr[c + 2] : 0.f, yv.w > 0.f ? gvr[c + 3] : 0.f);
                cs[0] += o.x; cs[1] += o.y; cs[2] += o.z; cs[3] += o.w;
                *reinterpret_cast<float4*>(Ds + j * GB_LDD + c) = o;
                *reinterpret_cast<float4*>(Zr + j * GB_LDD + c) = bz.v[u];
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
    if (UP) {
        // dOut slice from the upper layer's partials: lane t always holds column group t % 16 (512 % 16 == 0), so
        // its column sums stay in registers until the cross-lane reduction below
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
        const bool two = br.dy1 != nullptr;
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
    // rows rows .. rowsP of dOut / x_hat: zero (they are reduced over in the products below)
    for (int i = t; i < (rowsP - rows) * GB_LDD; i += GB_NT) Ds[rows * GB_LDD + i] = 0.f;
    for (int i = t; i < (rowsP - rows) * LDX; i += GB_NT) Xs[rows * LDX + i] = 0.f;
    // one lane per CSR slot, then one per self loop (LDS atomics: duplicate edges share an entry) -- a lane per ROW walked
    // a hub's slots as a chain of dependent LDS round trips while the rest of the workgroup waited
    for (int s = t; s < ne; s += GB_NT) {
        const int j = er[s];
        atomicAdd(&Ab[j * GB_LDJ + en[s]], dis_s[j] * ec[s]);
    }
    if (t < rows) atomicAdd(&Ab[t * GB_LDJ + t], dis_s[t] * dis_s[t] * loop_w);
    __syncthreads();
    if ((UP || POOL) && t < GC_N) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < GB_NT / 64; ++k) tot += (double)bs_s[k][t >> 2][t & 3];
        br.bias_parts[(size_t)b * H + ns0 + t] = tot;
    }
    BLK_CLK(2);
    auto ident = [](float v) { return v; };
    gc_f32x16 acc[2];
    // ---- P1: dz[:, ns] = Ab^T dOut[:, ns]   (rows i x 64 columns, reduction over the graph's rowsP nodes) ------------
    {
        const int rt = w >> 1, ct = w & 1;              // waves 0-3: one tile each; waves 4-7 wait at the barriers
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
        if (rt < R) gb_mma<1, 1, GB_LDJ, GB_LDD>(Ab + rt * 32 + li, nullptr, Ds + ct * 32 + li, nullptr, rowsP, lk, ident, acc);
        if (POOL && w >= 4) {
            // waves 4-7 (idle during P1): gn / gself of this slice, 4 lanes per item (16 columns each) straight from LDS
            const int q4 = (t - 256) & 3, it0 = (t - 256) >> 2;
            float* gn = br.gn + (size_t)sl * br.gn_stride;
            float* gs = br.gself + (size_t)sl * br.gself_stride;
            for (int it = it0; it - it0 < ne + rows; it += 64) {
                const bool ok = it < ne + rows, isedge = it < ne;
                const int itc = ok ? it : 0;
                const int jd = isedge ? er[itc] : itc - ne, js = isedge ? en[itc] : itc - ne;      // destination / source row
                const float4* a = reinterpret_cast<const float4*>(Ds + (ok ? jd : 0) * GB_LDD + 16 * q4);
                const float4* bsrc = reinterpret_cast<const float4*>(Zr + (ok ? js : 0) * GB_LDD + 16 * q4);
                float p = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) p = dot4(a[k], bsrc[k], p);
                p += __shfl_xor(p, 1, 64);
                p += __shfl_xor(p, 2, 64);
                if (ok && q4 == 0) {
                    if (isedge) gn[(LEAN || br.gn_slot) ? e0 + itc : ee[itc]] = p; else gs[g0 + itc - ne] = p;
                }
            }
        }
        __syncthreads();                                 // every wave is done reading dOut (and the z rows)
        if (POOL && LEAN) commit_x();                    // x_hat over the z rows (visible after the barrier below)
        if (rt < R) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                Ds[row * GB_LDD + ct * 32 + li] = acc[0][r];           // dz row-major over the dOut stage
            }
        }
        __syncthreads();
    }
    // ---- P2: partial dX'[:, :] = dz[:, ns] W[:, ns]^T   (rows i x K columns, reduction over the 64 columns of ns) ------
    if (w < 4 && w * 32 < K) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
        // both operands row-major in the reduction index n (dz rows in Ds, W rows in Ws, stride 68 = 4 mod 32): 16 B reads,
        // four MFMA steps per read; W is staged as loaded (no transposing scatter) and dz needs no transposed copy
        // (LEAN: the lane's W row -- 64 consecutive floats of row w * 32 + li -- comes straight from global memory / L2)
        const float* wrow = LEAN ? br.W + (size_t)min(w * 32 + li, K - 1) * H + ns0 : Ws + (w * 32 + li) * GB_LDD;
        if (R == 2) gb_mma_rowk2<true>(Ds + li * GB_LDD, Ds + (32 + li) * GB_LDD, wrow, GC_N, lk, acc[0], acc[1]);
        else gb_mma_rowk2<false>(Ds + li * GB_LDD, nullptr, wrow, GC_N, lk, acc[0], acc[1]);
        const int k = w * 32 + li;
        float* dxp = sl ? br.dxp1 : br.dxp0;
        // x_hat of all rows first, as ONE batch of unconditional LDS reads (rows rows .. rowsP are zero, as are their dz;
        // rows past rowsP hold stale LDS and are masked after the read): written as `q < R ? Xs[..] : 0` hipcc made 32
        // branches, each with its own ds_read + s_waitcnt lgkmcnt(0) -- 3.6 us of a 13 us kernel -- and before that, with the
        // read inside the guarded store block, one ~140 ns iteration at a time
        float xh[2][16];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) xh[q][r] = Xs[(q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * LDX + k];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) { asm volatile("" : "+v"(xh[q][r])); if (q >= R) xh[q][r] = 0.f; }
        // this lane's 32 terms of the two column sums in fp32 (four independent chains), everything across lanes,
        // workgroups and graphs in fp64 as before: 128 dependent fp64 conversions / adds per lane were 3.5 us here
        float f1[4] = {0.f, 0.f, 0.f, 0.f}, f2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float v = acc[q][r];                   // (row tile 1 of a one-tile graph: zero accumulators)
                f1[r & 3] += v;
                f2[r & 3] = fmaf(v, xh[q][r], f2[r & 3]);
            }
        }
        // (16-byte stores, four columns per lane: gc_store_tile)
        gc_store_tile(acc[0], dxp + (size_t)g0 * K + w * 32, K, rows, li, lk);
        if (R == 2) gc_store_tile(acc[1], dxp + (size_t)(g0 + 32) * K + w * 32, K, rows - 32, li, lk);
        double s1 = ((double)f1[0] + (double)f1[1]) + ((double)f1[2] + (double)f1[3]);
        double s2 = ((double)f2[0] + (double)f2[1]) + ((double)f2[2] + (double)f2[3]);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (lk == 0) {
            if (br.dacc_sum) {
                const size_t po = (size_t)stripe_of_block() * br.dacc_ss + k;
                atomicAdd(br.dacc_sum + po, s1); atomicAdd(br.dacc_prod + po, s2);
            } else { parts[k] = s1; parts[K + k] = s2; }
        }
    }
    BLK_CLK(3);
    // ---- P3: dW[:, ns] (this graph) = x'^T dz[:, ns]   (K rows x 64 columns, reduction over the graph's nodes) ---------
    if (w >= 4 && (w - 4) * 32 < K) {
        const int wq = w - 4, k = wq * 32 + li;
        const float gam = gam_s[k], bet = bet_s[k];
        auto affine = [&](float v) { return fmaf(v, gam, bet); };
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
        gb_mma<1, 2, LDX, GB_LDD>(Xs + wq * 32 + li, nullptr, Ds + li, Ds + 32 + li, rowsP, lk, affine, acc);
#pragma unroll
        for (int q = 0; q < 2; ++q) gc_store_tile(acc[q], slab + (size_t)(wq * 32) * H + ns0 + q * 32, H, 32, li, lk);
    }
    BLK_CLK(1);
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of the feature layer h0 = relu(BN0(x0) W_feat) (model.py:90-91, a GCNConv with gfn=True: no aggregation)
// per graph, fed like the UP variant above: dZ = BatchNorm_1-backward(dy0 + dy1) masked by h0 > 0 is built in LDS
// from the first backbone layer's partial dX', then
//     dW_feat (this graph's slab [F,H]) = x0_hat'^T dZ,   (sum dX0, sum dX0 * x0_hat) per feature with dX0 = dZ W_feat^T
// (the BatchNorm_0 affine gradients).  F is small (10 for SPMotif): plain FMA loops on LDS operands; replaces k_bn_bwd +
// the dual GEMM of the unfused path.  grid (B), 512 threads, graphs of at most FB_T nodes, F <= FB_F.
// ------------------------------------------------------------------------------------------------------------------
constexpr int FB_T = 64, FB_F = 64, FB_H = 128;
struct FeatBwdArgs {
    const float* dy0; const float* dy1;      // partials of the first backbone layer's dX' (dy1 null when H == 64)
    const float* y;                          // h0 [N,H]
    BNRef ubn; const double* udot_sum; const double* udot_prod;     // BatchNorm_1
    const float* x0;                         // [N,F]
    const float* W;                          // [F,H]
    BNRef bn0;
    float* slab;                             // [B][F,H]
    double* parts;                           // [B][2F]: (sum dX0, sum dX0 * x0_hat)
};
// gptr == null (round 6: graphs of 129-256 nodes behind the wide convolutions, engine_gwide.hpp): the units are uniform chunks of FB_T
// ROWS of the batch, b * FB_T .. -- nothing here looks at the graph structure (dW_feat and the two BatchNorm_0 sums are sums over
// rows), so a 240-node graph is simply four units with a slab and a partial row each.
// FMAX: the LDS arrays' feature capacity (16: 46 KB, three workgroups per CU -- SPMotif's F = 10 over 470 row chunks was two rounds
// of one workgroup per CU at 100 KB; 64: any F <= FB_F)
template <int FMAX>
__global__ void __launch_bounds__(GB_NT) k_feat_bwd(const int* __restrict__ gptr, const FeatBwdArgs a, int H, int F,
                                                  int* __restrict__ status, int N) {
    __shared__ __attribute__((aligned(16))) float Dz[FB_T * (FB_H + 4)];     // dZ rows [j][n]
    __shared__ float Xn[FB_T * FMAX];                    // x0_hat rows [j][f] (normalised, no affine)
    __shared__ __attribute__((aligned(16))) float Ws[FMAX * (FB_H + 4)];     // W_feat [f][n]
    __shared__ float um_s[FB_H], ur_s[FB_H], ug_s[FB_H], u1_s[FB_H], u2_s[FB_H];
    __shared__ float m0_s[FMAX], r0_s[FMAX], g0_s[FMAX], b0_s[FMAX];
    __shared__ float dX0[FB_T * FMAX];
    BLK_CLK(0);
    warm_kernargs<sizeof(FeatBwdArgs) + 32>();
    const int b = blockIdx.x, t = threadIdx.x, LDZ = FB_H + 4;
    const int g0 = gptr ? gptr[b] : b * FB_T, rows = gptr ? gptr[b + 1] - g0 : min(FB_T, N - g0);
    float* slab = a.slab + (size_t)b * F * H;
    double* parts = a.parts + (size_t)b * 2 * F;
    if (rows <= 0 || rows > FB_T) {
        if (rows > 0 && t == 0) atomicOr(status, 8);
        for (int i = t; i < F * H; i += GB_NT) slab[i] = 0.f;
        for (int i = t; i < 2 * F; i += GB_NT) parts[i] = 0.0;
        return;
    }
    const int H4 = H >> 2;
    // all loads first: the two partials and h0 (rows x H/4 float4 each, <= 8 per lane), x0, W, BN constants
    RoBatch<float4, 4> b0, b1, by, bwt;
    const float* d1 = a.dy1 ? a.dy1 : a.dy0;
    ro_issue<GB_NT>(b0, rows, H4, [&](int j, int c) { return *reinterpret_cast<const float4*>(a.dy0 + (size_t)(g0 + j) * H + 4 * c); });
    ro_issue<GB_NT>(b1, rows, H4, [&](int j, int c) { return *reinterpret_cast<const float4*>(d1 + (size_t)(g0 + j) * H + 4 * c); });
    ro_issue<GB_NT>(bwt, F, H4, [&](int f, int c) { return *reinterpret_cast<const float4*>(a.W + (size_t)f * H + 4 * c); });
    ro_issue<GB_NT>(by, rows, H4, [&](int j, int c) { return *reinterpret_cast<const float4*>(a.y + (size_t)(g0 + j) * H + 4 * c); });
    float xr[8];                                         // x0[g0 .. g0 + rows) is contiguous: rows * F <= 4096 floats
#pragma unroll
    for (int u = 0; u < 8; ++u) xr[u] = a.x0[(size_t)g0 * F + min(t + u * GB_NT, rows * F - 1)];
    // BatchNorm constants: unconditional loads on clamped columns, pinned with the tile loads (BNRaw, engine.hpp)
    const int uc = min(t, H - 1), fc = min(max(t - 128, 0), F - 1);
    BNRawS uraws = bn_raws_load(a.ubn, uc);              // (striped readers, engine.hpp)
    BNRaw raw0 = bn_raw_load(a.bn0, fc);
    StripeVal ud1s = stripe_load(a.udot_sum, uc, a.ubn.ss), ud2s = stripe_load(a.udot_prod, uc, a.ubn.ss);
    bn_raws_pin(uraws); bn_raw_pin(raw0);
    stripe_pin(ud1s); stripe_pin(ud2s);
    if (t < H) {
        float m1, r1;
        const BNRaw uraw = bn_raws_sum(a.ubn, uraws);
        const double ud1 = stripe_total(ud1s, a.ubn.ss), ud2 = stripe_total(ud2s, a.ubn.ss);
        bn_raw_mean_rstd(a.ubn, uraw, m1, r1);
        um_s[t] = m1; ur_s[t] = r1;
        ug_s[t] = uraw.g * r1;
        u1_s[t] = (float)(ud1 * (double)a.ubn.inv_n);
        u2_s[t] = (float)(ud2 * (double)a.ubn.inv_n);
    } else if (t - 128 < F && t >= 128) {
        const int f = t - 128;
        float m1, r1;
        bn_raw_mean_rstd(a.bn0, raw0, m1, r1);
        m0_s[f] = m1; r0_s[f] = r1;
        g0_s[f] = raw0.g;
        b0_s[f] = raw0.b;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(xr[u]));
    ro_commit<GB_NT>(bwt, F, H4, [&](int f, int c, const float4 v) { *reinterpret_cast<float4*>(Ws + f * LDZ + 4 * c) = v; });
    __syncthreads();
    {
        const bool two = a.dy1 != nullptr;
#pragma unroll
        for (int u = 0; u < 4; ++u) { ro_pin(b1.v[u]); ro_pin(by.v[u]); }
        // item (u, t) of a rows x H4 grid: walked like ro_commit does
        const int q = GB_NT / H4, r = GB_NT % H4;
        int row = t / H4, col = t % H4;
#pragma unroll
        for (int u = 0; u < 4; ++u) ro_pin(b0.v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (t + u * GB_NT < rows * H4) {
                const float4 v0 = b0.v[u], v1 = b1.v[u], yv = by.v[u];
                const int c = 4 * col;
                const float d[4] = {v0.x + (two ? v1.x : 0.f), v0.y + (two ? v1.y : 0.f), v0.z + (two ? v1.z : 0.f), v0.w + (two ? v1.w : 0.f)};
                const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float yn = (yy[k] - um_s[c + k]) * ur_s[c + k];
                    const float g1 = ug_s[c + k] * (d[k] - u1_s[c + k] - yn * u2_s[c + k]);
                    o[k] = yy[k] > 0.f ? g1 : 0.f;
                }
                *reinterpret_cast<float4*>(Dz + row * LDZ + c) = make_float4(o[0], o[1], o[2], o[3]);
            }
            row += q; col += r;
            if (col >= H4) { col -= H4; ++row; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = t + u * GB_NT;
            if (i < rows * F) { const int f = i % F; Xn[(i / F) * FMAX + f] = (xr[u] - m0_s[f]) * r0_s[f]; }
        }
    }
    __syncthreads();
    BLK_CLK(2);
    // dW_feat slab: output (f, 4 consecutive n), reduction over the graph's rows; x0_hat' = gamma0 x0_hat + beta0
    for (int o = t; o < F * H4; o += GB_NT) {
        const int f = o / H4, n = 4 * (o % H4);
        const float gam = g0_s[f], bet = b0_s[f];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int j = 0; j < rows; ++j) {
            const float xv = fmaf(Xn[j * FMAX + f], gam, bet);
            const float4 d = *reinterpret_cast<const float4*>(Dz + j * LDZ + n);
            acc.x = fmaf(xv, d.x, acc.x); acc.y = fmaf(xv, d.y, acc.y); acc.z = fmaf(xv, d.z, acc.z); acc.w = fmaf(xv, d.w, acc.w);
        }
        *reinterpret_cast<float4*>(slab + (size_t)f * H + n) = acc;
    }
    BLK_CLK(3);
    // BatchNorm_0 backward sums: dX0[j][f] = <dZ[j], W[f]>: lane = (row j = t / 4, quarter qn of the H columns) keeps its
    // quarter row of dZ in registers and walks the features; then one lane per feature sums over the rows (fixed order)
    {
        const int j = (t & 255) >> 2, qn = t & 3, nq4 = H >> 4;  // float4s per quarter row (H % 16 == 0)
        const int fh = (F + 1) >> 1, f_lo = t < 256 ? 0 : fh, f_hi = t < 256 ? fh : F;     // the two halves of the block split the features
        // (reads unconditional on a clamped float4 index, masked afterwards: `k < nq4 ? read : 0` compiles to one branch with
        //  its own LDS round trip per element -- 40 serial reads in the feature loop below)
        float4 dz[FB_H / 16];
#pragma unroll
        for (int k = 0; k < FB_H / 16; ++k)
            dz[k] = *reinterpret_cast<const float4*>(Dz + min(j, rows - 1) * LDZ + 4 * (qn * nq4 + min(k, nq4 - 1)));
#pragma unroll
        for (int k = 0; k < FB_H / 16; ++k) { ro_pin(dz[k]); if (k >= nq4) dz[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
        for (int f = f_lo; f < f_hi; ++f) {
            const float4* wr = reinterpret_cast<const float4*>(Ws + f * LDZ) + qn * nq4;
            float4 wv[FB_H / 16];
#pragma unroll
            for (int k = 0; k < FB_H / 16; ++k) wv[k] = wr[min(k, nq4 - 1)];
            float p = 0.f;
#pragma unroll
            for (int k = 0; k < FB_H / 16; ++k) p = dot4(dz[k], wv[k], p);
            p += __shfl_xor(p, 1, 64);
            p += __shfl_xor(p, 2, 64);
            if (qn == 0 && j < rows) dX0[j * FMAX + f] = p;
        }
    }
    __syncthreads();
    // eight lanes per feature, each over every eighth row (<= 8 terms, fp32), combined in fp64 by shuffle: one lane per
    // feature walked the graph's rows as a chain of 57 dependent LDS round trips + fp64 adds while 500 lanes idled
    {
        const int f = t >> 3, p = t & 7, fc = min(f, F - 1);
        float a1 = 0.f, a2 = 0.f;
        float dv[FB_T / 8], xv[FB_T / 8];
#pragma unroll
        for (int u = 0; u < FB_T / 8; ++u) {
            const int j = min(p + 8 * u, rows - 1);
            dv[u] = dX0[j * FMAX + fc]; xv[u] = Xn[j * FMAX + fc];
        }
#pragma unroll
        for (int u = 0; u < FB_T / 8; ++u) {
            const float v = p + 8 * u < rows ? dv[u] : 0.f;
            a1 += v; a2 = fmaf(v, xv[u], a2);
        }
        double s1 = (double)a1, s2 = (double)a2;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (p == 0 && f < F) { parts[f] = s1; parts[F + f] = s2; }
    }
    BLK_CLK(1);
}

// ------------------------------------------------------------------------------------------------------------------
// The same feature-layer backward on the matrix cores, for any F <= FM_F (one-hot degree features of the TU datasets:
// F = 109 / 139, datasets.py:16-20) -- and without the dX0 = dZ W_feat^T product at all.  With P = x0_hat^T dZ (this unit's
// [F,H] block) and cs = column sums of dZ, everything the layer needs is linear in P:
//     dW_feat   = (gamma0 x0_hat + beta0)^T dZ  = gamma0[f] P[f,:] + beta0[f] cs
//     sum_j dX0[j,f]            = <W[f,:], cs>            (dX0 = dZ W^T)
//     sum_j dX0[j,f] x0_hat[j,f] = <W[f,:], P[f,:]>
// so one MFMA product per unit replaces two products and a row pass.  NOBN: the layer above is not behind a BatchNorm
// (CausalGIN: h0 feeds GINConv directly): dZ = (dy0 + dy1) masked by h0 > 0.
//   grid (units), 512 threads; XU = x0 loads per lane (8: F <= 64, 20: F <= 160).
// ------------------------------------------------------------------------------------------------------------------
constexpr int FM_F = 160;
template <int XU, bool NOBN>
__global__ void __launch_bounds__(GB_NT) k_feat_bwd_mma(const int* __restrict__ gptr, const FeatBwdArgs a, int H, int F,
                                                      int* __restrict__ status) {
    constexpr int FP = XU == 8 ? 64 : FM_F, LDXN = FP + 1, LDZ = FB_H + 4;
    __shared__ __attribute__((aligned(16))) float Dz[FB_T * LDZ];           // dZ rows [j][n]
    __shared__ float Xn[FB_T * LDXN];                    // x0_hat rows [j][f] (normalised, no affine), zero beyond F / rows
    __shared__ float um_s[FB_H], ur_s[FB_H], ug_s[FB_H], u1_s[FB_H], u2_s[FB_H];
    __shared__ float m0_s[FP], r0_s[FP], g0_s[FP], b0_s[FP];
    __shared__ float cs_s[FB_H];
    __shared__ float csp[GB_NT / 64][32][4];
    __shared__ float s12[2][4][FP];                      // per column tile: partial <W[f,:], cs>, <W[f,:], P[f,:]>
    __shared__ float red_s[GB_NT / 64][2][32][33];       // per wave: the two product tiles [feature][column], summed over the columns
    BLK_CLK(0);
    warm_kernargs<sizeof(FeatBwdArgs) + 32>();
    const int b = blockIdx.x, t = threadIdx.x;
    const int g0 = gptr[b], rows = gptr[b + 1] - g0;
    float* slab = a.slab + (size_t)b * F * H;
    double* parts = a.parts + (size_t)b * 2 * F;
    if (rows <= 0 || rows > FB_T) {
        if (rows > 0 && t == 0) atomicOr(status, 8);
        for (int i = t; i < F * H; i += GB_NT) slab[i] = 0.f;
        for (int i = t; i < 2 * F; i += GB_NT) parts[i] = 0.0;
        return;
    }
    const int H4 = H >> 2, rowsP = (rows + 31) & ~31, nct = H >> 5, nft = (F + 31) >> 5;
    RoBatch<float4, 4> b0, b1, by;
    const float* d1 = a.dy1 ? a.dy1 : a.dy0;
    ro_issue<GB_NT>(b0, rows, H4, [&](int j, int c) { return *reinterpret_cast<const float4*>(a.dy0 + (size_t)(g0 + j) * H + 4 * c); });
    ro_issue<GB_NT>(b1, rows, H4, [&](int j, int c) { return *reinterpret_cast<const float4*>(d1 + (size_t)(g0 + j) * H + 4 * c); });
    ro_issue<GB_NT>(by, rows, H4, [&](int j, int c) { return *reinterpret_cast<const float4*>(a.y + (size_t)(g0 + j) * H + 4 * c); });
    float xr[XU];                                        // x0[g0 .. g0 + rows) is contiguous: rows * F floats
#pragma unroll
    for (int u = 0; u < XU; ++u) xr[u] = a.x0[(size_t)g0 * F + min(t + u * GB_NT, rows * F - 1)];
    // W rows of this wave's first output tile (the epilogue's <W[f,:], .> sums): requested now, not behind the product
    const int lane = t & 63, li = lane & 31, lk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ct = w % nct, fstep = (GB_NT / 64) / nct;
    float wv[16];
    auto load_w = [&](int ft) {
#pragma unroll
        for (int r = 0; r < 16; ++r) wv[r] = a.W[(size_t)min(ft * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, F - 1) * H + ct * 32 + li];
    };
    load_w(w / nct);
    const int uc = min(t, H - 1), fc = min(max(t - 128, 0), F - 1);
    BNRawS uraws;
    BNRaw raw0 = bn_raw_load(a.bn0, fc);
    StripeVal ud1s, ud2s;
    if (!NOBN) { uraws = bn_raws_load(a.ubn, uc); ud1s = stripe_load(a.udot_sum, uc, a.ubn.ss); ud2s = stripe_load(a.udot_prod, uc, a.ubn.ss); bn_raws_pin(uraws); stripe_pin(ud1s); stripe_pin(ud2s); }
    bn_raw_pin(raw0);
    if (!NOBN && t < H) {
        float m1, r1;
        const double ud1 = stripe_total(ud1s, a.ubn.ss), ud2 = stripe_total(ud2s, a.ubn.ss);
        const BNRaw uraw = bn_raws_sum(a.ubn, uraws);
        bn_raw_mean_rstd(a.ubn, uraw, m1, r1);
        um_s[t] = m1; ur_s[t] = r1;
        ug_s[t] = uraw.g * r1;
        u1_s[t] = (float)(ud1 * (double)a.ubn.inv_n);
        u2_s[t] = (float)(ud2 * (double)a.ubn.inv_n);
    }
    if (t >= 128 && t - 128 < FP) {
        const int f = t - 128;
        float m1, r1;
        bn_raw_mean_rstd(a.bn0, raw0, m1, r1);
        m0_s[f] = m1; r0_s[f] = f < F ? r1 : 0.f;
        g0_s[f] = raw0.g;
        b0_s[f] = raw0.b;
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) asm volatile("" : "+v"(xr[u]));
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(wv[r]));
    // zero the padding of the x0_hat tile (columns F .. FP of every row, rows rows .. rowsP): they are reduced over / read
    for (int i = t; i < rowsP * LDXN; i += GB_NT) Xn[i] = 0.f;
    __syncthreads();
    {
        const bool two = a.dy1 != nullptr;
#pragma unroll
        for (int u = 0; u < 4; ++u) { ro_pin(b0.v[u]); ro_pin(b1.v[u]); ro_pin(by.v[u]); }
        // item (u, t): row (t + u * 512) / H4, float4 column t % H4 (512 % H4 == 0: a lane keeps its column group)
        const int col = t % H4, c = 4 * col, rstep = GB_NT / H4;
        int row = t / H4;
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (row < rowsP) {
                float o[4] = {0.f, 0.f, 0.f, 0.f};
                if (row < rows) {
                    const float4 v0 = b0.v[u], v1 = b1.v[u], yv = by.v[u];
                    const float d[4] = {v0.x + (two ? v1.x : 0.f), v0.y + (two ? v1.y : 0.f), v0.z + (two ? v1.z : 0.f), v0.w + (two ? v1.w : 0.f)};
                    const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float g1 = d[k];
                        if (!NOBN) {
                            const float yn = (yy[k] - um_s[c + k]) * ur_s[c + k];
                            g1 = ug_s[c + k] * (d[k] - u1_s[c + k] - yn * u2_s[c + k]);
                        }
                        o[k] = yy[k] > 0.f ? g1 : 0.f;
                        cs[k] += o[k];
                    }
                }
                *reinterpret_cast<float4*>(Dz + row * LDZ + c) = make_float4(o[0], o[1], o[2], o[3]);
            }
            row += rstep;
        }
        // column sums of dZ: lanes of a wave with the same column group, then the eight waves through LDS
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (H4 <= 32) cs[k] += __shfl_xor(cs[k], 32, 64);
            if (H4 <= 16) cs[k] += __shfl_xor(cs[k], 16, 64);
        }
        const int lane = t & 63;
        if (lane < H4 && lane < 32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) csp[t >> 6][lane][k] = cs[k];
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = t + u * GB_NT;
            if (i < rows * F) { const int f = i % F; Xn[(i / F) * LDXN + f] = (xr[u] - m0_s[f]) * r0_s[f]; }
        }
    }
    __syncthreads();
    if (t < H) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < GB_NT / 64; ++k) tot += csp[k][t >> 2][t & 3];
        cs_s[t] = tot;
    }
    __syncthreads();
    BLK_CLK(2);
    // P = x0_hat^T dZ on the matrix cores: wave w takes column tile w % nct and the feature tiles w / nct, + 8 / nct, ..
    auto ident = [](float v) { return v; };
    for (int ft = w / nct; ft < nft; ft += fstep) {
        gc_f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
        gb_mma<1, 1, LDXN, LDZ>(Xn + ft * 32 + li, nullptr, Dz + ct * 32 + li, nullptr, rowsP, lk, ident, acc);
        const int h = ct * 32 + li;
        const float csh = cs_s[h];
        float wc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) wc[r] = wv[r];
        float p1[16], p2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = ft * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const float pv = acc[0][r];
            acc[1][r] = fmaf(g0_s[min(f, FP - 1)], pv, b0_s[min(f, FP - 1)] * csh);         // this unit's dW_feat entry
            p1[r] = f < F ? wc[r] * csh : 0.f;
            p2[r] = f < F ? wc[r] * pv : 0.f;
        }
        // sums over the tile's 32 columns through this wave's LDS scratch: lane (which = lane / 32, feature = lane % 32) adds
        // its row (160 ds_bpermute shuffles per tile, each behind its own lgkmcnt wait, were ~4 us per tile)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int fl = (r & 3) + 8 * (r >> 2) + 4 * lk;
            red_s[w][0][fl][li] = p1[r];
            red_s[w][1][fl][li] = p2[r];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): the wave's own LDS writes (no other wave touches red_s[w])
        {
            const float* rr = &red_s[w][lk][li][0];
            float v[32], tot = 0.f;
#pragma unroll
            for (int q = 0; q < 32; ++q) v[q] = rr[q];
#pragma unroll
            for (int q = 0; q < 32; ++q) tot += v[q];
            const int f = ft * 32 + li;
            if (f < FP) s12[lk][ct][f] = tot;
        }
        // the guarded stores go LAST, with no load in flight (hipcc waits for vmcnt(0) at the head of every guarded block --
        // with the next tile's W rows already requested, each of the 16 stores waited for them and for its predecessor:
        // 4 us per tile), and the next tile's W rows are requested behind them, under the next product
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = ft * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (f < F) slab[(size_t)f * H + h] = acc[1][r];
        }
        if (ft + fstep < nft) {
            load_w(ft + fstep);
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(wv[r]));
        }
    }
    BLK_CLK(3);
    __syncthreads();
    if (t < F) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < nct; ++k) { s1 += (double)s12[0][k][t]; s2 += (double)s12[1][k][t]; }
        parts[t] = s1; parts[F + t] = s2;
    }
    BLK_CLK(1);
}

}  // namespace cal
