// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak) for the
// dense linear layers of the hot path -- the only place MFMA is used (BASELINE.json north_star):
//   x @ W          gcn_conv.py:75, GATConv's lin          -> NN
//   Linear(x)      model.py:57-74, 102, 109 (x @ W^T)     -> NT
//   dX, dW         autograd of the above                  -> NT/NN and TN (split-K)
//
// Workgroup tile 64x64, K step 32, 4 waves (2x2), one 32x32 accumulator per wave.  Both operands
// are staged in LDS k-major (As[k][i], Bs[k][j]) so each MFMA operand fetch is one conflict-free
// ds_read_b32 of 32 consecutive floats per half-wave.  Two LDS stages, one barrier per K tile:
// the global loads of tile t+1 are issued raw into registers, then the 32 operand reads and the 16
// MFMAs of tile t run, and only then are the prefetched registers transformed and stored (the
// sched_barriers keep hipcc from waiting on them early).  64x64 tiles keep >= 230 workgroups in
// flight for the config-2 shape [7315,128]x[128,128] (256 CUs).
//
// Fusions (engine.hpp): BatchNorm-apply (+ node-attention row scale) on either operand while it is
// staged (model.py:90,94,112-113,127-131 -- BN outputs are never materialised), bias + ReLU
// epilogue, per-column sum / sum-of-squares of the output (the next BatchNorm's batch statistics)
// and the BN-backward column sums, pre-reduced per workgroup in fp64 and either written as one
// partial row per row-tile (large launches; finished by k_stats_final) or added atomically (small).
#include "engine.hpp"

namespace cal {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32;   // BK = 64 measured slower (9.6 vs 8.8 us at K = 128)
constexpr int NQ = BM * BK / 4 / 256;   // float4 per thread per operand tile
constexpr int LDT = 65;   // LDS row stride (floats) for tiles filled by transposing scalar stores
constexpr int LDD = 68;   // LDS row stride for tiles filled by direct 16B stores
constexpr int XMAX = 512; // max feature width of a BN-transformed k-contiguous operand
#ifdef CAL_GEMM_CLOCKS                    // profiling aid: phase timestamps (100 MHz) of workgroup (1,0,0)
__device__ long long g_gemm_clk[16];
__device__ long long g_gemm_blk[2 * 2048];    // start / end timestamp of every workgroup (x + gridDim.x * y)
#define GEMM_CLK(k) do { if (threadIdx.x == 0 && blockIdx.x == 1 && blockIdx.y == 0 && blockIdx.z == 0) g_gemm_clk[k] = wall_clock64(); } while (0)
#else
#define GEMM_CLK(k) do {} while (0)
#endif
constexpr int PRE_T = 4;  // K tiles a workgroup can preload at once (K chunk <= 128): see gemm_preloaded

// Operand tiles.  The operand is logically T[mn][k] (mn = row of A / column of B).
//   KC = true : memory is [mn][k] row-major (k contiguous)  -> transposing LDS store
//   KC = false: memory is [k][mn] row-major (mn contiguous) -> direct 16 B LDS store
// MODE 0 = interior tile: unconditional 16 B loads.
// MODE 1 = ragged in mn only (last row / column tile; K range whole, 16 B aligned, and for !KC
//          operands mn_end % 4 == 0): 16 B loads from a CLAMPED row / column group.  The rows or
//          columns past the end then hold copies of valid data, which only ever reach accumulator
//          rows / columns the epilogue never stores -- no zeroing, same speed as an interior tile
//          (the scalar path made the one ragged workgroup of a [7315,128] launch the critical path:
//          18 us against 5.6 us for its 229 neighbours).
// MODE 2 = anything else: every element from a clamped (always valid) address, zeroed at store time.
// No divergent control flow in any mode.
template <bool KC, int MODE>
__device__ __forceinline__ void tile_load(float4 (&r)[NQ], const float* __restrict__ p, int ld, int mn0, int mn_end,
                                          int k0, int k_end) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int f = threadIdx.x + q * 256;
        const int mn = KC ? f / (BK / 4) : (f % (BM / 4)) * 4;
        const int k = KC ? (f % (BK / 4)) * 4 : f / (BM / 4);
        if (MODE == 0) {
            r[q] = KC ? *reinterpret_cast<const float4*>(p + (size_t)(mn0 + mn) * ld + k0 + k)
                      : *reinterpret_cast<const float4*>(p + (size_t)(k0 + k) * ld + mn0 + mn);
        } else if (MODE == 1) {
            r[q] = KC ? *reinterpret_cast<const float4*>(p + (size_t)min(mn0 + mn, mn_end - 1) * ld + k0 + k)
                      : *reinterpret_cast<const float4*>(p + (size_t)(k0 + k) * ld + min(mn0 + mn, mn_end - 4));
        } else if (KC) {
            const float* row = p + (size_t)min(mn0 + mn, mn_end - 1) * ld;
            const int kl = k_end - 1;
            r[q] = make_float4(row[min(k0 + k, kl)], row[min(k0 + k + 1, kl)], row[min(k0 + k + 2, kl)], row[min(k0 + k + 3, kl)]);
        } else {
            const float* row = p + (size_t)min(k0 + k, k_end - 1) * ld;
            const int ml = mn_end - 1;
            r[q] = make_float4(row[min(mn0 + mn, ml)], row[min(mn0 + mn + 1, ml)], row[min(mn0 + mn + 2, ml)], row[min(mn0 + mn + 3, ml)]);
        }
    }
}

// XF: 0 = plain, 1 = BN scale/shift on the feature axis, 2 = per-storage-row scale, then BN.
// sc/sh: LDS tables indexed by (k - kb) for KC operands and by the tile-local mn for !KC ones.
template <bool KC, int MODE, int XF>
__device__ __forceinline__ void tile_store(const float4 (&r)[NQ], float* __restrict__ s, int mn0, int mn_end, int k0,
                                           int k_end, int kb, const float* __restrict__ rsp, int rs_stride,
                                           const float* sc, const float* sh) {
    constexpr int LD = KC ? LDT : LDD;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int f = threadIdx.x + q * 256;
        const int mn = KC ? f / (BK / 4) : (f % (BM / 4)) * 4;
        const int k = KC ? (f % (BK / 4)) * 4 : f / (BM / 4);
        float v[4] = {r[q].x, r[q].y, r[q].z, r[q].w};
        if (XF > 0) {
            float rs = 1.f;
            if (XF == 2) rs = rsp[(size_t)(KC ? min(mn0 + mn, mn_end - 1) : min(k0 + k, k_end - 1)) * rs_stride];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int fi = KC ? (k0 + k + j - kb) : (mn + j);     // feature index into the tables
                v[j] = fmaf(XF == 2 ? rs * v[j] : v[j], sc[fi], sh[fi]);
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = KC ? (mn0 + mn < mn_end && k0 + k + j < k_end) : (k0 + k < k_end && mn0 + mn + j < mn_end);
                v[j] = ok ? v[j] : 0.f;
            }
        }
        if (KC) {
            s[(k + 0) * LD + mn] = v[0]; s[(k + 1) * LD + mn] = v[1];
            s[(k + 2) * LD + mn] = v[2]; s[(k + 3) * LD + mn] = v[3];
        } else {
            *reinterpret_cast<float4*>(s + k * LD + mn) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

template <bool A_KC, bool B_KC, int XA, int XB, int MODE>
__device__ __forceinline__ void gemm_kloop(const GemmArgs& a, const GemmProb& pr, float* As, float* Bs, int m0, int n0,
                                           int kb, int ke, const float* sca, const float* sha, const float* scb,
                                           const float* shb, f32x16& acc, int wm, int wn, int li, int lk) {
    constexpr int LDA = A_KC ? LDT : LDD, LDB = B_KC ? LDT : LDD;
    constexpr int SA = BK * LDA, SB = BK * LDB;
    const int M = a.M, N = a.N;
    float4 ra[NQ], rb[NQ];
    tile_load<A_KC, MODE>(ra, pr.A, a.lda, m0, M, kb, ke);
    tile_load<B_KC, MODE>(rb, pr.B, a.ldb, n0, N, kb, ke);
    tile_store<A_KC, MODE, XA>(ra, As, m0, M, kb, ke, kb, pr.xa.rs, pr.xa.rs_stride, sca, sha);
    tile_store<B_KC, MODE, XB>(rb, Bs, n0, N, kb, ke, kb, pr.xb.rs, pr.xb.rs_stride, scb, shb);
    __syncthreads();
    int st = 0;
    for (int k0 = kb; k0 < ke; k0 += BK) {
        const bool more = k0 + BK < ke;
        if (more) {
            tile_load<A_KC, MODE>(ra, pr.A, a.lda, m0, M, k0 + BK, ke);
            tile_load<B_KC, MODE>(rb, pr.B, a.ldb, n0, N, k0 + BK, ke);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float* as = As + st * SA + wm + li;
        const float* bs = Bs + st * SB + wn + li;
        float av[BK / 2], bv[BK / 2];
#pragma unroll
        for (int i = 0; i < BK / 2; ++i) {
            av[i] = as[(2 * i + lk) * LDA];
            bv[i] = bs[(2 * i + lk) * LDB];
        }
#pragma unroll
        for (int i = 0; i < BK / 2; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            tile_store<A_KC, MODE, XA>(ra, As + (st ^ 1) * SA, m0, M, k0 + BK, ke, kb, pr.xa.rs, pr.xa.rs_stride, sca, sha);
            tile_store<B_KC, MODE, XB>(rb, Bs + (st ^ 1) * SB, n0, N, k0 + BK, ke, kb, pr.xb.rs, pr.xb.rs_stride, scb, shb);
        }
        __syncthreads();
        st ^= 1;
    }
}

__device__ __forceinline__ void pin4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// Short reductions (K chunk <= PRE_T * BK = 128: every node-level GEMM of the hot path and the
// 128-deep slices of the split-K weight gradients).  The streaming loop above pays one dependent
// global round trip per K tile (4 x ~1 us at K = 128, nothing else is resident on the CU to hide
// it); here ALL operand tiles are requested before anything waits -- even before the BatchNorm
// tables are built -- then staged into PRE_T LDS stages at once and multiplied back to back.
template <bool A_KC, bool B_KC, int MODE>
__device__ __forceinline__ void pre_issue(const GemmArgs& a, const GemmProb& pr, int m0, int n0, int kb, int ke, int nt,
                                          float4 (&ra)[PRE_T][NQ], float4 (&rb)[PRE_T][NQ]) {
#pragma unroll
    for (int t = 0; t < PRE_T; ++t)
        if (t < nt) {
            tile_load<A_KC, MODE>(ra[t], pr.A, a.lda, m0, a.M, kb + t * BK, ke);
            tile_load<B_KC, MODE>(rb[t], pr.B, a.ldb, n0, a.N, kb + t * BK, ke);
        }
    asm volatile("" ::: "memory");      // the loads stay here (not sunk next to their LDS stores), and nothing waits yet
}
template <bool A_KC, bool B_KC, int XA, int XB, int MODE>
__device__ __forceinline__ void pre_commit(const GemmArgs& a, const GemmProb& pr, float* As, float* Bs, int m0, int n0,
                                           int kb, int ke, int nt, const float* sca, const float* sha, const float* scb,
                                           const float* shb, const float4 (&ra)[PRE_T][NQ], const float4 (&rb)[PRE_T][NQ]) {
    constexpr int SA = BK * (A_KC ? LDT : LDD), SB = BK * (B_KC ? LDT : LDD);
#pragma unroll
    for (int t = 0; t < PRE_T; ++t)
        if (t < nt) {
            tile_store<A_KC, MODE, XA>(ra[t], As + t * SA, m0, a.M, kb + t * BK, ke, kb, pr.xa.rs, pr.xa.rs_stride, sca, sha);
            tile_store<B_KC, MODE, XB>(rb[t], Bs + t * SB, n0, a.N, kb + t * BK, ke, kb, pr.xb.rs, pr.xb.rs_stride, scb, shb);
        }
}

// LDS of one workgroup, carved from a raw buffer so that two differently-shaped bodies can share a kernel
// (k_gemm_dual): operand stages, BN tables, epilogue reduction scratch.
template <bool A_KC, bool B_KC>
struct GemmSmem {
    static constexpr int XW = A_KC || B_KC ? XMAX : BM;
    static constexpr int A = 0;
    static constexpr int B = A + PRE_T * BK * (A_KC ? LDT : LDD);
    static constexpr int SC = B + PRE_T * BK * (B_KC ? LDT : LDD);
    static constexpr int SH = SC + 2 * XW;
    static constexpr int RED = (SH + 2 * XW + 1) / 2 * 2;          // doubles: 8-byte aligned
    static constexpr int FLOATS = RED + 2 * 4 * 2 * 32;
};

// one 64x64 output tile; (bx, by, bz) = tile row, tile column, batch * nsplit + split; nbx = row tiles
template <bool A_KC, bool B_KC, int XA, int XB>
__device__ __forceinline__ void gemm_block(const GemmArgs& a, int vecA, int vecB, int bx, int by, int bz, int nbx,
                                           float* __restrict__ smem) {
    using SM = GemmSmem<A_KC, B_KC>;
    float* As = smem + SM::A;
    float* Bs = smem + SM::B;
    float (*xsc)[SM::XW] = reinterpret_cast<float (*)[SM::XW]>(smem + SM::SC);
    float (*xsh)[SM::XW] = reinterpret_cast<float (*)[SM::XW]>(smem + SM::SH);
    double (*red)[2][32] = reinterpret_cast<double (*)[2][32]>(smem + SM::RED);

    const int batch = bz / a.nsplit, split = bz % a.nsplit;
    const GemmProb& pr = a.p[batch];
    const int M = a.M, N = a.N, K = a.K;
    const int m0 = bx * BM, n0 = by * BN;
    const int kb = split * a.kchunk, ke = min(K, kb + a.kchunk);
    float* C = pr.C ? pr.C + (size_t)split * M * a.ldc : nullptr;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int li = lane & 31, lk = lane >> 5;
    // interior tile: no bounds handling, unconditional 16 B loads
    const bool kwhole = vecA && vecB && ((ke - kb) % BK) == 0;
    const bool full = kwhole && m0 + BM <= M && n0 + BN <= N;
    // ragged in mn only: !KC operands are read 4 columns at a time, so their extent must be a multiple of 4
    const bool clampable = kwhole && (A_KC || (M % 4 == 0 && M >= 4)) && (B_KC || (N % 4 == 0 && N >= 4));
    const int mode = full ? 0 : (clampable ? 1 : 2);
    const int nt = (ke - kb + BK - 1) / BK;
    const bool pre = nt >= 1 && nt <= PRE_T;
    GEMM_CLK(0);
#ifdef CAL_GEMM_CLOCKS
    if (threadIdx.x == 0 && bz == 0 && bx + nbx * by < 2048) g_gemm_blk[2 * (bx + nbx * by)] = wall_clock64();
#endif
    float4 ra[PRE_T][NQ], rb[PRE_T][NQ];
    if (pre) {
        if (mode == 0) pre_issue<A_KC, B_KC, 0>(a, pr, m0, n0, kb, ke, nt, ra, rb);
        else if (mode == 1) pre_issue<A_KC, B_KC, 1>(a, pr, m0, n0, kb, ke, nt, ra, rb);
        else pre_issue<A_KC, B_KC, 2>(a, pr, m0, n0, kb, ke, nt, ra, rb);
    }

    // BN scale/shift tables of the transformed operands; one block also updates the running stats
    if (XA > 0) {
        const int cnt = A_KC ? (ke - kb) : min(BM, M - m0);
        const int c0 = A_KC ? kb : m0;
        for (int t = threadIdx.x; t < cnt; t += 256) {
            bn_scale_shift<true>(pr.xa.bn, c0 + t, xsc[0][t], xsh[0][t]);
            if (pr.xa.bn.update && by == 0 && split == 0 && (A_KC ? bx == 0 : true)) bn_update_running<true>(pr.xa.bn, c0 + t);
        }
        if (!A_KC) for (int t = cnt + threadIdx.x; t < BM; t += 256) { xsc[0][t] = 0.f; xsh[0][t] = 0.f; }
    }
    if (XB > 0) {
        const int cnt = B_KC ? (ke - kb) : min(BN, N - n0);
        const int c0 = B_KC ? kb : n0;
        for (int t = threadIdx.x; t < cnt; t += 256) {
            bn_scale_shift<true>(pr.xb.bn, c0 + t, xsc[1][t], xsh[1][t]);
            if (pr.xb.bn.update && bx == 0 && split == 0 && (B_KC ? by == 0 : true)) bn_update_running<true>(pr.xb.bn, c0 + t);
        }
        if (!B_KC) for (int t = cnt + threadIdx.x; t < BN; t += 256) { xsc[1][t] = 0.f; xsh[1][t] = 0.f; }
    }
    if (XA > 0 || XB > 0) __syncthreads();
    GEMM_CLK(1);

    f32x16 acc, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
    if (pre) {
        constexpr int LDA = A_KC ? LDT : LDD, LDB = B_KC ? LDT : LDD;
        if (mode == 0) pre_commit<A_KC, B_KC, XA, XB, 0>(a, pr, As, Bs, m0, n0, kb, ke, nt, xsc[0], xsh[0], xsc[1], xsh[1], ra, rb);
        else if (mode == 1) pre_commit<A_KC, B_KC, XA, XB, 1>(a, pr, As, Bs, m0, n0, kb, ke, nt, xsc[0], xsh[0], xsc[1], xsh[1], ra, rb);
        else pre_commit<A_KC, B_KC, XA, XB, 2>(a, pr, As, Bs, m0, n0, kb, ke, nt, xsc[0], xsh[0], xsc[1], xsh[1], ra, rb);
        GEMM_CLK(2);
        __syncthreads();
        GEMM_CLK(3);
        // operands of tile t+1 are read while tile t multiplies; the sched_barriers keep hipcc from
        // re-interleaving "2 reads, wait, 2 MFMAs" (which exposes one LDS latency per MFMA pair);
        // two accumulator chains, since a dependent 32x32x2 MFMA cannot issue back to back
        float av[2][BK / 2], bv[2][BK / 2];
        auto read_ops = [&](int t, float (&ao)[BK / 2], float (&bo)[BK / 2]) {
            const float* as = As + t * BK * LDA + wm + li;
            const float* bs = Bs + t * BK * LDB + wn + li;
#pragma unroll
            for (int i = 0; i < BK / 2; ++i) {
                ao[i] = as[(2 * i + lk) * LDA];
                bo[i] = bs[(2 * i + lk) * LDB];
            }
        };
        read_ops(0, av[0], bv[0]);
#pragma unroll
        for (int t = 0; t < PRE_T; ++t)
            if (t < nt) {
                if (t + 1 < nt) read_ops(t + 1, av[(t + 1) & 1], bv[(t + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < BK / 2; i += 2) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][i], bv[t & 1][i], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][i + 1], bv[t & 1][i + 1], acc2, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
    } else if (ke > kb) {
        if (mode == 0) gemm_kloop<A_KC, B_KC, XA, XB, 0>(a, pr, As, Bs, m0, n0, kb, ke, xsc[0], xsh[0], xsc[1], xsh[1], acc, wm, wn, li, lk);
        else if (mode == 1) gemm_kloop<A_KC, B_KC, XA, XB, 1>(a, pr, As, Bs, m0, n0, kb, ke, xsc[0], xsh[0], xsc[1], xsh[1], acc, wm, wn, li, lk);
        else gemm_kloop<A_KC, B_KC, XA, XB, 2>(a, pr, As, Bs, m0, n0, kb, ke, xsc[0], xsh[0], xsc[1], xsh[1], acc, wm, wn, li, lk);
    }
    GEMM_CLK(4);
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int col = n0 + wn + li;
    const bool cok = col < N;
    const float bv = (pr.bias && cok) ? pr.bias[col] : 0.f;
    const bool want_st = pr.st_sum != nullptr, want_dot = pr.dot_sum != nullptr;
    float amean = 0.f, arstd = 0.f;
    float aux[16] = {};
    if (want_dot && cok) {
        // aux values and their row scales as one batch of unconditional loads (no scale: the aux pointer again, stride 0, value
        // ignored); a per-row `if (aux_rs)` made every row a load, a branch and a dependent second load (gemm_big.hip)
        const bool has_rs = pr.aux_rs != nullptr;
        const float* rsp = has_rs ? pr.aux_rs : pr.aux;
        const size_t rstr = has_rs ? (size_t)pr.aux_rs_stride : 0;
        float ars[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = min(m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk, M - 1);
            aux[r] = pr.aux[(size_t)row * N + col];
            ars[r] = rsp[(size_t)row * rstr];
        }
        bn_mean_rstd<true>(pr.aux_bn, col, amean, arstd);
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(aux[r]), "+v"(ars[r]));
#pragma unroll
        for (int r = 0; r < 16; ++r) aux[r] *= has_rs ? ars[r] : 1.f;
    }
    // hipcc re-inserts `s_waitcnt vmcnt(0)` at the head of every guarded block below while a load issued
    // before them may still be pending on some path; on gfx9 stores count in vmcnt too, so each store then
    // waits for the previous one's acknowledgement (16 x 125 ns measured).  Consume the loads here, once.
    asm volatile("" :: "v"(bv), "v"(amean), "v"(arstd));
    if (want_dot) {
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(aux[r]));
    }
    double s1 = 0.0, s2 = 0.0;
    GEMM_CLK(6);
    auto emit = [&](int r, int row) {
        float v = acc[r] + bv;
        if (a.relu) v = fmaxf(v, 0.f);
        if (C) C[(size_t)row * a.ldc + col] = v;
        if (want_st) { s1 += (double)v; s2 += (double)v * (double)v; }
        if (want_dot) {
            const float xn = (aux[r] - amean) * arstd;
            s1 += (double)v;
            s2 += (double)v * (double)xn;
        }
    };
    if (m0 + BM <= M && n0 + BN <= N) {        // interior tile: no per-element guards, the 16 stores stream out
#pragma unroll
        for (int r = 0; r < 16; ++r) emit(r, m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < M && cok) emit(r, row);
        }
    }
    GEMM_CLK(7);
    if (want_st || want_dot) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (lk == 0) { red[wave][0][li] = s1; red[wave][1][li] = s2; }
        __syncthreads();
        if (wave < 2 && lk == 0 && cok) {      // waves 0,1 own columns wn = 0 / 32; add the wm = 32 partner
            const double t1 = red[wave][0][li] + red[wave + 2][0][li];
            const double t2 = red[wave][1][li] + red[wave + 2][1][li];
            if (pr.parts) {                     // one partial row per row tile: [gridDim.x][2][N]
                pr.parts[((size_t)bx * 2 + 0) * N + col] = t1;
                pr.parts[((size_t)bx * 2 + 1) * N + col] = t2;
            } else {
                const size_t po = (size_t)(bx % NSTRIPE) * pr.st_ss + col;
                atomicAdd((want_st ? pr.st_sum : pr.dot_sum) + po, t1);
                atomicAdd((want_st ? pr.st_sq : pr.dot_prod) + po, t2);
            }
        }
    }
    GEMM_CLK(5);
#ifdef CAL_GEMM_CLOCKS
    if (threadIdx.x == 0 && bz == 0 && bx + nbx * by < 2048) g_gemm_blk[2 * (bx + nbx * by) + 1] = wall_clock64();
#endif
}

template <bool A_KC, bool B_KC, int XA, int XB>
__global__ void __launch_bounds__(256) k_gemm(const GemmArgs a, int vecA, int vecB) {
    __shared__ __attribute__((aligned(16))) float smem[GemmSmem<A_KC, B_KC>::FLOATS];
    warm_kernargs<sizeof(GemmArgs) + 16>();             // (common.hpp: the argument segment in one round of scalar loads)
    gemm_block<A_KC, B_KC, XA, XB>(a, vecA, vecB, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, smem);
}

// Two independent GEMM launches in one grid (1-D block index): blocks [0, n1) run problem set 1 (template
// parameters *1), the rest problem set 2.  Used for the dX / dW pair of every layer's backward: both read
// the same dZ, neither depends on the other, and alone each leaves half of the chip idle (230 tiles on 256
// CUs at one ~7 us workgroup each) -- run back to back they cost two kernel latencies, together about one.
struct DualGrid { int gx1, gy1, n1; int gx2, gy2; };
template <bool A1, bool B1, int XA1, int XB1, bool A2, bool B2, int XA2, int XB2>
__global__ void __launch_bounds__(256) k_gemm_dual(const GemmArgs a1, int vecA1, int vecB1, const GemmArgs a2, int vecA2,
                                                   int vecB2, const DualGrid g) {
    constexpr int F1 = GemmSmem<A1, B1>::FLOATS, F2 = GemmSmem<A2, B2>::FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[F1 > F2 ? F1 : F2];
    warm_kernargs<2 * sizeof(GemmArgs) + 48>();
    int b = blockIdx.x;
    if (b < g.n1) {
        gemm_block<A1, B1, XA1, XB1>(a1, vecA1, vecB1, b % g.gx1, (b / g.gx1) % g.gy1, b / (g.gx1 * g.gy1), g.gx1, smem);
    } else {
        b -= g.n1;
        gemm_block<A2, B2, XA2, XB2>(a2, vecA2, vecB2, b % g.gx2, (b / g.gx2) % g.gy2, b / (g.gx2 * g.gy2), g.gx2, smem);
    }
}

__global__ void k_splitk_reduce(const float* __restrict__ part, float* __restrict__ out, int64_t n, int S,
                                int accumulate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f;
    int z = 0;
    for (; z + 1 < S; z += 2) { s0 += part[(size_t)z * n + i]; s1 += part[(size_t)(z + 1) * n + i]; }
    if (z < S) s0 += part[(size_t)z * n + i];
    out[i] = accumulate ? out[i] + (s0 + s1) : (s0 + s1);
}

// row tiles of a node-level GEMM (= partial statistic rows its epilogue writes): 128-row tiles on the throughput kernel
// hasC = false: a statistics-only launch (C == nullptr: the dot sums of bn_feat's backward) -- the weight-resident kernel does not take
// those (launch_gemm_wres), so its partial rows must not be sized for it (advisor, round 5: such a step failed with -2 at F, H in {128, 256})
int gemm_row_tiles(int M, int N, int K, bool hasC) { return (hasC && gemm_wres_rows(M, N, K)) ? gemm_wres_parts() : gemm_big_rows(M, K) ? cdiv(M, 128) : cdiv(M, BM); }

template <bool A_KC, bool B_KC, int XA, int XB>
static void launch_one(const GemmArgs& a, dim3 grid, int vecA, int vecB, hipStream_t stream) {
    hipLaunchKernelGGL((k_gemm<A_KC, B_KC, XA, XB>), grid, dim3(256), 0, stream, a, vecA, vecB);
}

int launch_gemm(bool transA, bool transB, const GemmArgs& a, int nbatch, hipStream_t stream);

// alignment flags and operand-transform classes of a batch (0 = plain, 1 = BN, 2 = row scale + BN)
static int gemm_classify(bool a_kc, bool b_kc, const GemmArgs& a, int nbatch, int& vecA, int& vecB, int& xa, int& xb) {
    vecA = (a.lda % 4 == 0); vecB = (a.ldb % 4 == 0);
    xa = -1; xb = -1;
    for (int b = 0; b < nbatch; ++b) {
        vecA = vecA && aligned16(a.p[b].A);
        vecB = vecB && aligned16(a.p[b].B);
        const int ma = a.p[b].xa.has_bn ? (a.p[b].xa.rs ? 2 : 1) : 0, mb = a.p[b].xb.has_bn ? (a.p[b].xb.rs ? 2 : 1) : 0;
        if ((a.p[b].xa.rs && !a.p[b].xa.has_bn) || (a.p[b].xb.rs && !a.p[b].xb.has_bn)) { set_error("launch_gemm: row scale without BN is not instantiated"); return 2; }
        if ((xa >= 0 && xa != ma) || (xb >= 0 && xb != mb)) { set_error("launch_gemm: mixed operand transforms in one batch"); return 2; }
        xa = ma; xb = mb;
        if ((ma && a_kc) || (mb && b_kc)) {
            if (a.kchunk > XMAX) { set_error("launch_gemm: BN-transformed operand wider than %d", XMAX); return 2; }
        }
    }
    return 0;
}

// dX = dZ W^T (NT, set `ax`, first in block order: it is on the critical path) together with
// dW = op(X)^T dZ (TN with the BN / row-scale transform on X, set `aw`, usually split-K) in ONE launch.
int launch_gemm_dual(const GemmArgs& ax, int nbx, const GemmArgs& aw, int nbw, hipStream_t stream) {
    if (ax.M == 0 || ax.N == 0 || nbx == 0) return launch_gemm(true, false, aw, nbw, stream);
    if (aw.M == 0 || aw.N == 0 || nbw == 0) return launch_gemm(false, true, ax, nbx, stream);
    if (gemm_wres_rows(ax.M, ax.N, ax.K) || gemm_wres_grad(aw.M, aw.N, aw.K)) {     // weight-resident kernels: one launch each
        if (int rc = launch_gemm(false, true, ax, nbx, stream)) return rc;
        return launch_gemm(true, false, aw, nbw, stream);
    }
    if (int r = launch_gemm_big_dual(ax, nbx, aw, nbw, stream)) return r < 0 ? 2 : 0;
    if (gemm_big_rows(ax.M, ax.K) || gemm_big_grad(aw.M, aw.N, aw.K)) {      // one of the two alone
        if (int rc = launch_gemm(false, true, ax, nbx, stream)) return rc;
        return launch_gemm(true, false, aw, nbw, stream);
    }
    int vax, vbx, xax, xbx, vaw, vbw, xaw, xbw;
    if (int rc = gemm_classify(true, true, ax, nbx, vax, vbx, xax, xbx)) return rc;
    if (int rc = gemm_classify(false, false, aw, nbw, vaw, vbw, xaw, xbw)) return rc;
    if (xax != 0 || xbx != 0 || xbw != 0) { set_error("launch_gemm_dual: operand-transform combination not instantiated"); return 2; }
    DualGrid g;
    g.gx1 = cdiv(ax.M, BM); g.gy1 = cdiv(ax.N, BN); g.n1 = g.gx1 * g.gy1 * nbx * ax.nsplit;
    g.gx2 = cdiv(aw.M, BM); g.gy2 = cdiv(aw.N, BN);
    const int n2 = g.gx2 * g.gy2 * nbw * aw.nsplit;
    const dim3 grid(g.n1 + n2);
    if (xaw == 0) hipLaunchKernelGGL((k_gemm_dual<true, true, 0, 0, false, false, 0, 0>), grid, dim3(256), 0, stream, ax, vax, vbx, aw, vaw, vbw, g);
    else if (xaw == 1) hipLaunchKernelGGL((k_gemm_dual<true, true, 0, 0, false, false, 1, 0>), grid, dim3(256), 0, stream, ax, vax, vbx, aw, vaw, vbw, g);
    else hipLaunchKernelGGL((k_gemm_dual<true, true, 0, 0, false, false, 2, 0>), grid, dim3(256), 0, stream, ax, vax, vbx, aw, vaw, vbw, g);
    CAL_CHECK_LAUNCH("k_gemm_dual");
    return 0;
}

int launch_gemm(bool transA, bool transB, const GemmArgs& a, int nbatch, hipStream_t stream) {
    if (a.M == 0 || a.N == 0 || nbatch == 0) return 0;
    if (int r = launch_gemm_wres(transA, transB, a, nbatch, stream)) return r < 0 ? 2 : 0;
    if (!transA && gemm_wres_rows(a.M, a.N, a.K))
        for (int b = 0; b < nbatch; ++b)
            if (a.p[b].parts && a.p[b].C) { set_error("launch_gemm: statistics rows were sized for the weight-resident kernel, which cannot take this launch"); return 2; }
    if (int r = launch_gemm_big(transA, transB, a, nbatch, stream)) return r < 0 ? 2 : 0;
    if (!transA && gemm_big_rows(a.M, a.K))
        for (int b = 0; b < nbatch; ++b)
            if (a.p[b].parts) { set_error("launch_gemm: statistics rows were sized for the 128-row kernel, which cannot take this launch"); return 2; }
    int vecA, vecB, xa, xb;
    const bool a_kc = !transA, b_kc = transB;
    if (int rc = gemm_classify(a_kc, b_kc, a, nbatch, vecA, vecB, xa, xb)) return rc;
    dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), nbatch * a.nsplit);
    bool ok = true;
    if (a_kc && !b_kc) {          // NN
        if (xb != 0) ok = false;
        else if (xa == 0) launch_one<true, false, 0, 0>(a, grid, vecA, vecB, stream);
        else if (xa == 1) launch_one<true, false, 1, 0>(a, grid, vecA, vecB, stream);
        else launch_one<true, false, 2, 0>(a, grid, vecA, vecB, stream);
    } else if (a_kc && b_kc) {    // NT
        if (xb != 0 || xa == 2) ok = false;
        else if (xa == 0) launch_one<true, true, 0, 0>(a, grid, vecA, vecB, stream);
        else launch_one<true, true, 1, 0>(a, grid, vecA, vecB, stream);
    } else if (!a_kc && !b_kc) {  // TN
        if (xb == 0 && xa == 0) launch_one<false, false, 0, 0>(a, grid, vecA, vecB, stream);
        else if (xb == 0 && xa == 1) launch_one<false, false, 1, 0>(a, grid, vecA, vecB, stream);
        else if (xb == 0 && xa == 2) launch_one<false, false, 2, 0>(a, grid, vecA, vecB, stream);
        else if (xb == 1 && xa == 0) launch_one<false, false, 0, 1>(a, grid, vecA, vecB, stream);
        else ok = false;
    } else {                      // TT
        if (xa != 0 || xb != 0) ok = false;
        else launch_one<false, true, 0, 0>(a, grid, vecA, vecB, stream);
    }
    if (!ok) { set_error("launch_gemm: operand-transform combination (%d,%d) not instantiated for this layout", xa, xb); return 2; }
    CAL_CHECK_LAUNCH("k_gemm");
    return 0;
}

// split-K factor for a weight-gradient GEMM: aim at ~256 workgroups, >= 4 K tiles per slice
int splitk_for(int64_t M, int64_t N, int64_t K, int nbatch) {
    if (K < (1ll << 31) && gemm_wres_grad((int)M, (int)N, (int)K)) return gemm_wres_grad_splits((int)K, nbatch);
    if (K < (1ll << 31) && gemm_big_grad((int)M, (int)N, (int)K)) return gemm_big_grad_splits((int)K);
    int64_t tiles = (int64_t)cdiv(M, BM) * cdiv(N, BN) * nbatch;
    if (tiles >= 128 || K <= 8 * BK) return 1;
    int64_t s = 256 / tiles;
    int64_t maxs = K / (4 * BK);
    if (s > maxs) s = maxs;
    if (s > 128) s = 128;
    return (int)(s < 1 ? 1 : s);
}

void gemm_set_split(GemmArgs& a, int S) {
    int kchunk = (int)((((int64_t)a.K + S - 1) / S + BK - 1) / BK * BK);
    if (kchunk == 0) kchunk = BK;
    // a slice just over PRE_T tiles would fall back to the streaming loop: cut it at PRE_T tiles instead
    // (not for the direct-operand gradient kernel, gemm_wres.hip: its slab count was sized as S, and a shorter chunk would make
    //  more slices than slabs -- 375 for 256 at K = 47 950 nodes, a write past the workspace found by tests/tools/fuzz_gemm_wres.py)
    if (S > 1 && kchunk > PRE_T * BK && kchunk <= (PRE_T + 2) * BK && !gemm_wres_grad(a.M, a.N, a.K)) kchunk = PRE_T * BK;
    a.kchunk = kchunk;
    a.nsplit = a.K == 0 ? 1 : cdiv(a.K, kchunk);
}

}  // namespace cal

using namespace cal;

#ifdef CAL_GEMM_CLOCKS
CAL_EXPORT int cal_debug_gemm_clocks(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cal::g_gemm_clk), sizeof(long long) * 16);
}
CAL_EXPORT int cal_debug_gemm_blocks(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cal::g_gemm_blk), sizeof(long long) * 4096);
}
#endif

// experiment hook: same contract as cal_gemm (transA must be 0, no split-K) through the K-split kernel
CAL_EXPORT int cal_gemm_ks(int transB, const float* A, const float* B, float* C, const float* bias, int relu,
                           int64_t M, int64_t N, int64_t K, void* stream_) {
    GemmArgs a = {};
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.lda = (int)K; a.ldb = (int)(transB ? K : N); a.ldc = (int)N;
    a.relu = relu; a.kchunk = (int)K; a.nsplit = 1;
    a.p[0].A = A; a.p[0].B = B; a.p[0].bias = bias; a.p[0].C = C;
    return launch_gemm_ks(transB != 0, a, 1, (hipStream_t)stream_);
}

CAL_EXPORT int64_t cal_gemm_ws(int64_t M, int64_t N, int64_t K) {
    int s = splitk_for(M, N, K, 1);
    return s > 1 ? (int64_t)(s + 1) * M * N : 0;
}

// C[M,N] = op(A) op(B) (+ bias[N]) (ReLU), row-major, ldc = N.
//   transA = 0: A is [M,K]   transA = 1: A is stored [K,M]
//   transB = 0: B is [K,N]   transB = 1: B is stored [N,K]
// ws: cal_gemm_ws(M,N,K) floats (split-K partials; may be null when that is 0).
CAL_EXPORT int cal_gemm(int transA, int transB, const float* A, const float* B, float* C, const float* bias, int relu,
                        float* ws, int64_t M, int64_t N, int64_t K, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (M == 0 || N == 0) return 0;
    CAL_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "sizes out of range");
    int S = (bias || relu) ? 1 : splitk_for(M, N, K, 1);
    CAL_REQUIRE(S == 1 || ws != nullptr, "split-K workspace missing");
    GemmArgs a = {};
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.lda = (int)(transA ? M : K); a.ldb = (int)(transB ? K : N); a.ldc = (int)N;
    a.relu = relu;
    gemm_set_split(a, S);
    a.p[0].A = A; a.p[0].B = B; a.p[0].bias = bias;
    a.p[0].C = a.nsplit > 1 ? ws : C;
    int rc = launch_gemm(transA != 0, transB != 0, a, 1, stream);
    if (rc) return rc;
    if (a.nsplit > 1) {
        hipLaunchKernelGGL(k_splitk_reduce, dim3(cdiv(M * N, 256)), dim3(256), 0, stream, ws, C, M * N, a.nsplit, 0);
        CAL_CHECK_LAUNCH("k_splitk_reduce");
    }
    return 0;
}
