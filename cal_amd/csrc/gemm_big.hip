// Throughput variant of the fp32 MFMA GEMM (gemm.hip) for node-level products with tens of thousands of rows
// ([160k,256] x [256,256], 21 GFLOP each, 15 per train step):
//
//   workgroup tile 128 x 128, K step 32, 4 waves (2 x 2), each wave a 64 x 64 block = 2 x 2 accumulators of the
//   32x32x2 f32 MFMA.  One k-pair now feeds FOUR MFMAs from four ds_read_b32 (the 64 x 64 kernel: two reads per
//   MFMA), a K tile is 64 MFMAs = 4096 matrix-core cycles per wave between barriers (there: 1024), and an operand
//   slab is fetched by half as many workgroups.  The 64 x 64 kernel stays the choice below ~16k rows, where its
//   4x larger grid is what fills the 256 CUs (config 2: 115 row tiles).
//
// Same contract as gemm_block (engine.hpp GemmArgs / GemmProb): BatchNorm (+ row scale) applied to operand A while it
// is staged, bias / ReLU epilogue, per-column statistics or BN-backward dot sums in fp64 (partial row per ROW TILE OF
// 128 -- gemm_row_tiles() accounts for it -- or atomics), split-K slices into slabs.  Layouts: NN, NT (k-contiguous
// operands, K % 32 == 0) and TN (weight gradients: K = node rows, any K, zero-filled tail).
#include "engine.hpp"

namespace cal {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace big {

constexpr int T = 128, BK = 32;
constexpr int NQ = T * BK / 4 / 256;      // float4 per thread per operand tile (4)
constexpr int LDT = T + 1;                // LDS row stride of tiles filled by transposing scalar stores (bank = 4 kq + mn)
constexpr int LDD = T + 4;                // ... by direct 16 B stores
constexpr int XMAX = 512;                 // BN table width (k range of a k-contiguous operand, or the 128 tile columns)

__device__ __forceinline__ void pin4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// Operand tile T[mn][k] -> registers.  KC: memory [mn][k] (k contiguous); !KC: memory [k][mn].  Always 16 B loads from
// clamped (valid) addresses: rows / columns past the end hold copies of valid data that only reach accumulator
// entries the epilogue never stores; k rows past the end of a !KC operand are zeroed at store time.
template <bool KC>
__device__ __forceinline__ void load(float4 (&r)[NQ], const float* __restrict__ p, int ld, int mn0, int mn_end, int k0, int k_end) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int f = threadIdx.x + q * 256;
        if (KC) {
            const int mn = f / (BK / 4), k = (f % (BK / 4)) * 4;
            r[q] = *reinterpret_cast<const float4*>(p + (size_t)min(mn0 + mn, mn_end - 1) * ld + k0 + k);
        } else {
            const int mn = (f % (T / 4)) * 4, k = f / (T / 4);
            r[q] = *reinterpret_cast<const float4*>(p + (size_t)min(k0 + k, k_end - 1) * ld + min(mn0 + mn, mn_end - 4));
        }
    }
}

// XF: 0 plain, 1 BN scale/shift on the feature axis, 2 per-storage-row scale then BN (as gemm.hip tile_store)
template <bool KC, int XF>
__device__ __forceinline__ void store(const float4 (&r)[NQ], float* __restrict__ s, int mn0, int mn_end, int k0, int k_end,
                                      int kb, const float* __restrict__ rsp, int rs_stride, const float* sc, const float* sh) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int f = threadIdx.x + q * 256;
        const int mn = KC ? f / (BK / 4) : (f % (T / 4)) * 4;
        const int k = KC ? (f % (BK / 4)) * 4 : f / (T / 4);
        float v[4] = {r[q].x, r[q].y, r[q].z, r[q].w};
        if (XF > 0) {
            float rs = 1.f;
            if (XF == 2) rs = rsp[(size_t)(KC ? min(mn0 + mn, mn_end - 1) : min(k0 + k, k_end - 1)) * rs_stride];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int fi = KC ? (k0 + k + j - kb) : (mn + j);
                v[j] = fmaf(XF == 2 ? rs * v[j] : v[j], sc[fi], sh[fi]);
            }
        }
        if (KC) {
            s[(k + 0) * LDT + mn] = v[0]; s[(k + 1) * LDT + mn] = v[1];
            s[(k + 2) * LDT + mn] = v[2]; s[(k + 3) * LDT + mn] = v[3];
        } else {
            const bool ok = k0 + k < k_end;
            *reinterpret_cast<float4*>(s + k * LDD + mn) = ok ? make_float4(v[0], v[1], v[2], v[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

template <bool A_KC, bool B_KC>
struct Smem {
    static constexpr int LDA = A_KC ? LDT : LDD, LDB = B_KC ? LDT : LDD;
    static constexpr int SA = BK * LDA, SB = BK * LDB;
    static constexpr int A = 0;
    static constexpr int B = A + 2 * SA;
    static constexpr int SC = (B + 2 * SB + 3) / 4 * 4;
    static constexpr int SH = SC + XMAX;
    static constexpr int RED = SH + XMAX;               // doubles (8-byte aligned: every term above is even)
    static constexpr int FLOATS = RED + 2 * 4 * 2 * 2 * 32;
};

template <bool A_KC, bool B_KC, int XA>
__device__ __forceinline__ void block(const GemmArgs& a, int bx, int by, int bz, float* __restrict__ smem, bool phase_slot) {
    using SM = Smem<A_KC, B_KC>;
    constexpr int LDA = SM::LDA, LDB = SM::LDB, SA = SM::SA, SB = SM::SB;
    float* As = smem + SM::A;
    float* Bs = smem + SM::B;
    float* xsc = smem + SM::SC;
    float* xsh = smem + SM::SH;
    double (*red)[2][2][32] = reinterpret_cast<double (*)[2][2][32]>(smem + SM::RED);

    const int batch = bz / a.nsplit, split = bz % a.nsplit;
    const GemmProb& pr = a.p[batch];
    const int M = a.M, N = a.N, K = a.K;
    const int m0 = bx * T, n0 = by * T;
    const int kb = split * a.kchunk, ke = min(K, kb + a.kchunk);
    float* C = pr.C ? pr.C + (size_t)split * M * a.ldc : nullptr;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int li = lane & 31, lk = lane >> 5;

    // Two register sets of operand tiles: tile t+2 is requested while tile t multiplies, and tile t+1 (requested a full
    // tile earlier) is staged into the other LDS stage in the MIDDLE of tile t's MFMA stream -- the LDS stores issue
    // between the last 16 MFMAs instead of after them, so a wave's matrix pipe never idles on its own store phase
    // (with the store after the MFMAs both resident workgroups of a CU ran ~1.5 us of every 5.9 us tile without any
    // MFMA in flight: 70 TF; phase timestamps in scripts/big_gemm_probe.py).
    const int nt = (ke - kb + BK - 1) / BK;
    // The two workgroups resident on a CU would otherwise run in lockstep -- all 512 load their first tiles together,
    // multiply together and burst their 64 KB of C together (2 + 7 us per ~50 us round with no MFMA in flight anywhere).
    // The second resident set (linear ids 256..511, dispatched into the CUs' second slots) starts half a K loop late
    // once; every later workgroup inherits the slot's phase, so one workgroup's epilogue / prologue always runs under
    // its neighbour's MFMA stream.
    if (phase_slot) {
        const long long t0 = wall_clock64();
        const long long wait = (long long)nt * 120;      // ~1.2 us per K tile, in 100 MHz ticks
        while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
    float4 ra0[NQ], rb0[NQ], ra1[NQ], rb1[NQ];
    load<A_KC>(ra0, pr.A, a.lda, m0, M, kb, ke);
    load<B_KC>(rb0, pr.B, a.ldb, n0, N, kb, ke);
    if (nt > 1) {
        load<A_KC>(ra1, pr.A, a.lda, m0, M, kb + BK, ke);
        load<B_KC>(rb1, pr.B, a.ldb, n0, N, kb + BK, ke);
    }
    if (XA > 0) {
        const int cnt = A_KC ? (ke - kb) : min(T, M - m0);
        const int c0 = A_KC ? kb : m0;
        for (int t = threadIdx.x; t < cnt; t += 256) {
            bn_scale_shift(pr.xa.bn, c0 + t, xsc[t], xsh[t]);
            if (pr.xa.bn.update && by == 0 && split == 0 && (A_KC ? bx == 0 : true)) bn_update_running(pr.xa.bn, c0 + t);
        }
        if (!A_KC) for (int t = cnt + threadIdx.x; t < T; t += 256) { xsc[t] = 0.f; xsh[t] = 0.f; }
        __syncthreads();
    }
    store<A_KC, XA>(ra0, As, m0, M, kb, ke, kb, pr.xa.rs, pr.xa.rs_stride, xsc, xsh);
    store<B_KC, 0>(rb0, Bs, n0, N, kb, ke, kb, nullptr, 0, nullptr, nullptr);
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][0][i] = 0.f; acc[0][1][i] = 0.f; acc[1][0][i] = 0.f; acc[1][1][i] = 0.f; }

    // la/lb: the set tile t came from (free again) <- tile t+2;  sa/sb: the set holding tile t+1 -> LDS stage (t+1) & 1
    auto tile = [&](int t, float4 (&la)[NQ], float4 (&lb)[NQ], const float4 (&sa)[NQ], const float4 (&sb)[NQ]) {
        const int st = t & 1, k0 = kb + t * BK;
        // One basic block per tile (no branches: past the end the loads re-fetch tile 0 and the staging writes a stage
        // nobody reads), so the scheduler may lace every non-MFMA instruction into the MFMA stream; the
        // sched_group_barrier patterns below say how: a wave that issues 8 LDS reads, or 100 VALU + 12 LDS writes of a
        // staging phase, in one piece leaves its matrix pipe idle for that long (measured: 3.3 us per tile for 1.72 us
        // of MFMA with one workgroup per CU).
        const int kl = t + 2 < nt ? k0 + 2 * BK : kb;
        const float* as = As + st * SA + wm + li;
        const float* bs = Bs + st * SB + wn + li;
        float av[2][4][2], bv[2][4][2];
        auto read_ops = [&](int g, float (&ao)[4][2], float (&bo)[4][2]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = 2 * (4 * g + i) + lk;
                ao[i][0] = as[kk * LDA]; ao[i][1] = as[kk * LDA + 32];
                bo[i][0] = bs[kk * LDB]; bo[i][1] = bs[kk * LDB + 32];
            }
        };
        auto mma = [&](int g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][i][0], bv[g & 1][i][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][i][0], bv[g & 1][i][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][i][1], bv[g & 1][i][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][i][1], bv[g & 1][i][1], acc[1][1], 0, 0, 0);
            }
        };
        constexpr int MFMA = 0x008, VALU = 0x002, VMEM_RD = 0x020, DS_RD = 0x100, DS_WR = 0x200;
        read_ops(0, av[0], bv[0]);                       // the one LDS latency a tile exposes (right after the barrier)
        __builtin_amdgcn_sched_barrier(0);
        // group 0: + the global loads of tile t+2 and the operand reads of group 1
        load<A_KC>(la, pr.A, a.lda, m0, M, kl, ke);
        load<B_KC>(lb, pr.B, a.ldb, n0, N, kl, ke);
        read_ops(1, av[1], bv[1]);
        mma(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(VALU, 4, 0);
            __builtin_amdgcn_sched_group_barrier(VMEM_RD, 1, 0);
            __builtin_amdgcn_sched_group_barrier(DS_RD, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(MFMA, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
        // groups 1, 2: + the operand reads of the next group
#pragma unroll
        for (int g = 1; g < 3; ++g) {
            read_ops(g + 1, av[(g + 1) & 1], bv[(g + 1) & 1]);
            mma(g);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(MFMA, 1, 0);
                __builtin_amdgcn_sched_group_barrier(VALU, 1, 0);
                __builtin_amdgcn_sched_group_barrier(DS_RD, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(MFMA, 8, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // group 3: + the staging of tile t+1 (requested a whole tile ago) into the other LDS stage
        store<A_KC, XA>(sa, As + (st ^ 1) * SA, m0, M, k0 + BK, ke, kb, pr.xa.rs, pr.xa.rs_stride, xsc, xsh);
        store<B_KC, 0>(sb, Bs + (st ^ 1) * SB, n0, N, k0 + BK, ke, kb, nullptr, 0, nullptr, nullptr);
        mma(3);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(VALU, 6, 0);
            __builtin_amdgcn_sched_group_barrier(DS_WR, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(MFMA, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    for (int t = 0; t < nt; t += 2) {
        tile(t, ra0, rb0, ra1, rb1);
        if (t + 1 < nt) tile(t + 1, ra1, rb1, ra0, rb0);
    }

    // epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const bool want_st = pr.st_sum != nullptr, want_dot = pr.dot_sum != nullptr;
    const bool interior = m0 + T <= M && n0 + T <= N;
    double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0};
    // Everything the epilogue reads comes first, as ONE batch of unconditional loads on clamped rows / columns: the 4 x 16 aux
    // values of the wave's four accumulators, the 2 x 16 row scales (no scale: the aux pointer again, stride 0, value ignored),
    // the BatchNorm constants and the bias.  Per accumulator (16 loads, wait, 16 stores, four times) every tile paid four
    // round trips; with `if (aux_rs) x *= aux_rs[row]` inside the loop every row was a load, a branch and a dependent second
    // load (the two-branch backward, the only caller with row scales: 585 us per branch against 455 us without them).
    float auxv[2][2][16], bvv[2] = {0.f, 0.f}, amean[2] = {0.f, 0.f}, arstd[2] = {0.f, 0.f};
#pragma unroll
    for (int sn = 0; sn < 2; ++sn) {
        const int colc = min(n0 + wn + sn * 32 + li, N - 1);
        if (pr.bias) bvv[sn] = pr.bias[colc];
    }
    if (want_dot) {
        const bool has_rs = pr.aux_rs != nullptr;
        const float* rsp = has_rs ? pr.aux_rs : pr.aux;
        const size_t rstr = has_rs ? (size_t)pr.aux_rs_stride : 0;
        float ars[2][16];
#pragma unroll
        for (int sm = 0; sm < 2; ++sm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t row = (size_t)min(m0 + wm + sm * 32 + 4 * lk + (r & 3) + 8 * (r >> 2), M - 1);
                ars[sm][r] = rsp[row * rstr];
#pragma unroll
                for (int sn = 0; sn < 2; ++sn) auxv[sn][sm][r] = pr.aux[row * N + min(n0 + wn + sn * 32 + li, N - 1)];
            }
#pragma unroll
        for (int sn = 0; sn < 2; ++sn) bn_mean_rstd(pr.aux_bn, min(n0 + wn + sn * 32 + li, N - 1), amean[sn], arstd[sn]);
#pragma unroll
        for (int sm = 0; sm < 2; ++sm)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(ars[sm][r]), "+v"(auxv[0][sm][r]), "+v"(auxv[1][sm][r]));
#pragma unroll
        for (int sm = 0; sm < 2; ++sm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float f = has_rs ? ars[sm][r] : 1.f;
                auxv[0][sm][r] *= f; auxv[1][sm][r] *= f;
            }
    }
    // consume the loads before the guarded stores (else every store waits for the previous one: gemm.hip)
    asm volatile("" :: "v"(bvv[0]), "v"(bvv[1]), "v"(amean[0]), "v"(amean[1]), "v"(arstd[0]), "v"(arstd[1]));
#pragma unroll
    for (int sn = 0; sn < 2; ++sn) {
        const int col = n0 + wn + sn * 32 + li;
        const bool cok = col < N;
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
            const int rbase = m0 + wm + sm * 32 + 4 * lk;
            const float (&aux)[16] = auxv[sn][sm];
            auto emit = [&](int r, int row) {
                float v = acc[sm][sn][r] + bvv[sn];
                if (a.relu) v = fmaxf(v, 0.f);
                if (C) C[(size_t)row * a.ldc + col] = v;
                if (want_st) { s1[sn] += (double)v; s2[sn] += (double)v * (double)v; }
                if (want_dot) {
                    const float xn = (aux[r] - amean[sn]) * arstd[sn];
                    s1[sn] += (double)v;
                    s2[sn] += (double)v * (double)xn;
                }
            };
            if (interior) {
#pragma unroll
                for (int r = 0; r < 16; ++r) emit(r, rbase + (r & 3) + 8 * (r >> 2));
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < M && cok) emit(r, row);
                }
            }
        }
    }
    if (want_st || want_dot) {
#pragma unroll
        for (int sn = 0; sn < 2; ++sn) {
            s1[sn] += __shfl_xor(s1[sn], 32, 64);
            s2[sn] += __shfl_xor(s2[sn], 32, 64);
            if (lk == 0) { red[wave][sn][0][li] = s1[sn]; red[wave][sn][1][li] = s2[sn]; }
        }
        __syncthreads();
        if (wave < 2 && lk == 0) {          // waves 0,1 own columns wn = 0 / 64; add the wm = 64 partner (wave + 2)
#pragma unroll
            for (int sn = 0; sn < 2; ++sn) {
                const int col = n0 + wn + sn * 32 + li;
                if (col < N) {
                    const double t1 = red[wave][sn][0][li] + red[wave + 2][sn][0][li];
                    const double t2 = red[wave][sn][1][li] + red[wave + 2][sn][1][li];
                    if (pr.parts) {             // one partial row per 128-row tile: [row tiles][2][N]
                        pr.parts[((size_t)bx * 2 + 0) * N + col] = t1;
                        pr.parts[((size_t)bx * 2 + 1) * N + col] = t2;
                    } else {
                        atomicAdd((want_st ? pr.st_sum : pr.dot_sum) + col, t1);
                        atomicAdd((want_st ? pr.st_sq : pr.dot_prod) + col, t2);
                    }
                }
            }
        }
    }
}

// workgroup b of a 1-D grid -> (outer, inner): the `inner` tiles of one `outer` index (they share an operand strip) sit 8
// workgroup ids apart, i.e. on the same XCD (ids are dealt round-robin) right after one another; the last outer % 8
// indices keep the plain order
__device__ __forceinline__ void xcd_order(int b, int outer, int inner, int& o, int& i) {
    const int full = (outer >> 3) * 8 * inner;
    if (b < full) { const int grp = b / (8 * inner); o = grp * 8 + (b & 7); i = (b >> 3) % inner; }
    else { const int r = b - full, rem = outer & 7; o = (outer >> 3) * 8 + r % rem; i = r / rem; }
}
template <bool A_KC, bool B_KC, int XA>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_gemm_big(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[Smem<A_KC, B_KC>::FLOATS];
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    // tiles that share an operand strip back to back on one XCD (xcd_order below): the column tiles of a row tile for the
    // node-level products (A_KC), the output tiles of a split-K slice for the weight gradients
    const int per = gridDim.x * gridDim.y;
    int bx, by, bz;
    if (A_KC) {
        bz = lin / per;
        xcd_order(lin - bz * per, gridDim.x, gridDim.y, bx, by);
    } else {
        int tl;
        xcd_order(lin, gridDim.z, per, bz, tl);
        bx = tl % gridDim.x; by = tl / gridDim.x;
    }
    block<A_KC, B_KC, XA>(a, bx, by, bz, smem, lin >= 256 && lin < 512);
}

// dX = dZ W^T (NT) and dW = op(X)^T dZ (TN, split-K) of one layer in one grid.  The LONG tiles go first (a split-K slice is
// 32 K tiles, ~60 us; an NT tile 8 K tiles + its epilogue, ~16 us): 628 slices on 512 resident slots leave a second round that
// is 23 % full, and behind the 2500 NT tiles that round was the tail of the launch (~45 of 455 us with most of the chip idle);
// dispatched first, the slices' stragglers are covered by NT tiles and the launch ends on 16 us tiles.
struct Grid2 { int gx1, gy1, n1; int gx2, gy2, nz2; };
template <int XA2>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_gemm_big_dual(const GemmArgs a1, const GemmArgs a2, const Grid2 g) {
    constexpr int F1 = Smem<true, true>::FLOATS, F2 = Smem<false, false>::FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[F1 > F2 ? F1 : F2];
    int b = blockIdx.x;
    const int tiles = g.gx2 * g.gy2, n2 = tiles * g.nz2;
    if (b >= n2) {
        b -= n2;
        // the column tiles of one row tile read the same dZ rows: back to back on ONE XCD (xcd_order), so the second read is
        // an L2 hit -- in row-tile-major order they were 1250 workgroups apart on different XCDs and the strip came from HBM
        // once per column tile (1.78 GB per launch against 0.53 GB algorithmic)
        const int per = g.gx1 * g.gy1, bz = b / per;
        int rt, ct;
        xcd_order(b - bz * per, g.gx1, g.gy1, rt, ct);
        block<true, true, 0>(a1, rt, ct, bz, smem, false);
    } else {
        // likewise the gx2 * gy2 output tiles of one split-K slice (same 1024-node strips of both operands)
        int z, tl;
        xcd_order(b, g.nz2, tiles, z, tl);
        block<false, false, XA2>(a2, tl % g.gx2, tl / g.gx2, z, smem, false);
    }
}

}  // namespace big

// ---- selection -----------------------------------------------------------------------------------------------
// node-level products (M rows = nodes) on the 128 x 128 kernel: enough row tiles to fill the chip, k-contiguous K whole
bool gemm_big_rows(int M, int K) { return M >= 16384 && K % big::BK == 0 && K >= big::BK && K <= big::XMAX; }
// weight gradients (K = nodes): worth it from 64 x 64 outputs up
bool gemm_big_grad(int M, int N, int K) { return K >= 16384 && M >= 64 && N >= 64 && M % 4 == 0 && N % 4 == 0; }
// split-K slice of such a gradient: 32 K tiles per workgroup (a 128 x 128 slab per ~60 us of matrix-core work); 64 from 64 k rows up,
// where the slices are plenty either way and half as many slabs go through k_finish (config 5: 247 -> 124 MB, 48 -> 35 us)
int gemm_big_grad_splits(int K) { return cdiv(K, K >= 65536 ? 2048 : 1024); }

static bool big_aligned(const GemmArgs& a, int nbatch, bool xb_plain) {
    bool ok = a.lda % 4 == 0 && a.ldb % 4 == 0 && xb_plain;
    for (int b = 0; b < nbatch; ++b) {
        ok = ok && aligned16(a.p[b].A) && aligned16(a.p[b].B) && !a.p[b].xb.has_bn && !a.p[b].xb.rs;
        if (a.p[b].xa.rs && !a.p[b].xa.has_bn) ok = false;
    }
    return ok;
}
static int xa_class(const GemmArgs& a, int nbatch) {
    int x = -1;
    for (int b = 0; b < nbatch; ++b) {
        const int m = a.p[b].xa.has_bn ? (a.p[b].xa.rs ? 2 : 1) : 0;
        if (x >= 0 && x != m) return -1;
        x = m;
    }
    return x;
}

// 1 = launched on the big kernel, 0 = not applicable (caller falls back to gemm.hip), < 0 = error
int launch_gemm_big(bool transA, bool transB, const GemmArgs& a, int nbatch, hipStream_t stream) {
    const bool a_kc = !transA, b_kc = transB;
    if (!a_kc && b_kc) return 0;
    const bool rows = a_kc && gemm_big_rows(a.M, a.K) && a.nsplit == 1;
    const bool grad = !a_kc && !b_kc && gemm_big_grad(a.M, a.N, a.K) && a.kchunk % big::BK == 0;
    if (!rows && !grad) return 0;
    const int xa = xa_class(a, nbatch);
    if (!big_aligned(a, nbatch, true) || xa < 0 || (a_kc && xa > 0 && a.kchunk > big::XMAX) || (!b_kc && a.N % 4 != 0)) {
        for (int b = 0; b < nbatch; ++b)
            if (rows && a.p[b].parts) { set_error("launch_gemm_big: operands of a statistics GEMM must be 16-byte aligned"); return -2; }
        return 0;
    }
    const dim3 grid(cdiv(a.M, big::T), cdiv(a.N, big::T), nbatch * a.nsplit);
    using namespace big;
    if (a_kc && !b_kc) {
        if (xa == 0) hipLaunchKernelGGL((k_gemm_big<true, false, 0>), grid, dim3(256), 0, stream, a);
        else if (xa == 1) hipLaunchKernelGGL((k_gemm_big<true, false, 1>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_gemm_big<true, false, 2>), grid, dim3(256), 0, stream, a);
    } else if (a_kc && b_kc) {
        if (xa == 0) hipLaunchKernelGGL((k_gemm_big<true, true, 0>), grid, dim3(256), 0, stream, a);
        else if (xa == 1) hipLaunchKernelGGL((k_gemm_big<true, true, 1>), grid, dim3(256), 0, stream, a);
        else return 0;
    } else {
        if (xa == 0) hipLaunchKernelGGL((k_gemm_big<false, false, 0>), grid, dim3(256), 0, stream, a);
        else if (xa == 1) hipLaunchKernelGGL((k_gemm_big<false, false, 1>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_gemm_big<false, false, 2>), grid, dim3(256), 0, stream, a);
    }
    if (hipGetLastError() != hipSuccess) { set_error("k_gemm_big: launch failed"); return -2; }
    return 1;
}

// NT (ax) + TN (aw) in one grid; same return convention
int launch_gemm_big_dual(const GemmArgs& ax, int nbx, const GemmArgs& aw, int nbw, hipStream_t stream) {
    if (!(gemm_big_rows(ax.M, ax.K) && ax.nsplit == 1 && gemm_big_grad(aw.M, aw.N, aw.K) && aw.kchunk % big::BK == 0)) return 0;
    const int xw = xa_class(aw, nbw);
    if (!big_aligned(ax, nbx, true) || !big_aligned(aw, nbw, true) || xa_class(ax, nbx) != 0 || xw < 0) {
        for (int b = 0; b < nbx; ++b)
            if (ax.p[b].parts) { set_error("launch_gemm_big_dual: operands of a statistics GEMM must be 16-byte aligned"); return -2; }
        return 0;
    }
    big::Grid2 g;
    g.gx1 = cdiv(ax.M, big::T); g.gy1 = cdiv(ax.N, big::T); g.n1 = g.gx1 * g.gy1 * nbx;
    g.gx2 = cdiv(aw.M, big::T); g.gy2 = cdiv(aw.N, big::T); g.nz2 = nbw * aw.nsplit;
    const dim3 grid(g.n1 + g.gx2 * g.gy2 * nbw * aw.nsplit);
    if (xw == 0) hipLaunchKernelGGL((big::k_gemm_big_dual<0>), grid, dim3(256), 0, stream, ax, aw, g);
    else if (xw == 1) hipLaunchKernelGGL((big::k_gemm_big_dual<1>), grid, dim3(256), 0, stream, ax, aw, g);
    else hipLaunchKernelGGL((big::k_gemm_big_dual<2>), grid, dim3(256), 0, stream, ax, aw, g);
    if (hipGetLastError() != hipSuccess) { set_error("k_gemm_big_dual: launch failed"); return -2; }
    return 1;
}

}  // namespace cal
