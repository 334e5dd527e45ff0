// Weight-resident fp32 MFMA GEMMs for node-level products with tens of thousands of rows and a hidden width of 128 / 256
// ([160k,256] x [256,256] forward, dX = dZ W^T and dW = X^T dZ backward; gcn_conv.py:75).
//
// Round 5.  The 128 x 128 tile kernel of gemm_big.hip stages both operands through LDS with a barrier per 32-wide k tile and
// ran at 0.60-0.65 matrix-core busy: a workgroup's prologue / epilogue and its barrier phases leave the pipe idle, two
// resident workgroups do not cover each other (profiles/r4/pmc_issue_ba5000_*, profiles/r4/DESIGN_r4.md section 8).  These kernels have no
// barrier and no operand staging inside the loop:
//
//   k_wres (C = op(A) W, NN / NT).  A persistent workgroup keeps one 128-column half of the WEIGHT in LDS for its whole life
//     (k-contiguous rows, stride K + 4 floats: a ds_read_b128 hands a lane four consecutive k of its column, conflict-free)
//     and every wave walks its own 32-row blocks of the NODE operand, which it reads STRAIGHT from global memory into MFMA
//     operand registers: lane (row, k half) takes 64 contiguous bytes of its row per 32-wide k step (any bijection of k onto
//     (instruction, lane half) is a valid reduction order as long as both operands use it).  Per 16 MFMAs of a wave: four
//     ds_read_b128, no barrier, no LDS write.  The epilogue of row block i (C stores, bias / ReLU, fp64 column statistics or
//     the BatchNorm-backward dot sums) rides in the MFMA stream of block i + 1 (two accumulator sets), so a wave's matrix
//     stream never pauses; the statistics stay in registers for the life of the workgroup and leave as ONE partial row.
//     One wave per SIMD (256 threads, 1 workgroup per CU: 133 KB of LDS), accumulators + operand ring in the 512-register file.
//   k_tn (dW = op(X)^T dZ, k = node rows).  Both operands straight from global memory: a dwordx2 / dwordx4 load of a node row
//     hands lane i the columns 2i..2i+1 / 4i..4i+3, i.e. the A / B registers of 2 / 4 MFMAs whose output rows / columns are
//     interleaved; no LDS at all, one 256 x 256 slab per node range (k_finish sums the slabs).
//
// Measured on MI355X at the config-5 shape (scripts/micro/gemm_wres.hip, profiles/r5/micro_gemm_wres.txt): the matrix cores
// clock down under a dense fp32 MFMA stream on real data (1.95-2.2 GHz against 2.4), which caps every formulation.
// Same contract as gemm_block (engine.hpp GemmArgs / GemmProb).
#include "engine.hpp"
#include <type_traits>
#include <cstdlib>

namespace cal {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace wres {

constexpr int NW = 4;                 // waves per workgroup, one per SIMD
constexpr int NT = NW * 64;
constexpr int GRID = 256;             // persistent workgroups (one per CU of an MI355X)
constexpr int PARTS = 256;            // partial statistics rows per problem (gemm_row_tiles)
constexpr int NBUF = 4;               // operand ring: the loads of k step s + 3 are in flight while step s multiplies

// EPI: 0 none, 1 column sums / sums of squares of C, 2 BatchNorm-backward dot sums against aux, 3 the same with a row scale on aux
template <bool B_KC, int XA, int KS, int EPI>
__global__ void __launch_bounds__(NT) k_wres(const GemmArgs a, const int nbatch) {
    constexpr int K = KS * 32, LDW = K + 4;
    constexpr int VPG = 64 / (4 * KS);        // C values of the previous block emitted per 16-MFMA group
    constexpr int VPS = 4 * VPG;              // ... per k step
    static_assert(KS % NBUF == 0, "ring slots must line up across row blocks");
    __shared__ __attribute__((aligned(16))) float Ws[128 * LDW];
    __shared__ __attribute__((aligned(16))) float tsc[K], tsh[K];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m = lane & 31, kq = lane >> 5;
    // workgroup -> (problem, column half, member): the halves of one problem sit 8 workgroup ids apart, i.e. on the same XCD
    // (ids are dealt round-robin) and walk the same rows at the same time: the second read of a row is an L2 hit
    const int nh = a.N >> 7, ncol = nbatch * nh;
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int group = idx % ncol, member = (idx / ncol) * 8 + xcd, members = ((int)gridDim.x >> 3) / ncol * 8;
    const int batch = group / nh, n0 = (group % nh) * 128;
    const GemmProb& pr = a.p[batch];

    if (B_KC) {
        for (int i = tid; i < 128 * K / 4; i += NT) {
            const int n = i / (K / 4), k4 = i % (K / 4);
            *reinterpret_cast<float4*>(Ws + n * LDW + 4 * k4) = *reinterpret_cast<const float4*>(pr.B + (size_t)(n0 + n) * a.ldb + 4 * k4);
        }
    } else {
        // lanes along k: the transposing LDS stores of a wave fall on 32 consecutive banks (lanes along n: on two)
        for (int i = tid; i < 32 * K; i += NT) {
            const int k = i % K, n4 = i / K;
            const float4 v = *reinterpret_cast<const float4*>(pr.B + (size_t)k * a.ldb + n0 + 4 * n4);
            Ws[(4 * n4 + 0) * LDW + k] = v.x; Ws[(4 * n4 + 1) * LDW + k] = v.y;
            Ws[(4 * n4 + 2) * LDW + k] = v.z; Ws[(4 * n4 + 3) * LDW + k] = v.w;
        }
    }
    if (XA > 0) {
        for (int i = tid; i < K; i += NT) {
            bn_scale_shift(pr.xa.bn, i, tsc[i], tsh[i]);
            if (pr.xa.bn.update && n0 == 0 && member == 0) bn_update_running(pr.xa.bn, i);
        }
    }
    __syncthreads();

    const int stride = members * NW;
    const int nblk = (a.M + 31) >> 5;
    int rb = member * NW + wave;
    int wofs = (m * LDW + 16 * kq) / 4;           // laundered once per row block: the fragments are loop-invariant, and hipcc
                                                  // would hoist all 16 * KS ds_read_b128 out of the loop and spill them
    constexpr bool want_st = EPI == 1, want_dot = EPI >= 2, has_rs = EPI == 3;
    // everything the loop reads from the argument block, once: left to itself hipcc re-reads the fields of a.p[batch] (a dynamic
    // index into the kernel-argument segment) with s_load + s_waitcnt lgkmcnt(0) inside the MFMA stream -- 500 scalar loads
    // and 690 waits in the dot-sum instantiation, 259 us per launch against 190 us without the sums
    const int M = a.M, ldc = a.ldc, lda = a.lda, auxld = a.N, relu = a.relu;
    const float* const Ap = pr.A;
    const float* const xrs = pr.xa.rs; const int xrs_stride = pr.xa.rs_stride;
    const float* const aux_rs = pr.aux_rs; const int aux_rs_stride = pr.aux_rs_stride;
    float bvv[4], amean[4], arstd[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int col = n0 + nb * 32 + m;
        bvv[nb] = pr.bias ? pr.bias[col] : 0.f;
        amean[nb] = 0.f; arstd[nb] = 0.f;
        if (want_dot) bn_mean_rstd(pr.aux_bn, col, amean[nb], arstd[nb]);
    }
    float* const Cb = pr.C + n0 + m;                 // (launch_gemm_wres takes only launches that store C)
    const float* const auxb = want_dot ? pr.aux + n0 + m : nullptr;

    float4 abuf[NBUF][4];
    float ars[NBUF];                              // row scale of the operand row (XA == 2), rides with the ring
    auto issue = [&](int blk, int s, float4 (&dst)[4], float& rs) {
        const size_t row = (size_t)min(blk * 32 + m, M - 1);
        const float* p = Ap + row * lda + s * 32 + 16 * kq;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const float4*>(p + 4 * q);
        if (XA == 2) rs = xrs[row * xrs_stride];
    };
    float4 bf[2][4];
    auto readb = [&](int s, int j4, float4 (&dst)[4]) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) dst[nb] = *reinterpret_cast<const float4*>(Ws + 4 * wofs + nb * 32 * LDW + s * 32 + 4 * j4);
    };
    // aux values (and row scales) of the previous block's C entries e = VPS * s .. VPS * s + VPS - 1, NAUX - 1 k steps ahead of
    // use (one step = ~2 us of MFMA work at one wave per SIMD was not enough under load: 266 us per NT + dot-sum launch against
    // 190 us without the sums).  Issued BEFORE the operand loads of the same step: vmcnt counts in order, so waiting for them
    // never waits for younger loads.
    constexpr int NAUX = KS >= 8 ? 4 : 2;
    static_assert(KS % NAUX == 0, "aux slots must line up across row blocks");
    float auxv[NAUX][VPS], auxr[NAUX][VPS];
    auto issue_aux = [&](int blk, int s, float (&dv)[VPS], float (&dr)[VPS]) {
#pragma unroll
        for (int u = 0; u < VPS; ++u) {
            const int e = VPS * s + u, nb = e >> 4, r = e & 15;
            const size_t row = (size_t)min(blk * 32 + 4 * kq + (r & 3) + 8 * (r >> 2), M - 1);
            dv[u] = auxb[row * auxld + nb * 32];
            dr[u] = has_rs ? aux_rs[row * aux_rs_stride] : 1.f;
        }
    };
    // column sums: the 16 values a lane holds per column block and row block in fp32 (f1 / f2, flushed once per row block),
    // everything across row blocks, waves and workgroups in fp64 -- per value in fp64 it was 64 x (2 conversions + add + fma)
    // of half-rate VALU work per block inside the MFMA stream (as in k_gconv_bwd's epilogue, engine_gconv_bwd.hpp)
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    float f1[4] = {0.f, 0.f, 0.f, 0.f}, f2[4] = {0.f, 0.f, 0.f, 0.f};
    auto emit = [&](int nb, float acc, float* cp, size_t ofs, float av, float ar, bool ok) {
        float v = acc + bvv[nb];
        v = relu ? fmaxf(v, 0.f) : v;
        if (ok) {
            cp[ofs] = v;
            if (want_st) { f1[nb] += v; f2[nb] = fmaf(v, v, f2[nb]); }
            if (want_dot) {
                const float xn = (av * ar - amean[nb]) * arstd[nb];
                f1[nb] += v; f2[nb] = fmaf(v, xn, f2[nb]);
            }
        }
    };
    auto flush_sums = [&]() {
        if (EPI != 0) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) { s1[nb] += (double)f1[nb]; s2[nb] += (double)f2[nb]; f1[nb] = 0.f; f2[nb] = 0.f; }
        }
    };

#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s) issue(rb, s, abuf[s], ars[s]);
    readb(0, 0, bf[0]);
    f32x16 acc[4], prev[4];
    // one row block `cur` into `acc`; with HAVEP the 64 C values of row block `pb` held in `prev` leave VPG per 16-MFMA group.
    // No branch inside: a basic-block boundary would end the lacing of LDS reads and stores into the MFMA stream.
    auto block = [&](auto HAVEP, int cur, int pb, int nxt) {
        constexpr bool havep = decltype(HAVEP)::value;
        asm volatile("" : "+v"(wofs));
        const int pbc = havep ? pb : cur;
        float* cp = Cb + (size_t)(pbc * 32 + 4 * kq) * ldc;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            if (want_dot) {
                constexpr int LEAD = NAUX - 1;
                const int sa = s + LEAD;
                if (sa < KS) issue_aux(pbc, sa, auxv[sa % NAUX], auxr[sa % NAUX]); else issue_aux(cur, sa - KS, auxv[sa % NAUX], auxr[sa % NAUX]);
            }
            const int sp = s + NBUF - 1;
            if (sp < KS) issue(cur, sp, abuf[sp % NBUF], ars[sp % NBUF]); else issue(nxt, sp - KS, abuf[sp % NBUF], ars[sp % NBUF]);
            float av[16];
            {
                const float4 (&src)[4] = abuf[s % NBUF];
#pragma unroll
                for (int q = 0; q < 4; ++q) { av[4 * q] = src[q].x; av[4 * q + 1] = src[q].y; av[4 * q + 2] = src[q].z; av[4 * q + 3] = src[q].w; }
            }
            if (XA > 0) {
                const float rs = XA == 2 ? ars[s % NBUF] : 1.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 c = *reinterpret_cast<const float4*>(tsc + s * 32 + 16 * kq + 4 * q);
                    const float4 h = *reinterpret_cast<const float4*>(tsh + s * 32 + 16 * kq + 4 * q);
                    av[4 * q] = fmaf(XA == 2 ? rs * av[4 * q] : av[4 * q], c.x, h.x);
                    av[4 * q + 1] = fmaf(XA == 2 ? rs * av[4 * q + 1] : av[4 * q + 1], c.y, h.y);
                    av[4 * q + 2] = fmaf(XA == 2 ? rs * av[4 * q + 2] : av[4 * q + 2], c.z, h.z);
                    av[4 * q + 3] = fmaf(XA == 2 ? rs * av[4 * q + 3] : av[4 * q + 3], c.w, h.w);
                }
            }
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const int sn = j4 == 3 ? (s + 1) % KS : s, jn = (j4 + 1) & 3;
                __builtin_amdgcn_sched_barrier(0);
                readb(sn, jn, bf[(j4 + 1) & 1]);
                const float4 (&bb)[4] = bf[j4 & 1];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float x = av[4 * j4 + jj];
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        const float y = jj == 0 ? bb[nb].x : jj == 1 ? bb[nb].y : jj == 2 ? bb[nb].z : bb[nb].w;
                        if (s == 0 && j4 == 0 && jj == 0) {
                            f32x16 z;
#pragma unroll
                            for (int r = 0; r < 16; ++r) z[r] = 0.f;
                            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, z, 0, 0, 0);
                        } else acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[nb], 0, 0, 0);
                    }
                }
                if (havep) {
#pragma unroll
                    for (int u = 0; u < VPG; ++u) {
                        const int e = VPS * s + VPG * j4 + u, nb = e >> 4, r = e & 15, ro = (r & 3) + 8 * (r >> 2);
                        emit(nb, prev[nb][r], cp, (size_t)ro * ldc + nb * 32, auxv[s % NAUX][VPG * j4 + u], auxr[s % NAUX][VPG * j4 + u], true);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            }
        }
    };
    // the wave's last block: plain epilogue, rows past M (a partial last block of the matrix) neither stored nor counted
    auto drain = [&](const f32x16 (&acc)[4], int blk) {
        const int rbase = blk * 32 + 4 * kq;
        float* cp = Cb + (size_t)rbase * ldc;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            float axv[16], axr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t row = (size_t)min(rbase + (r & 3) + 8 * (r >> 2), M - 1);
                axv[r] = want_dot ? auxb[row * auxld + nb * 32] : 0.f;
                axr[r] = has_rs ? aux_rs[row * aux_rs_stride] : 1.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                emit(nb, acc[nb][r], cp, (size_t)ro * ldc + nb * 32, axv[r], axr[r], rbase + ro < M);
            }
        }
    };
    if (rb < nblk) {
        block(std::false_type(), rb, -1, rb + stride);
        int pb = rb;
        rb += stride;
        while (rb < nblk) {
            // 64 register moves per 512 MFMAs (and one drain of the matrix pipe, ~200 cycles of 33k) instead of a second
            // copy of the loop body with the accumulator sets swapped
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) prev[nb] = acc[nb];
            block(std::true_type(), rb, pb, rb + stride);
            flush_sums();
            pb = rb; rb += stride;
        }
        drain(acc, pb);
        flush_sums();
    }

    if (EPI != 0) {
        __syncthreads();                      // every wave is done with the weight tile: reuse it
        double (*red)[4][2][32] = reinterpret_cast<double (*)[4][2][32]>(Ws);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            s1[nb] += __shfl_xor(s1[nb], 32, 64); s2[nb] += __shfl_xor(s2[nb], 32, 64);
            if (kq == 0) { red[wave][nb][0][m] = s1[nb]; red[wave][nb][1][m] = s2[nb]; }
        }
        __syncthreads();
        {
            const int nb = tid >> 6, w = (tid >> 5) & 1, c = tid & 31, col = n0 + nb * 32 + c;
            double t = 0.0;
#pragma unroll
            for (int v = 0; v < NW; ++v) t += red[v][nb][w][c];
            if (pr.parts) {
                // PARTS rows per problem, [row][2][N]: this workgroup's row and zeros in the rows no member owns
                pr.parts[((size_t)member * 2 + w) * a.N + col] = t;
                for (int z = member + members; z < PARTS; z += members) pr.parts[((size_t)z * 2 + w) * a.N + col] = 0.0;
            } else {
                double* dst = want_st ? (w ? pr.st_sq : pr.st_sum) : (w ? pr.dot_prod : pr.dot_sum);
                atomicAdd(dst + col, t);
            }
        }
    }
}

// dW[256][256] = op(X)^T dZ over the node range of one split; 8 waves = 4 (m) x 2 (n), wave tile 64 x 128.
// XA: BatchNorm (1) / row scale + BatchNorm (2) on X in storage coordinates (row = node, column = feature m)
constexpr int TNT = 512, RING = 8;
template <int XA>
__global__ void __launch_bounds__(TNT) k_tn(const GemmArgs a) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, ks = lane >> 5;
    const int mw0 = (wave >> 1) * 64, nw0 = (wave & 1) * 128;
    const int batch = blockIdx.x / a.nsplit, split = blockIdx.x % a.nsplit;
    const GemmProb& pr = a.p[batch];
    const int kb = split * a.kchunk, ke = min(a.K, kb + a.kchunk);
    const int np = (ke - kb + 1) / 2;                      // k pairs
    // Every workgroup walks its node range from a different starting pair (and wraps): the ranges are kchunk * lda * 4 bytes
    // apart -- 640 KB at config 5 -- so workgroups that start at their first row and advance in lockstep all ask the same few
    // HBM channels for their k-th row at the same time (measured: 270-280 us against 220 us with ranges 626 rows apart).
    const int rot = np > 0 ? (int)((blockIdx.x * 37u) % (unsigned)np) : 0;
    float sc[2] = {1.f, 1.f}, sh[2] = {0.f, 0.f};
    if (XA > 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bn_scale_shift(pr.xa.bn, mw0 + 2 * i + c, sc[c], sh[c]);
            if (pr.xa.bn.update && split == 0 && (wave & 1) == 0 && ks == 0) bn_update_running(pr.xa.bn, mw0 + 2 * i + c);
        }
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][d][r] = 0.f;
    float2 ar[RING]; float4 br[RING]; float rr[RING];
    auto pair_of = [&](int kp) { int q = kp + rot; q -= q >= np ? np : 0; return q; };     // kp < np + RING <= 2 np is not needed: clamp below
    auto issue = [&](int kp, float2& av, float4& bv, float& rs) {
        const size_t k = (size_t)min(kb + 2 * pair_of(max(min(kp, np - 1), 0)) + ks, a.K - 1);
        av = *reinterpret_cast<const float2*>(pr.A + k * a.lda + mw0 + 2 * i);
        bv = *reinterpret_cast<const float4*>(pr.B + k * a.ldb + nw0 + 4 * i);
        if (XA == 2) rs = pr.xa.rs[k * pr.xa.rs_stride];
    };
#pragma unroll
    for (int u = 0; u < RING; ++u) issue(u, ar[u], br[u], rr[u]);
    for (int kp = 0; kp < np; kp += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            float2 av = ar[u]; const float4 bv = br[u];
            if (XA == 2) { av.x *= rr[u]; av.y *= rr[u]; }
            if (XA > 0) { av.x = fmaf(av.x, sc[0], sh[0]); av.y = fmaf(av.y, sc[1], sh[1]); }
            if (!(kp + u < np && kb + 2 * pair_of(max(min(kp + u, np - 1), 0)) + ks < ke)) { av.x = 0.f; av.y = 0.f; }   // rows past the range: nothing
            issue(kp + u + RING, ar[u], br[u], rr[u]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.y, acc[0][1], 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.z, acc[0][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.w, acc[0][3], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[1][1], 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.z, acc[1][2], 0, 0, 0);
            acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.w, acc[1][3], 0, 0, 0);
        }
    }
    // accumulator (c, d), entry r of lane (i, ks): row m = mw0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * ks) + c, column n = nw0 + 4 * i + d
    float* C = pr.C + (size_t)split * a.M * a.ldc;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = mw0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * ks) + c;
            *reinterpret_cast<float4*>(C + (size_t)mm * a.ldc + nw0 + 4 * i) = make_float4(acc[c][0][r], acc[c][1][r], acc[c][2][r], acc[c][3][r]);
        }
}

}  // namespace wres

// ---- selection -----------------------------------------------------------------------------------------------
// node-level products on the weight-resident kernel: K = 128 / 256 whole, N one or two 128-column halves, enough rows
// CAL_AMD_WRES=0: these kernels off (the 128 x 128 tile kernels of gemm_big.hip take the launches: A/B measurements, bisecting)
static bool wres_on() { static const bool on = [] { const char* v = getenv("CAL_AMD_WRES"); return !(v && v[0] == '0'); }(); return on; }
bool gemm_wres_rows(int M, int N, int K) { return wres_on() && M >= 16384 && (K == 128 || K == 256) && (N == 128 || N == 256); }
// weight gradients on the direct-operand kernel: exactly 256 x 256 over a long node axis
bool gemm_wres_grad(int M, int N, int K) { return wres_on() && K >= 16384 && M == 256 && N == 256; }
// node ranges of such a gradient: one 256 x 256 slab per workgroup, ~one workgroup per CU over the whole batch
int gemm_wres_grad_splits(int K, int nbatch) {
    const int s = wres::GRID / (nbatch < 1 ? 1 : nbatch);
    return s < 1 ? 1 : (K / 64 < s ? (K / 64 < 1 ? 1 : K / 64) : s);
}
int gemm_wres_parts() { return wres::PARTS; }

static bool wres_aligned(const GemmArgs& a, int nbatch) {
    bool ok = a.lda % 4 == 0 && a.ldb % 4 == 0;
    for (int b = 0; b < nbatch; ++b)
        ok = ok && aligned16(a.p[b].A) && aligned16(a.p[b].B) && !a.p[b].xb.has_bn && !a.p[b].xb.rs && !(a.p[b].xa.rs && !a.p[b].xa.has_bn);
    return ok;
}
static int wres_xa_class(const GemmArgs& a, int nbatch) {
    int x = -1;
    for (int b = 0; b < nbatch; ++b) {
        const int m = a.p[b].xa.has_bn ? (a.p[b].xa.rs ? 2 : 1) : 0;
        if (x >= 0 && x != m) return -1;
        x = m;
    }
    return x;
}
// epilogue class of a batch: 0 none, 1 statistics, 2 dot statistics, 3 dot statistics with a row scale, -1 mixed
static int wres_epi_class(const GemmArgs& a, int nbatch) {
    int x = -1;
    for (int b = 0; b < nbatch; ++b) {
        const int m = a.p[b].dot_sum ? (a.p[b].aux_rs ? 3 : 2) : a.p[b].st_sum ? 1 : 0;
        if (x >= 0 && x != m) return -1;
        x = m;
    }
    return x;
}

template <bool B_KC, int XA, int KS>
static void wres_launch_epi(int epi, const GemmArgs& a, int nbatch, hipStream_t stream) {
    using namespace wres;
    if (epi == 0) hipLaunchKernelGGL((k_wres<B_KC, XA, KS, 0>), dim3(GRID), dim3(NT), 0, stream, a, nbatch);
    else if (epi == 1) hipLaunchKernelGGL((k_wres<B_KC, XA, KS, 1>), dim3(GRID), dim3(NT), 0, stream, a, nbatch);
    else if (epi == 2) hipLaunchKernelGGL((k_wres<B_KC, XA, KS, 2>), dim3(GRID), dim3(NT), 0, stream, a, nbatch);
    else hipLaunchKernelGGL((k_wres<B_KC, XA, KS, 3>), dim3(GRID), dim3(NT), 0, stream, a, nbatch);
}
template <bool B_KC, int KS>
static void wres_launch_xa(int xa, int epi, const GemmArgs& a, int nbatch, hipStream_t stream) {
    if (xa == 0) wres_launch_epi<B_KC, 0, KS>(epi, a, nbatch, stream);
    else if (xa == 1) wres_launch_epi<B_KC, 1, KS>(epi, a, nbatch, stream);
    else wres_launch_epi<B_KC, 2, KS>(epi, a, nbatch, stream);
}

// 1 = launched, 0 = not applicable (the caller goes on to gemm_big.hip / gemm.hip), < 0 = error
int launch_gemm_wres(bool transA, bool transB, const GemmArgs& a, int nbatch, hipStream_t stream) {
    if (!transA) {
        if (!(gemm_wres_rows(a.M, a.N, a.K) && a.nsplit == 1)) return 0;
        const int xa = wres_xa_class(a, nbatch), epi = wres_epi_class(a, nbatch);
        const int ncol = nbatch * (a.N / 128);
        const bool fits = ncol == 1 || ncol == 2 || ncol == 4;
        bool hasC = true;
        for (int b = 0; b < nbatch; ++b) hasC = hasC && a.p[b].C != nullptr;
        if (!wres_aligned(a, nbatch) || xa < 0 || epi < 0 || !fits || !hasC) {
            for (int b = 0; b < nbatch; ++b)
                if (a.p[b].parts && a.p[b].C) { set_error("launch_gemm_wres: a statistics GEMM sized for the weight-resident kernel cannot take this launch"); return -2; }      // (C == nullptr: sized by gemm_row_tiles(.., hasC = false) for the tile kernels)
            return 0;
        }
        if (a.K == 256) { if (transB) wres_launch_xa<true, 8>(xa, epi, a, nbatch, stream); else wres_launch_xa<false, 8>(xa, epi, a, nbatch, stream); }
        else { if (transB) wres_launch_xa<true, 4>(xa, epi, a, nbatch, stream); else wres_launch_xa<false, 4>(xa, epi, a, nbatch, stream); }
        if (hipGetLastError() != hipSuccess) { set_error("k_wres: launch failed"); return -2; }
        return 1;
    }
    if (transB) return 0;
    if (!(gemm_wres_grad(a.M, a.N, a.K) && a.kchunk % 2 == 0)) return 0;
    const int xa = wres_xa_class(a, nbatch);
    bool ok = wres_aligned(a, nbatch) && xa >= 0 && a.ldc % 4 == 0;
    for (int b = 0; b < nbatch; ++b) ok = ok && a.p[b].C && aligned16(a.p[b].C) && !a.p[b].bias && !a.p[b].st_sum && !a.p[b].dot_sum && !a.relu;
    if (!ok) return 0;
    const dim3 grid(nbatch * a.nsplit);
    if (xa == 0) hipLaunchKernelGGL((wres::k_tn<0>), grid, dim3(wres::TNT), 0, stream, a);
    else if (xa == 1) hipLaunchKernelGGL((wres::k_tn<1>), grid, dim3(wres::TNT), 0, stream, a);
    else hipLaunchKernelGGL((wres::k_tn<2>), grid, dim3(wres::TNT), 0, stream, a);
    if (hipGetLastError() != hipSuccess) { set_error("k_tn: launch failed"); return -2; }
    return 1;
}

}  // namespace cal
