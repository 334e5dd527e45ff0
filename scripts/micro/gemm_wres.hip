// Microbenchmark for the config-5 GEMM shapes ([160k,256] x [256,256] and its two gradients), fp32 MFMA:
//   k_wres : C = A B with the WEIGHT operand resident in LDS for the life of a persistent workgroup (k-contiguous rows,
//            ds_read_b128 fragments) and the NODE operand read straight from global memory into MFMA operand registers
//            (lane (row, k-half) takes 64 contiguous bytes of its row per 32-wide k step) -- no operand staging, no barrier
//            inside the loop, every wave runs its own 32-row blocks.
//   k_tn   : dW = X^T dZ (k = node rows) with BOTH operands straight from global memory: a dwordx2 / dwordx4 row load
//            hands lane i the columns 2i..2i+1 / 4i..4i+3, i.e. the A / B registers of 2 / 4 MFMAs whose output rows /
//            columns are interleaved; no LDS at all, split over node ranges into slabs.
// Baseline: the library's cal_gemm (gemm_big.hip) on the same buffers.
// hipcc --offload-arch=gfx950 -O3 -w gemm_wres.hip -o gemm_wres -ldl ; run from the repository root
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <functional>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct WArgs {
    const float* A; const float* B; float* C; const float* sc; const float* sh; double* parts; long long* dbg;
    int M, N, K, lda, ldb, ldc, nblk;
};

template <bool B_KC, int XA, int KS, int NBUF, bool LOADA = true, bool STOREC = true, bool READB = true, int NW = 8>
__global__ void __launch_bounds__(NW * 64) k_wres(const WArgs a) {
    constexpr int K = KS * 32, LDW = K + 4;
    static_assert(KS % NBUF == 0, "ring slots must line up across row blocks");
    __shared__ __attribute__((aligned(16))) float Ws[128 * LDW];
    __shared__ __attribute__((aligned(16))) float tsc[K], tsh[K];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m = lane & 31, kq = lane >> 5;
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int half = idx & 1, pair = (idx >> 1) * 8 + xcd, npairs = gridDim.x >> 1;
    const int n0 = half * 128;
    if (B_KC) {
        for (int i = tid; i < 128 * K / 4; i += NW * 64) {
            const int n = i / (K / 4), k4 = i % (K / 4);
            *reinterpret_cast<float4*>(Ws + n * LDW + 4 * k4) = *reinterpret_cast<const float4*>(a.B + (size_t)(n0 + n) * a.ldb + 4 * k4);
        }
    } else {
        for (int i = tid; i < 32 * K; i += NW * 64) {
            const int k = i % K, n4 = i / K;
            const float4 v = *reinterpret_cast<const float4*>(a.B + (size_t)k * a.ldb + n0 + 4 * n4);
            Ws[(4 * n4 + 0) * LDW + k] = v.x; Ws[(4 * n4 + 1) * LDW + k] = v.y;
            Ws[(4 * n4 + 2) * LDW + k] = v.z; Ws[(4 * n4 + 3) * LDW + k] = v.w;
        }
    }
    if (XA) for (int i = tid; i < K; i += NW * 64) { tsc[i] = a.sc[i]; tsh[i] = a.sh[i]; }
    __syncthreads();

    const int stride = npairs * NW;
    int rb = pair * NW + wave;
    int wofs = (m * LDW + 16 * kq) / 4;                 // laundered once per row block: the fragments are loop-invariant, and
                                                    // hipcc would hoist all 128 ds_read_b128 out of the loop and spill them
    float4 abuf[NBUF][4];
    auto issue = [&](int blk, int s, float4 (&dst)[4]) {
        const float* p = a.A + (size_t)min(blk * 32 + m, a.M - 1) * a.lda + s * 32 + 16 * kq;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const float4*>(p + 4 * q);
    };
    float4 bf[2][4];
    auto readb = [&](int s, int j4, float4 (&dst)[4]) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) dst[nb] = *reinterpret_cast<const float4*>(Ws + 4 * wofs + nb * 32 * LDW + s * 32 + 4 * j4);
    };
#pragma unroll
    for (int s = 0; s < (LOADA ? NBUF - 1 : NBUF); ++s) issue(rb, s, abuf[s]);
    readb(0, 0, bf[0]); readb(0, 1, bf[1]);
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    while (rb < a.nblk) {
        asm volatile("" : "+v"(wofs));
        const int rbn = rb + stride;
        f32x16 acc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            const int sp = s + NBUF - 1;
            if (LOADA) { if (sp < KS) issue(rb, sp, abuf[sp % NBUF]); else issue(rbn, sp - KS, abuf[sp % NBUF]); }
            float av[16];
            {
                const float4 (&src)[4] = abuf[s % NBUF];
#pragma unroll
                for (int q = 0; q < 4; ++q) { av[4 * q] = src[q].x; av[4 * q + 1] = src[q].y; av[4 * q + 2] = src[q].z; av[4 * q + 3] = src[q].w; }
            }
            if (XA) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 c = *reinterpret_cast<const float4*>(tsc + s * 32 + 16 * kq + 4 * q);
                    const float4 h = *reinterpret_cast<const float4*>(tsh + s * 32 + 16 * kq + 4 * q);
                    av[4 * q] = fmaf(av[4 * q], c.x, h.x); av[4 * q + 1] = fmaf(av[4 * q + 1], c.y, h.y);
                    av[4 * q + 2] = fmaf(av[4 * q + 2], c.z, h.z); av[4 * q + 3] = fmaf(av[4 * q + 3], c.w, h.w);
                }
            }
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const int sn = j4 == 3 ? (s + 1) % KS : s, jn = (j4 + 1) & 3;
                __builtin_amdgcn_sched_barrier(0);
                if (READB) readb(sn, jn, bf[(j4 + 1) & 1]);
                const float4 (&bb)[4] = bf[j4 & 1];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float x = av[4 * j4 + jj];
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        const float y = jj == 0 ? bb[nb].x : jj == 1 ? bb[nb].y : jj == 2 ? bb[nb].z : bb[nb].w;
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[nb], 0, 0, 0);
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
        // epilogue: row = rb*32 + (r&3) + 8*(r>>2) + 4*kq, col = n0 + nb*32 + m
        const int rbase = rb * 32 + 4 * kq;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            float* cp = a.C + (size_t)rbase * a.ldc + n0 + nb * 32 + m;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                const float v = acc[nb][r];
                if (rbase + ro < a.M) {
                    if (STOREC || v == 123.456f) cp[(size_t)ro * a.ldc] = v;
                    if (a.parts) { s1[nb] += (double)v; s2[nb] += (double)v * (double)v; }
                }
            }
        }
        rb = rbn;
    }
    if (a.parts) {
        __syncthreads();
        double (*red)[4][2][32] = reinterpret_cast<double (*)[4][2][32]>(Ws);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            s1[nb] += __shfl_xor(s1[nb], 32, 64); s2[nb] += __shfl_xor(s2[nb], 32, 64);
            if (kq == 0) { red[wave][nb][0][m] = s1[nb]; red[wave][nb][1][m] = s2[nb]; }
        }
        __syncthreads();
        if (tid < 256) {
            const int nb = tid >> 6, w = (tid >> 5) & 1, c = tid & 31;
            double t = 0;
            for (int v = 0; v < NW; ++v) t += red[v][nb][w][c];
            a.parts[((size_t)(pair) * 2 + w) * a.N + n0 + nb * 32 + c] = t;
        }
    }
}


// ---- version 2: the epilogue of row block i (C stores + column statistics) rides in the MFMA stream of block i+1 (two accumulator
// sets), so a wave's matrix stream never pauses; only whole 32-row blocks (the caller / a tail path takes M % 32 rows).
template <bool B_KC, int XA, int KS, int NBUF, bool STATS, int NW = 8, bool LOADA = true, bool STOREC = true, bool READB = true>
__global__ void __launch_bounds__(NW * 64) k_wres2(const WArgs a) {
    constexpr int K = KS * 32, LDW = K + 4;
    const long long wentry = wall_clock64();
    static_assert(KS % NBUF == 0, "ring slots must line up across row blocks");
    __shared__ __attribute__((aligned(16))) float Ws[128 * LDW];
    __shared__ __attribute__((aligned(16))) float tsc[K], tsh[K];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m = lane & 31, kq = lane >> 5;
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int half = idx & 1, pair = (idx >> 1) * 8 + xcd, npairs = gridDim.x >> 1;
    const int n0 = half * 128;
    if (B_KC) {
        for (int i = tid; i < 128 * K / 4; i += NW * 64) {
            const int n = i / (K / 4), k4 = i % (K / 4);
            *reinterpret_cast<float4*>(Ws + n * LDW + 4 * k4) = *reinterpret_cast<const float4*>(a.B + (size_t)(n0 + n) * a.ldb + 4 * k4);
        }
    } else {
        for (int i = tid; i < 32 * K; i += NW * 64) {
            const int k = i % K, n4 = i / K;
            const float4 v = *reinterpret_cast<const float4*>(a.B + (size_t)k * a.ldb + n0 + 4 * n4);
            Ws[(4 * n4 + 0) * LDW + k] = v.x; Ws[(4 * n4 + 1) * LDW + k] = v.y;
            Ws[(4 * n4 + 2) * LDW + k] = v.z; Ws[(4 * n4 + 3) * LDW + k] = v.w;
        }
    }
    if (XA) for (int i = tid; i < K; i += NW * 64) { tsc[i] = a.sc[i]; tsh[i] = a.sh[i]; }
    __syncthreads();

    const long long c0 = clock64(), w0 = wall_clock64();
    const int stride = npairs * NW;
    const int nfull = a.M / 32;                      // whole blocks only
    int rb = pair * NW + wave;
    int wofs = (m * LDW + 16 * kq) / 4;
    float4 abuf[NBUF][4];
    auto issue = [&](int blk, int s, float4 (&dst)[4]) {
        const float* p = a.A + (size_t)min(blk * 32 + m, a.M - 1) * a.lda + s * 32 + 16 * kq;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const float4*>(p + 4 * q);
    };
    float4 bf[2][4];
    auto readb = [&](int s, int j4, float4 (&dst)[4]) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) dst[nb] = *reinterpret_cast<const float4*>(Ws + 4 * wofs + nb * 32 * LDW + s * 32 + 4 * j4);
    };
#pragma unroll
    for (int s = 0; s < (LOADA ? NBUF - 1 : NBUF); ++s) issue(rb, s, abuf[s]);
    readb(0, 0, bf[0]); readb(0, 1, bf[1]);
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    f32x16 accA[4], accB[4];
    // one row block into `acc`; the 64 values of `prev` (row block pb, -1: none) go out two per 16-MFMA group
    auto block = [&](f32x16 (&acc)[4], const f32x16 (&prev)[4], int cur, int pb, int nxt) {
        asm volatile("" : "+v"(wofs));
        float* cp = a.C + (size_t)(pb * 32 + 4 * kq) * a.ldc + n0 + m;
        const bool havep = pb >= 0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            const int sp = s + NBUF - 1;
            if (LOADA) { if (sp < KS) issue(cur, sp, abuf[sp % NBUF]); else issue(nxt, sp - KS, abuf[sp % NBUF]); }
            float av[16];
            {
                const float4 (&src)[4] = abuf[s % NBUF];
#pragma unroll
                for (int q = 0; q < 4; ++q) { av[4 * q] = src[q].x; av[4 * q + 1] = src[q].y; av[4 * q + 2] = src[q].z; av[4 * q + 3] = src[q].w; }
            }
            if (XA) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 c = *reinterpret_cast<const float4*>(tsc + s * 32 + 16 * kq + 4 * q);
                    const float4 h = *reinterpret_cast<const float4*>(tsh + s * 32 + 16 * kq + 4 * q);
                    av[4 * q] = fmaf(av[4 * q], c.x, h.x); av[4 * q + 1] = fmaf(av[4 * q + 1], c.y, h.y);
                    av[4 * q + 2] = fmaf(av[4 * q + 2], c.z, h.z); av[4 * q + 3] = fmaf(av[4 * q + 3], c.w, h.w);
                }
            }
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const int sn = j4 == 3 ? (s + 1) % KS : s, jn = (j4 + 1) & 3;
                __builtin_amdgcn_sched_barrier(0);
                if (READB) readb(sn, jn, bf[(j4 + 1) & 1]);
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
                // two values of the previous block: e = 2 * (4 s + j4) + {0, 1} -> (nb, r)
                if (havep) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int e = 2 * (4 * s + j4) + u;
                        if (e < 64) {
                            const int nb = e >> 4, r = e & 15, ro = (r & 3) + 8 * (r >> 2);
                            const float v = prev[nb][r];
                            if (STOREC) cp[(size_t)ro * a.ldc + nb * 32] = v; else asm volatile("" :: "v"(v));
                            if (STATS) { s1[nb] += (double)v; s2[nb] += (double)v * (double)v; }
                        }
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
    int pb = -1;
    while (rb < nfull) {
        block(accA, accB, rb, pb, rb + stride);
        pb = rb; rb += stride;
        if (rb >= nfull) {      // drain A
            float* cp = a.C + (size_t)(pb * 32 + 4 * kq) * a.ldc + n0 + m;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = accA[nb][r];
                    cp[(size_t)((r & 3) + 8 * (r >> 2)) * a.ldc + nb * 32] = v;
                    if (STATS) { s1[nb] += (double)v; s2[nb] += (double)v * (double)v; }
                }
            pb = -1;
            break;
        }
        block(accB, accA, rb, pb, rb + stride);
        pb = rb; rb += stride;
        if (rb >= nfull) {      // drain B
            float* cp = a.C + (size_t)(pb * 32 + 4 * kq) * a.ldc + n0 + m;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = accB[nb][r];
                    cp[(size_t)((r & 3) + 8 * (r >> 2)) * a.ldc + nb * 32] = v;
                    if (STATS) { s1[nb] += (double)v; s2[nb] += (double)v * (double)v; }
                }
            pb = -1;
            break;
        }
    }
    if (a.dbg && tid == 0 && (b == 0 || b == 100)) { a.dbg[2 * (b != 0)] = clock64() - c0; a.dbg[2 * (b != 0) + 1] = wall_clock64() - w0; }
    if (a.dbg && lane == 0) { a.dbg[8 + 2 * (b * NW + wave)] = w0; if (tid == 0 && b == 0) a.dbg[4] = w0 - wentry; a.dbg[8 + 2 * (b * NW + wave) + 1] = wall_clock64(); }
    if (STATS) {
        __syncthreads();
        double (*red)[4][2][32] = reinterpret_cast<double (*)[4][2][32]>(Ws);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            s1[nb] += __shfl_xor(s1[nb], 32, 64); s2[nb] += __shfl_xor(s2[nb], 32, 64);
            if (kq == 0) { red[wave][nb][0][m] = s1[nb]; red[wave][nb][1][m] = s2[nb]; }
        }
        __syncthreads();
        if (tid < 256) {
            const int nb = tid >> 6, w = (tid >> 5) & 1, c = tid & 31;
            double t = 0;
            for (int v = 0; v < NW; ++v) t += red[v][nb][w][c];
            a.parts[((size_t)(pair) * 2 + w) * a.N + n0 + nb * 32 + c] = t;
        }
    }
}

struct TArgs { const float* X; const float* D; float* slab; int K, ldx, ldd, kchunk; };
// dW[256][256] per workgroup over its node range; 8 waves = 4 (m) x 2 (n), wave tile 64 x 128
template <int R, bool SYNC = false, bool STORE = true>
__global__ void __launch_bounds__(512) k_tn(const TArgs a) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, ks = lane >> 5;
    const int mw0 = (wave >> 1) * 64, nw0 = (wave & 1) * 128;
    const int kb = blockIdx.x * a.kchunk, ke = min(a.K, kb + a.kchunk);
    const int np = (ke - kb + 1) / 2;                      // k pairs
    f32x16 acc[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][d][r] = 0.f;
    float2 ar[R]; float4 br[R];
    auto issue = [&](int kp, float2& av, float4& bv) {
        const int k = min(kb + 2 * kp + ks, a.K - 1);
        av = *reinterpret_cast<const float2*>(a.X + (size_t)k * a.ldx + mw0 + 2 * i);
        bv = *reinterpret_cast<const float4*>(a.D + (size_t)k * a.ldd + nw0 + 4 * i);
    };
#pragma unroll
    for (int u = 0; u < R; ++u) issue(u, ar[u], br[u]);
    for (int kp = 0; kp < np; kp += R) {
        if (SYNC) __syncthreads();
#pragma unroll
        for (int u = 0; u < R; ++u) {
            float2 av = ar[u]; float4 bv = br[u];
            const bool ok = kb + 2 * (kp + u) + ks < ke;
            if (!ok) { av.x = 0.f; av.y = 0.f; }
            issue(kp + u + R, ar[u], br[u]);
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
    float* sl = a.slab + (size_t)blockIdx.x * 256 * 256;
    if (!STORE && acc[0][0][0] != 123.f) return;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ri = (r & 3) + 8 * (r >> 2) + 4 * ks;
            const int mm = mw0 + 2 * ri + c;
            *reinterpret_cast<float4*>(sl + (size_t)mm * 256 + nw0 + 4 * i) = make_float4(acc[c][0][r], acc[c][1][r], acc[c][2][r], acc[c][3][r]);
        }
}


// 4 waves (one per SIMD) = 2 (m) x 2 (n), wave tile 128 x 128: a dwordx4 of each operand row feeds 16 MFMAs
template <int R, bool SYNC>
__global__ void __launch_bounds__(256) k_tn4(const TArgs a) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, ks = lane >> 5;
    const int mw0 = (wave >> 1) * 128, nw0 = (wave & 1) * 128;
    const int kb = blockIdx.x * a.kchunk, ke = min(a.K, kb + a.kchunk);
    const int np = (ke - kb + 1) / 2;
    f32x16 acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][d][r] = 0.f;
    float4 ar[R], br[R];
    auto issue = [&](int kp, float4& av, float4& bv) {
        const int k = min(kb + 2 * kp + ks, a.K - 1);
        av = *reinterpret_cast<const float4*>(a.X + (size_t)k * a.ldx + mw0 + 4 * i);
        bv = *reinterpret_cast<const float4*>(a.D + (size_t)k * a.ldd + nw0 + 4 * i);
    };
#pragma unroll
    for (int u = 0; u < R; ++u) issue(u, ar[u], br[u]);
    for (int kp = 0; kp < np; kp += R) {
        if (SYNC) __syncthreads();
#pragma unroll
        for (int u = 0; u < R; ++u) {
            float4 av = ar[u]; const float4 bv = br[u];
            if (!(kb + 2 * (kp + u) + ks < ke)) av = make_float4(0.f, 0.f, 0.f, 0.f);
            issue(kp + u + R, ar[u], br[u]);
            const float aa[4] = {av.x, av.y, av.z, av.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[c][d] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[c], bb[d], acc[c][d], 0, 0, 0);
        }
    }
    float* sl = a.slab + (size_t)blockIdx.x * 256 * 256;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = mw0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * ks) + c;
            *reinterpret_cast<float4*>(sl + (size_t)mm * 256 + nw0 + 4 * i) = make_float4(acc[c][0][r], acc[c][1][r], acc[c][2][r], acc[c][3][r]);
        }
}

// dW = X^T dZ with both operand row tiles staged ONCE per CU through LDS in their natural (node-major) layout: a ds_read_b64 /
// ds_read_b128 of a node row hands lane i the columns 2i..2i+1 / 4i..4i+3 (the interleaved-row trick of k_tn, conflict-free
// without padding); 8 waves = 4 (m) x 2 (n), wave tile 64 x 128; tile = TK node rows, two LDS stages, global loads one tile ahead.
template <int TK>
__global__ void __launch_bounds__(512) k_tnl(const TArgs a) {
    constexpr int NQ = TK * 64 / 512;                // float4 per thread per operand tile
    __shared__ __attribute__((aligned(16))) float Xs[2][TK * 256], Ds[2][TK * 256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, ks = lane >> 5;
    const int mw0 = (wave >> 1) * 64, nw0 = (wave & 1) * 128;
    const int kb = blockIdx.x * a.kchunk, ke = min(a.K, kb + a.kchunk);
    const int nt = (ke - kb + TK - 1) / TK;
    f32x16 acc[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][d][r] = 0.f;
    float4 rx[NQ], rd[NQ];
    auto gload = [&](int t) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int f = tid + q * 512, row = f >> 6, c4 = (f & 63) * 4;
            const int k = kb + t * TK + row;
            const size_t kc = (size_t)min(k, a.K - 1);
            rx[q] = *reinterpret_cast<const float4*>(a.X + kc * a.ldx + c4);
            rd[q] = *reinterpret_cast<const float4*>(a.D + kc * a.ldd + c4);
        }
    };
    auto sstore = [&](int t, int st) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int f = tid + q * 512, row = f >> 6, c4 = (f & 63) * 4;
            const bool ok = kb + t * TK + row < ke;
            *reinterpret_cast<float4*>(&Xs[st][row * 256 + c4]) = ok ? rx[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&Ds[st][row * 256 + c4]) = rd[q];
        }
    };
    gload(0); sstore(0, 0);
    if (nt > 1) gload(1);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int st = t & 1;
        const float* xs = &Xs[st][ks * 256 + mw0 + 2 * i];
        const float* ds = &Ds[st][ks * 256 + nw0 + 4 * i];
        float2 av[2]; float4 bv[2];
        av[0] = *reinterpret_cast<const float2*>(xs); bv[0] = *reinterpret_cast<const float4*>(ds);
#pragma unroll
        for (int p = 0; p < TK / 2; ++p) {
            if (p + 1 < TK / 2) {
                av[(p + 1) & 1] = *reinterpret_cast<const float2*>(xs + (p + 1) * 512);
                bv[(p + 1) & 1] = *reinterpret_cast<const float4*>(ds + (p + 1) * 512);
            }
            const float2 x = av[p & 1]; const float4 y = bv[p & 1];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.y, acc[0][1], 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.z, acc[0][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.w, acc[0][3], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.y, acc[1][1], 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.z, acc[1][2], 0, 0, 0);
            acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.w, acc[1][3], 0, 0, 0);
        }
        if (t + 1 < nt) sstore(t + 1, st ^ 1);
        if (t + 2 < nt) gload(t + 2);
        __syncthreads();
    }
    float* sl = a.slab + (size_t)blockIdx.x * 256 * 256;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = mw0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * ks) + c;
            *reinterpret_cast<float4*>(sl + (size_t)mm * 256 + nw0 + 4 * i) = make_float4(acc[c][0][r], acc[c][1][r], acc[c][2][r], acc[c][3][r]);
        }
}

typedef int (*gemm_fn)(int, int, const float*, const float*, float*, const float*, int, float*, int64_t, int64_t, int64_t, void*);
typedef int64_t (*ws_fn)(int64_t, int64_t, int64_t);

template <class F>
static float time_us(F f, int it = 10) {
    f(); CK(hipDeviceSynchronize());
    hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    CK(hipEventRecord(s));
    for (int i = 0; i < it; ++i) f();
    CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
    float ms; CK(hipEventElapsedTime(&ms, s, e));
    return ms * 1000.f / it;
}

static float* g_fill_a = nullptr; static float* g_fill_b = nullptr; static size_t g_fill_n = 0; static int g_fill = 0;
template <class F>
static float time_us_filled(F f, int it = 10) {
    f(); CK(hipDeviceSynchronize());
    hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    float tot = 0;
    for (int i = 0; i < it; ++i) {
        for (int r = 0; r < g_fill; ++r) CK(hipMemcpyAsync(g_fill_b, g_fill_a, g_fill_n, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(s)); f(); CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
        float ms; CK(hipEventElapsedTime(&ms, s, e)); tot += ms;
    }
    return tot * 1000.f / it;
}
static float time_any(std::function<void()> f) { return g_fill ? time_us_filled(f, 10) : time_us(f, 10); }
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 160000, N = 256, K = argc > 2 ? atoi(argv[2]) : 256;
    std::vector<float> hA((size_t)M * K), hB((size_t)K * N), hBt((size_t)N * K), hsc(K), hsh(K);
    srand(7);
    for (auto& v : hA) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hB) v = (rand() % 2001 - 1000) / 8000.f;
    for (int k = 0; k < K; ++k) for (int n = 0; n < N; ++n) hBt[(size_t)n * K + k] = hB[(size_t)k * N + n];
    for (int k = 0; k < K; ++k) { hsc[k] = 0.5f + (k % 7) * 0.1f; hsh[k] = (k % 5) * 0.05f - 0.1f; }
    float *A, *B, *Bt, *C, *sc, *sh, *slab, *D; double* parts;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&Bt, hB.size() * 4));
    CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&D, (size_t)M * N * 4)); CK(hipMalloc(&sc, K * 4)); CK(hipMalloc(&sh, K * 4));
    CK(hipMalloc(&parts, 256 * 2 * N * 8)); CK(hipMalloc(&slab, (size_t)512 * 256 * 256 * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Bt, hBt.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sc, hsc.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sh, hsh.data(), K * 4, hipMemcpyHostToDevice));
    g_fill = argc > 3 ? atoi(argv[3]) : 0; g_fill_n = (size_t)192 << 20; if (g_fill) { CK(hipMalloc(&g_fill_a, g_fill_n)); CK(hipMalloc(&g_fill_b, g_fill_n)); }
    const double gf = 2.0 * M * N * K / 1e6;       // MFLOP -> TF = gf / us
    std::vector<float> hC((size_t)M * N);
    auto check = [&](const char* what, bool xa) {
        CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int t = 0; t < 97; ++t) {
            const int r = (int)(((long long)t * 7919 * 131) % M);
            for (int n = 0; n < N; n += 3) {
                double acc = 0;
                for (int k = 0; k < K; ++k) { const double x = hA[(size_t)r * K + k]; acc += (xa ? x * hsc[k] + hsh[k] : x) * hB[(size_t)k * N + n]; }
                worst = fmax(worst, fabs(acc - hC[(size_t)r * N + n]));
            }
        }
        // last rows too
        for (int r = M - 40; r < M; ++r) for (int n = 0; n < N; n += 5) {
            double acc = 0;
            for (int k = 0; k < K; ++k) { const double x = hA[(size_t)r * K + k]; acc += (xa ? x * hsc[k] + hsh[k] : x) * hB[(size_t)k * N + n]; }
            worst = fmax(worst, fabs(acc - hC[(size_t)r * N + n]));
        }
        printf("   %-28s max |err| vs fp64 on sampled rows: %.3g\n", what, worst);
    };
    // ---- library baseline --------------------------------------------------------------------------------------------
    void* h = dlopen("cal_amd/lib/libcalhip.so", RTLD_NOW);
    gemm_fn cal_gemm = h ? (gemm_fn)dlsym(h, "cal_gemm") : nullptr;
    ws_fn cal_gemm_ws = h ? (ws_fn)dlsym(h, "cal_gemm_ws") : nullptr;
    if (cal_gemm && K == 256) {
        float* ws = nullptr;
        int64_t wsn = cal_gemm_ws ? cal_gemm_ws(256, 256, M) : 0;
        CK(hipMalloc(&ws, (size_t)(wsn > 4 ? wsn : 4) * 4 + (64 << 20)));
        float t = time_any([&] { cal_gemm(0, 0, A, B, C, nullptr, 0, ws, M, N, K, nullptr); });
        printf("library NN  %8.1f us  %6.1f TF\n", t, gf / t); check("library NN", false);
        t = time_any([&] { cal_gemm(0, 1, A, Bt, C, nullptr, 0, ws, M, N, K, nullptr); });
        printf("library NT  %8.1f us  %6.1f TF\n", t, gf / t);
        t = time_any([&] { cal_gemm(1, 0, A, D, slab, nullptr, 0, ws, 256, 256, M, nullptr); });
        printf("library TN  %8.1f us  %6.1f TF  (dW = A^T D, incl. its slab reduction)\n", t, gf / t);
    } else printf("libcalhip.so not found: no baseline\n");
    // ---- k_wres ---------------------------------------------------------------------------------------------------------
    long long* dbg; CK(hipMalloc(&dbg, 64 + 256 * 12 * 16)); CK(hipMemset(dbg, 0, 64));
    WArgs wa{A, B, C, sc, sh, nullptr, dbg, M, N, K, K, N, N, (M + 31) / 32};
    int nthreads = 512;
    auto run = [&](auto kern, const char* name, WArgs w, bool xa) {
        CK(hipMemset(C, 0, (size_t)M * N * 4));
        float t = time_any([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(nthreads), 0, 0, w); });
        CK(hipGetLastError());
        long long hd[8]; CK(hipMemcpy(hd, dbg, 64, hipMemcpyDeviceToHost)); CK(hipMemset(dbg, 0, 64));
        printf("%-34s %8.1f us  %6.1f TF", name, t, gf / t);
        if (hd[1] > 0) printf("   [wave 0 of wg 0: %.1f us at %.0f MHz; wg 100: %.1f us at %.0f MHz]", hd[1] / 100.0, hd[0] * 100.0 / hd[1], hd[3] / 100.0, hd[3] ? hd[2] * 100.0 / hd[3] : 0.0);
        printf("\n");
        if (hd[1] > 0) {
            const int nw = nthreads / 64;
            std::vector<long long> hw(256 * nw * 2);
            CK(hipMemcpy(hw.data(), dbg + 8, hw.size() * 8, hipMemcpyDeviceToHost));
            long long t0 = hw[0], smax = 0, emin = 1LL << 62, emax = 0; double dsum = 0, dmax = 0, dmin = 1e30;
            for (int i = 0; i < 256 * nw; ++i) t0 = std::min(t0, hw[2 * i]);
            for (int i = 0; i < 256 * nw; ++i) {
                smax = std::max(smax, hw[2 * i] - t0); emin = std::min(emin, hw[2 * i + 1] - t0); emax = std::max(emax, hw[2 * i + 1] - t0);
                const double d = (hw[2 * i + 1] - hw[2 * i]) / 100.0; dsum += d; dmax = std::max(dmax, d); dmin = std::min(dmin, d);
            }
            printf("      staging of wg 0: %.1f us\n", hd[4] / 100.0);
            printf("      waves: last start +%.1f us, first end +%.1f, last end +%.1f; duration min %.1f mean %.1f max %.1f us\n",
                   smax / 100.0, emin / 100.0, emax / 100.0, dmin, dsum / (256 * nw), dmax);
            // per-XCD mean end
            for (int x = 0; x < 8; ++x) { double e = 0; int c = 0; for (int w = 0; w < 256; ++w) if (w % 8 == x) for (int v = 0; v < nw; ++v) { e += (hw[2 * (w * nw + v) + 1] - t0) / 100.0; ++c; } printf(" xcd%d %.0f", x, e / c); }
            printf("\n");
        }
        check(name, xa);
    };
    if (K == 64) {
        run(k_wres2<false, 0, 2, 2, false>, "K=64 k_wres2 NN plain", wa, false);
        run(k_wres2<false, 0, 2, 2, false, 8, false, false, true>, "K=64 k_wres2 NN MFMA + B reads", wa, false);
        run(k_wres2<false, 0, 2, 2, false, 8, false, false, false>, "K=64 k_wres2 NN MFMA only", wa, false);
        nthreads = 256;
        run(k_wres2<false, 0, 2, 2, false, 4, false, false, false>, "K=64 k_wres2 NN MFMA only 4 waves", wa, false);
        return 0;
    }
    run(k_wres<false, 0, 8, 2>, "k_wres NN plain nbuf2", wa, false);
    run(k_wres<false, 0, 8, 4>, "k_wres NN plain nbuf4", wa, false);
    run(k_wres<false, 1, 8, 2>, "k_wres NN BN nbuf2", wa, true);
    run(k_wres<false, 1, 8, 4>, "k_wres NN BN nbuf4", wa, true);
    WArgs wp = wa; wp.parts = parts;
    run(k_wres<false, 1, 8, 2>, "k_wres NN BN stats nbuf2", wp, true);
    run(k_wres<false, 1, 8, 4>, "k_wres NN BN stats nbuf4", wp, true);
    run(k_wres<false, 0, 8, 2, false, false>, "k_wres NN no A loads, no C stores", wa, false);
    run(k_wres<false, 0, 8, 2, true, false>, "k_wres NN A loads, no C stores", wa, false);
    run(k_wres<false, 0, 8, 2, false, true>, "k_wres NN no A loads, C stores", wa, false);
    run(k_wres<false, 0, 8, 2, false, false, false>, "k_wres NN MFMA only (no LDS reads)", wa, false);
    nthreads = 256;
    run(k_wres<false, 0, 8, 2, false, false, false, 4>, "k_wres NN MFMA only, 4 waves", wa, false);
    run(k_wres<false, 0, 8, 2, false, false, true, 4>, "k_wres NN MFMA + LDS, 4 waves", wa, false);
    run(k_wres<false, 0, 8, 2, true, true, true, 4>, "k_wres NN full, 4 waves", wa, false);
    nthreads = 768;
    run(k_wres<false, 0, 8, 2, false, false, false, 12>, "k_wres NN MFMA only, 12 waves", wa, false);
    run(k_wres<false, 0, 8, 2, false, false, true, 12>, "k_wres NN MFMA + LDS, 12 waves", wa, false);
    run(k_wres<false, 0, 8, 2, true, true, true, 12>, "k_wres NN full, 12 waves", wa, false);
    nthreads = 512;
    nthreads = 512;
    run(k_wres2<false, 0, 8, 2, false>, "k_wres2 NN plain", wa, false);
    run(k_wres2<false, 1, 8, 2, false>, "k_wres2 NN BN", wa, true);
    run(k_wres2<false, 1, 8, 2, true>, "k_wres2 NN BN stats", wp, true);
    run(k_wres2<false, 0, 8, 2, false, 8, false, true, true>, "k_wres2 NN no A loads", wa, false);
    run(k_wres2<false, 0, 8, 2, false, 8, true, false, true>, "k_wres2 NN no C stores", wa, false);
    run(k_wres2<false, 0, 8, 2, false, 8, true, true, false>, "k_wres2 NN no B reads", wa, false);
    run(k_wres2<false, 0, 8, 2, false, 8, false, false, true>, "k_wres2 NN MFMA + B reads", wa, false);
    run(k_wres2<false, 0, 8, 2, false, 8, false, false, false>, "k_wres2 NN MFMA only", wa, false);
    nthreads = 256;
    run(k_wres2<false, 0, 8, 2, false, 4, false, false, false>, "k_wres2 NN MFMA only 4 waves", wa, false);
    run(k_wres2<false, 0, 8, 2, false, 4>, "k_wres2 NN plain 4 waves", wa, false);
    run(k_wres2<false, 0, 8, 4, false, 4>, "k_wres2 NN plain 4 waves nbuf4", wa, false);
    run(k_wres2<false, 1, 8, 4, true, 4>, "k_wres2 NN BN stats 4 waves nbuf4", wp, true);
    nthreads = 512;
    WArgs wt = wa; wt.B = Bt; wt.ldb = K;
    run(k_wres<true, 0, 8, 2>, "k_wres NT plain nbuf2", wt, false);
    run(k_wres<true, 0, 8, 4>, "k_wres NT plain nbuf4", wt, false);
    // ---- k_tn -----------------------------------------------------------------------------------------------------------
    {
        CK(hipMemcpy(D, C, (size_t)M * N * 4, hipMemcpyDeviceToDevice));      // D = A B (any data)
        for (int nwg : {256, 512}) {
            int kchunk = ((M + nwg - 1) / nwg + 1) / 2 * 2;
            TArgs ta{A, D, slab, M, K, N, kchunk};
            const int g = (M + kchunk - 1) / kchunk;
            float tl16 = time_any([&] { hipLaunchKernelGGL((k_tnl<16>), dim3(g), dim3(512), 0, 0, ta); });
            float tl32 = time_any([&] { hipLaunchKernelGGL((k_tnl<32>), dim3(g), dim3(512), 0, 0, ta); });
            CK(hipGetLastError());
            printf("k_tnl %d wgs (LDS-staged natural layout): TK 16 %8.1f us %6.1f TF | TK 32 %8.1f us %6.1f TF\n", g, tl16, gf / tl16, tl32, gf / tl32);
            {
                std::vector<float> hs((size_t)g * 65536), hD((size_t)M * N);
                CK(hipMemcpy(hs.data(), slab, hs.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hD.data(), D, hD.size() * 4, hipMemcpyDeviceToHost));
                double worst = 0, scale = 0;
                for (int t = 0; t < 40; ++t) {
                    const int mm = (t * 37) % 256, nn = (t * 91 + 5) % 256;
                    double ref = 0; for (int k = 0; k < M; ++k) ref += (double)hA[(size_t)k * K + mm] * hD[(size_t)k * N + nn];
                    double got = 0; for (int z = 0; z < g; ++z) got += hs[(size_t)z * 65536 + mm * 256 + nn];
                    worst = fmax(worst, fabs(ref - got)); scale = fmax(scale, fabs(ref));
                }
                printf("   k_tnl<32> max |err| %.3g (scale %.3g)\n", worst, scale);
            }
            float ts = time_any([&] { hipLaunchKernelGGL((k_tn<8, true>), dim3(g), dim3(512), 0, 0, ta); });
            float tns = time_any([&] { hipLaunchKernelGGL((k_tn<8, true, false>), dim3(g), dim3(512), 0, 0, ta); });
            float t44 = time_any([&] { hipLaunchKernelGGL((k_tn4<4, true>), dim3(g), dim3(256), 0, 0, ta); });
            float t48 = time_any([&] { hipLaunchKernelGGL((k_tn4<8, true>), dim3(g), dim3(256), 0, 0, ta); });
            float t48n = time_any([&] { hipLaunchKernelGGL((k_tn4<8, false>), dim3(g), dim3(256), 0, 0, ta); });
            printf("k_tn %d wgs: ring 8 + barrier %8.1f us %6.1f TF | same, no slab store %8.1f us | k_tn4 (4 waves, 128x128) ring 4 + barrier %8.1f us %6.1f TF, ring 8 + barrier %8.1f us %6.1f TF, ring 8 no barrier %8.1f us\n",
                   g, ts, gf / ts, tns, t44, gf / t44, t48, gf / t48, t48n);
            float t4 = time_any([&] { hipLaunchKernelGGL(k_tn<4>, dim3(g), dim3(512), 0, 0, ta); });
            float t8 = time_any([&] { hipLaunchKernelGGL(k_tn<8>, dim3(g), dim3(512), 0, 0, ta); });
            float t12 = time_any([&] { hipLaunchKernelGGL(k_tn<12>, dim3(g), dim3(512), 0, 0, ta); });
            CK(hipGetLastError());
            printf("k_tn %d workgroups (kchunk %d): ring 4 %8.1f us %6.1f TF | ring 8 %8.1f us %6.1f TF | ring 12 %8.1f us %6.1f TF  (slabs not reduced)\n",
                   g, kchunk, t4, gf / t4, t8, gf / t8, t12, gf / t12);
            // check: sum slabs on the host for a few entries
            std::vector<float> hs((size_t)g * 65536), hD((size_t)M * N);
            CK(hipMemcpy(hs.data(), slab, hs.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hD.data(), D, hD.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0, scale = 0;
            for (int t = 0; t < 40; ++t) {
                const int mm = (t * 37) % 256, nn = (t * 91 + 5) % 256;
                double ref = 0; for (int k = 0; k < M; ++k) ref += (double)hA[(size_t)k * K + mm] * hD[(size_t)k * N + nn];
                double got = 0; for (int z = 0; z < g; ++z) got += hs[(size_t)z * 65536 + mm * 256 + nn];
                worst = fmax(worst, fabs(ref - got)); scale = fmax(scale, fabs(ref));
            }
            printf("   k_tn max |err| %.3g (scale %.3g)\n", worst, scale);
        }
    }
    return 0;
}
