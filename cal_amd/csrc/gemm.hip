// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak) for the
// dense linear layers of the hot path -- the only place MFMA is used (BASELINE.json north_star):
//   x @ W          gcn_conv.py:75, GATConv's lin          -> NN
//   Linear(x)      model.py:57-74, 102, 109 (x @ W^T)     -> NT
//   dX, dW         autograd of the above                  -> NT/NN and TN (split-K)
//
// Workgroup tile 64x64, K step 32, 4 waves (2x2), one 32x32 accumulator per wave.  Both operands
// are staged in LDS k-major (As[k][i], Bs[k][j]) so each MFMA operand fetch is one conflict-free
// ds_read_b32 of 32 consecutive floats per half-wave; the next K tile is prefetched into registers
// while the current one feeds the MFMAs.  64x64 tiles keep >= 230 workgroups in flight for the
// config-2 shape [7315,128]x[128,128] (256 CUs).
//
// Fusions (engine.hpp): BatchNorm-apply (+ node-attention row scale) on either operand while it is
// staged (model.py:90,94,112-113,127-131 -- BN outputs are never materialised), bias + ReLU
// epilogue, per-column sum / sum-of-squares of the output (the next BatchNorm's batch statistics)
// and the BN-backward column sums, accumulated in fp64 with one atomic per column per workgroup.
#include "engine.hpp"

namespace cal {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int LDT = 65;   // LDS row stride (floats) for tiles filled by transposing scalar stores
constexpr int LDD = 68;   // LDS row stride for tiles filled by direct 16B stores
constexpr int XMAX = 512; // max feature width of a BN-transformed k-contiguous operand

// Operand tile loader.  The operand is logically T[mn][k] (mn = row of A / column of B).
//   KC = true : memory is [mn][k] row-major (k contiguous)  -> transposing store
//   KC = false: memory is [k][mn] row-major (mn contiguous) -> direct store
// sc/sh: LDS arrays with the BN scale/shift of the feature (storage column) axis, indexed by
// (k - kb) for KC and by the tile-local mn for !KC; null when the operand has no BN transform.
template <bool KC>
struct Loader {
    static constexpr int LD = KC ? LDT : LDD;
    float4 r[2];
    __device__ __forceinline__ void load(const float* __restrict__ p, int ld, int mn0, int mn_end, int k0, int k_end,
                                         int kb, bool vec, const Xform& xf, const float* sc, const float* sh) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int f = threadIdx.x + q * 256;
            int mn, k;
            if (KC) { mn = f / (BK / 4); k = (f % (BK / 4)) * 4; }
            else { k = f / (BM / 4); mn = (f % (BM / 4)) * 4; }
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            bool ok[4] = {false, false, false, false};
            if (KC) {
                int gm = mn0 + mn;
                if (gm < mn_end) {
                    const float* src = p + (size_t)gm * ld + k0 + k;
                    if (vec && k0 + k + 3 < k_end) {
                        float4 t = *reinterpret_cast<const float4*>(src);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                        ok[0] = ok[1] = ok[2] = ok[3] = true;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (k0 + k + j < k_end) { v[j] = src[j]; ok[j] = true; }
                    }
                    float rs = xf.rs ? xf.rs[(size_t)gm * xf.rs_stride] : 1.f;
                    if (xf.rs || sc) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (ok[j]) {
                            float t = rs * v[j];
                            v[j] = sc ? fmaf(t, sc[k0 + k + j - kb], sh[k0 + k + j - kb]) : t;
                        }
                    }
                }
            } else {
                int gk = k0 + k;
                if (gk < k_end) {
                    const float* src = p + (size_t)gk * ld + mn0 + mn;
                    if (vec && mn0 + mn + 3 < mn_end) {
                        float4 t = *reinterpret_cast<const float4*>(src);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                        ok[0] = ok[1] = ok[2] = ok[3] = true;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (mn0 + mn + j < mn_end) { v[j] = src[j]; ok[j] = true; }
                    }
                    float rs = xf.rs ? xf.rs[(size_t)gk * xf.rs_stride] : 1.f;
                    if (xf.rs || sc) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (ok[j]) {
                            float t = rs * v[j];
                            v[j] = sc ? fmaf(t, sc[mn + j], sh[mn + j]) : t;
                        }
                    }
                }
            }
            r[q] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ s) const {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int f = threadIdx.x + q * 256;
            if (KC) {
                int mn = f / (BK / 4), k = (f % (BK / 4)) * 4;
                s[(k + 0) * LD + mn] = r[q].x; s[(k + 1) * LD + mn] = r[q].y;
                s[(k + 2) * LD + mn] = r[q].z; s[(k + 3) * LD + mn] = r[q].w;
            } else {
                int k = f / (BM / 4), mn = (f % (BM / 4)) * 4;
                *reinterpret_cast<float4*>(s + k * LD + mn) = r[q];
            }
        }
    }
};

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(256) k_gemm(const GemmArgs a, int vecA, int vecB) {
    __shared__ __attribute__((aligned(16))) float As[BK * Loader<A_KC>::LD];
    __shared__ __attribute__((aligned(16))) float Bs[BK * Loader<B_KC>::LD];
    __shared__ float xsc[2][A_KC || B_KC ? XMAX : BM];
    __shared__ float xsh[2][A_KC || B_KC ? XMAX : BM];
    __shared__ double red[4][2][32];

    const int batch = blockIdx.z / a.nsplit, split = blockIdx.z % a.nsplit;
    const GemmProb& pr = a.p[batch];
    const int M = a.M, N = a.N, K = a.K;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kb = split * a.kchunk, ke = min(K, kb + a.kchunk);
    float* C = pr.C ? pr.C + (size_t)split * M * a.ldc : nullptr;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int li = lane & 31, lk = lane >> 5;

    // BN scale/shift of the transformed operands into LDS; one block also updates running stats
    const float *sca = nullptr, *sha = nullptr, *scb = nullptr, *shb = nullptr;
    if (pr.xa.has_bn) {
        const int cnt = A_KC ? (ke - kb) : min(BM, M - m0);
        const int c0 = A_KC ? kb : m0;
        for (int t = threadIdx.x; t < cnt; t += 256) {
            bn_scale_shift(pr.xa.bn, c0 + t, xsc[0][t], xsh[0][t]);
            if (pr.xa.bn.update && blockIdx.y == 0 && split == 0 && (A_KC ? blockIdx.x == 0 : true)) bn_update_running(pr.xa.bn, c0 + t);
        }
        sca = xsc[0]; sha = xsh[0];
    }
    if (pr.xb.has_bn) {
        const int cnt = B_KC ? (ke - kb) : min(BN, N - n0);
        const int c0 = B_KC ? kb : n0;
        for (int t = threadIdx.x; t < cnt; t += 256) {
            bn_scale_shift(pr.xb.bn, c0 + t, xsc[1][t], xsh[1][t]);
            if (pr.xb.bn.update && blockIdx.x == 0 && split == 0 && (B_KC ? blockIdx.y == 0 : true)) bn_update_running(pr.xb.bn, c0 + t);
        }
        scb = xsc[1]; shb = xsh[1];
    }
    if (pr.xa.has_bn || pr.xb.has_bn) __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    Loader<A_KC> la;
    Loader<B_KC> lb;
    la.load(pr.A, a.lda, m0, M, kb, ke, kb, vecA, pr.xa, sca, sha);
    lb.load(pr.B, a.ldb, n0, N, kb, ke, kb, vecB, pr.xb, scb, shb);
    for (int k0 = kb; k0 < ke; k0 += BK) {
        la.store(As);
        lb.store(Bs);
        __syncthreads();
        if (k0 + BK < ke) {
            la.load(pr.A, a.lda, m0, M, k0 + BK, ke, kb, vecA, pr.xa, sca, sha);
            lb.load(pr.B, a.ldb, n0, N, k0 + BK, ke, kb, vecB, pr.xb, scb, shb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av = As[(kk + lk) * Loader<A_KC>::LD + wm + li];
            float bv = Bs[(kk + lk) * Loader<B_KC>::LD + wn + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int col = n0 + wn + li;
    const bool cok = col < N;
    const float bv = (pr.bias && cok) ? pr.bias[col] : 0.f;
    const bool want_st = pr.st_sum != nullptr, want_dot = pr.dot_sum != nullptr;
    float amean = 0.f, arstd = 0.f;
    if (want_dot && cok && pr.has_aux) bn_mean_rstd(pr.aux_bn, col, amean, arstd);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < M && cok) {
            float v = acc[r] + bv;
            if (a.relu) v = fmaxf(v, 0.f);
            if (C) C[(size_t)row * a.ldc + col] = v;
            if (want_st) { s1 += (double)v; s2 += (double)v * (double)v; }
            if (want_dot) {
                float x = pr.aux[(size_t)row * N + col];
                if (pr.aux_rs) x *= pr.aux_rs[(size_t)row * pr.aux_rs_stride];
                float xn = (x - amean) * arstd;
                s1 += (double)v;
                s2 += (double)v * (double)xn;
            }
        }
    }
    if (want_st || want_dot) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (lk == 0) { red[wave][0][li] = s1; red[wave][1][li] = s2; }
        __syncthreads();
        if (wave < 2 && lk == 0 && cok) {      // waves 0,1 own columns wn = 0 / 32; add the wm = 32 partner
            double t1 = red[wave][0][li] + red[wave + 2][0][li];
            double t2 = red[wave][1][li] + red[wave + 2][1][li];
            double* d1 = want_st ? pr.st_sum : pr.dot_sum;
            double* d2 = want_st ? pr.st_sq : pr.dot_prod;
            atomicAdd(d1 + col, t1);
            atomicAdd(d2 + col, t2);
        }
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

int launch_gemm(bool transA, bool transB, const GemmArgs& a, int nbatch, hipStream_t stream) {
    if (a.M == 0 || a.N == 0 || nbatch == 0) return 0;
    int vecA = (a.lda % 4 == 0), vecB = (a.ldb % 4 == 0);
    for (int b = 0; b < nbatch; ++b) {
        vecA = vecA && aligned16(a.p[b].A);
        vecB = vecB && aligned16(a.p[b].B);
        const bool a_kc = !transA, b_kc = transB;
        if ((a.p[b].xa.has_bn && a_kc) || (a.p[b].xb.has_bn && b_kc)) {
            if (a.kchunk > XMAX) { set_error("launch_gemm: BN-transformed operand wider than %d", XMAX); return 2; }
        }
    }
    dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), nbatch * a.nsplit);
    if (!transA && !transB) hipLaunchKernelGGL((k_gemm<true, false>), grid, dim3(256), 0, stream, a, vecA, vecB);
    else if (!transA && transB) hipLaunchKernelGGL((k_gemm<true, true>), grid, dim3(256), 0, stream, a, vecA, vecB);
    else if (transA && !transB) hipLaunchKernelGGL((k_gemm<false, false>), grid, dim3(256), 0, stream, a, vecA, vecB);
    else hipLaunchKernelGGL((k_gemm<false, true>), grid, dim3(256), 0, stream, a, vecA, vecB);
    CAL_CHECK_LAUNCH("k_gemm");
    return 0;
}

int splitk_for(int64_t M, int64_t N, int64_t K, int nbatch) {
    int64_t tiles = (int64_t)cdiv(M, BM) * cdiv(N, BN) * nbatch;
    if (tiles >= 128 || K <= 4 * BK) return 1;
    int64_t s = 512 / tiles;
    int64_t maxs = K / (2 * BK);
    if (s > maxs) s = maxs;
    if (s > 256) s = 256;
    return (int)(s < 1 ? 1 : s);
}

void gemm_set_split(GemmArgs& a, int S) {
    int kchunk = (int)((((int64_t)a.K + S - 1) / S + BK - 1) / BK * BK);
    if (kchunk == 0) kchunk = BK;
    a.kchunk = kchunk;
    a.nsplit = a.K == 0 ? 1 : cdiv(a.K, kchunk);
}

}  // namespace cal

using namespace cal;

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
