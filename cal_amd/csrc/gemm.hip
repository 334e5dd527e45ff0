// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak) for the
// dense linear layers of the hot path -- the only place MFMA is used (BASELINE.json north_star):
//   x @ W          gcn_conv.py:75, GATConv's lin          -> cal_gemm(NN)
//   Linear(x)      model.py:57-74, 102, 109 (x @ W^T)     -> cal_gemm(NT)
//   dX, dW         autograd of the above                  -> cal_gemm(NT/NN) and cal_gemm(TN, split-K)
//
// Workgroup tile 64x64, K step 32, 4 waves (2x2), one 32x32 accumulator per wave.  Both operands
// are staged in LDS k-major (As[k][i], Bs[k][j]) so each MFMA operand fetch is one conflict-free
// ds_read_b32 of 32 consecutive floats per half-wave; the next K tile is prefetched into registers
// while the current one feeds the MFMAs.  64x64 tiles keep >= 230 workgroups in flight for the
// config-2 shape [7315,128]x[128,128] (256 CUs).
#include "common.hpp"

namespace cal {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int LDT = 65;   // LDS row stride (floats) for tiles filled by transposing scalar stores
constexpr int LDD = 68;   // LDS row stride for tiles filled by direct 16B stores

// Operand tile loader.  The operand is logically T[mn][k] (mn = row of A / column of B).
//   KC = true : memory is [mn][k] row-major (k contiguous)  -> transposing store
//   KC = false: memory is [k][mn] row-major (mn contiguous) -> direct store
template <bool KC>
struct Loader {
    static constexpr int LD = KC ? LDT : LDD;
    float4 r[2];
    // each thread moves 2 float4 per tile (64 x 32 floats / 256 threads)
    __device__ __forceinline__ void load(const float* __restrict__ p, int ld, int mn0, int mn_end, int k0, int k_end,
                                         bool vec) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int f = threadIdx.x + q * 256;
            int mn, k;
            if (KC) { mn = f / (BK / 4); k = (f % (BK / 4)) * 4; }
            else { k = f / (BM / 4); mn = (f % (BM / 4)) * 4; }
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (KC) {
                int gm = mn0 + mn;
                if (gm < mn_end) {
                    const float* src = p + (size_t)gm * ld + k0 + k;
                    if (vec && k0 + k + 3 < k_end) {
                        float4 t = *reinterpret_cast<const float4*>(src);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (k0 + k + j < k_end) v[j] = src[j];
                    }
                }
            } else {
                int gk = k0 + k;
                if (gk < k_end) {
                    const float* src = p + (size_t)gk * ld + mn0 + mn;
                    if (vec && mn0 + mn + 3 < mn_end) {
                        float4 t = *reinterpret_cast<const float4*>(src);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (mn0 + mn + j < mn_end) v[j] = src[j];
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

// C[M,N] (+)= A_op[M,K] * B_op[K,N]  (+bias[N]) (ReLU).  gridDim.z = split-K slices; slice z
// handles k in [z*kchunk, min(K,(z+1)*kchunk)) and writes C + z*M*N (caller reduces when > 1).
template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(256) k_gemm(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                              float* __restrict__ C, int ldc, const float* __restrict__ bias, int relu,
                                              int M, int N, int K, int kchunk, int vecA, int vecB) {
    __shared__ __attribute__((aligned(16))) float As[BK * Loader<A_KC>::LD];
    __shared__ __attribute__((aligned(16))) float Bs[BK * Loader<B_KC>::LD];
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kb = blockIdx.z * kchunk, ke = min(K, kb + kchunk);
    C += (size_t)blockIdx.z * M * ldc;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int li = lane & 31, lk = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    Loader<A_KC> la;
    Loader<B_KC> lb;
    la.load(A, lda, m0, M, kb, ke, vecA);
    lb.load(B, ldb, n0, N, kb, ke, vecB);
    for (int k0 = kb; k0 < ke; k0 += BK) {
        la.store(As);
        lb.store(Bs);
        __syncthreads();
        if (k0 + BK < ke) {
            la.load(A, lda, m0, M, k0 + BK, ke, vecA);
            lb.load(B, ldb, n0, N, k0 + BK, ke, vecB);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a = As[(kk + lk) * Loader<A_KC>::LD + wm + li];
            float b = Bs[(kk + lk) * Loader<B_KC>::LD + wn + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int col = n0 + wn + li;
    if (col < N) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < M) {
                float v = acc[r] + bv;
                if (relu) v = fmaxf(v, 0.f);
                C[(size_t)row * ldc + col] = v;
            }
        }
    }
}

__global__ void k_splitk_reduce(const float* __restrict__ part, float* __restrict__ out, int64_t n, int S,
                                int accumulate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += part[(size_t)z * n + i];
    out[i] = accumulate ? out[i] + s : s;
}

}  // namespace cal

using namespace cal;

static inline int splitk_for(int64_t M, int64_t N, int64_t K) {
    int64_t tiles = (int64_t)cdiv(M, BM) * cdiv(N, BN);
    if (tiles >= 128 || K <= 4 * BK) return 1;
    int64_t s = 512 / tiles;
    int64_t maxs = K / (2 * BK);
    if (s > maxs) s = maxs;
    if (s > 256) s = 256;
    return (int)(s < 1 ? 1 : s);
}

CAL_EXPORT int64_t cal_gemm_ws(int64_t M, int64_t N, int64_t K) {
    int s = splitk_for(M, N, K);
    return s > 1 ? (int64_t)s * M * N : 0;
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
    int S = splitk_for(M, N, K);
    CAL_REQUIRE(S == 1 || ws != nullptr, "split-K workspace missing");
    int lda = (int)(transA ? M : K), ldb = (int)(transB ? K : N);
    int vecA = aligned16(A) && (lda % 4 == 0), vecB = aligned16(B) && (ldb % 4 == 0);
    int kchunk = (int)(((K + S - 1) / S + BK - 1) / BK * BK);
    if (kchunk == 0) kchunk = BK;
    S = K == 0 ? 1 : cdiv(K, kchunk);
    dim3 grid(cdiv(M, BM), cdiv(N, BN), S);
    float* dst = S > 1 ? ws : C;
    const float* b2 = S > 1 ? nullptr : bias;
    int r2 = S > 1 ? 0 : relu;
    if (!transA && !transB)
        hipLaunchKernelGGL((k_gemm<true, false>), grid, dim3(256), 0, stream, A, lda, B, ldb, dst, (int)N, b2, r2, (int)M, (int)N, (int)K, kchunk, vecA, vecB);
    else if (!transA && transB)
        hipLaunchKernelGGL((k_gemm<true, true>), grid, dim3(256), 0, stream, A, lda, B, ldb, dst, (int)N, b2, r2, (int)M, (int)N, (int)K, kchunk, vecA, vecB);
    else if (transA && !transB)
        hipLaunchKernelGGL((k_gemm<false, false>), grid, dim3(256), 0, stream, A, lda, B, ldb, dst, (int)N, b2, r2, (int)M, (int)N, (int)K, kchunk, vecA, vecB);
    else
        hipLaunchKernelGGL((k_gemm<false, true>), grid, dim3(256), 0, stream, A, lda, B, ldb, dst, (int)N, b2, r2, (int)M, (int)N, (int)K, kchunk, vecA, vecB);
    CAL_CHECK_LAUNCH("k_gemm");
    if (S > 1) {
        CAL_REQUIRE(bias == nullptr && !relu, "epilogue not supported with split-K");
        hipLaunchKernelGGL(k_splitk_reduce, dim3(cdiv(M * N, 256)), dim3(256), 0, stream, ws, C, M * N, S, 0);
        CAL_CHECK_LAUNCH("k_splitk_reduce");
    }
    return 0;
}
